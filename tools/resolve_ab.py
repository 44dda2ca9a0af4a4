#!/usr/bin/env python3
"""A/B of the resolver's query order (option resolve_sorted: 0 = the askers' bucket order, 1 = grouped by the 512 level-1 groups
of the target) on BASELINE.json configs[1]'s reads: resolver time (ms_succ covers the grouping) and build time, one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import torch  # noqa: E402

torch.zeros(1, device="cuda")
import _dbg  # noqa: E402

for (n, glen, err, k) in ((10000000, 50000000, 0.01, 31), (10000000, 50000000, 0.01, 21), (3000000, 15000000, 0.02, 31)):
    g = _dbg.Graph()
    g.synth_reads(1, glen, n, 150, err)
    for rs in (0, 1, 0, 1):
        g.set_option("resolve_sorted", rs)
        g.build(k)
        g.build(k)
        st, sz = g.stats(), g.sizes()
        print(n, k, "resolve_sorted", rs, sz["n_nodes"], st["n_queries"], "succ ms", round(st["ms_succ"], 3), "build ms", round(st["ms_build_total"], 3), flush=True)
    g.close()
