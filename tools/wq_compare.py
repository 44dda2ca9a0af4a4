#!/usr/bin/env python3
"""A/B of the two-word count kernels (option wcount_kernel: 1 = k_wsk_count, 2 = k_wsk_count2) on the same synthetic reads in
one process: sizes, query count (differs by a few per million: a table near its capacity overflows or not depending on the
order of the claims, and a bucket counted in hash sub-ranges asks for the successors in the other sub-range) and count time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import torch  # noqa: E402

torch.zeros(1, device="cuda")
import _dbg  # noqa: E402

CASES = ((1000000, 5000000, 150, 0.01, 63), (10000000, 50000000, 150, 0.01, 63), (10000000, 50000000, 150, 0.01, 47), (10000000, 50000000, 150, 0.01, 32))
for (n, glen, L, err, k) in CASES:
    g = _dbg.Graph()
    g.synth_reads(1, glen, n, L, err)
    for ck in (1, 2, 1, 2):
        g.set_option("wcount_kernel", ck)
        g.build(k)
        g.build(k)
        st, sz = g.stats(), g.sizes()
        print(n, k, "kernel", ck, sz["n_nodes"], sz["n_edges"], st["n_queries"], st["n_buckets"], "count ms", round(st["ms_count"], 3),
              "build ms", round(st["ms_build_total"], 3), flush=True)
    g.close()
