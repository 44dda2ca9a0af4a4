#!/usr/bin/env python3
"""Wall time and memory of a multi-pass build beyond 2^32 nodes on one GPU (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import torch
import _dbg

torch.zeros(1, device="cuda")
n = int(float(os.environ.get("READS", "45")) * 1e6)
err = float(os.environ.get("ERR", "0.05"))
passes = int(os.environ.get("PASSES", "8"))
g = _dbg.Graph()
t0 = time.perf_counter(); g.synth_reads(1, n * 5, n, 150, err); torch.cuda.synchronize()
print(f"synth {time.perf_counter() - t0:.2f} s", flush=True)
for it in range(2):
    t0 = time.perf_counter()
    g.build_multipass(31, passes)
    dt = time.perf_counter() - t0
    free, total = torch.cuda.mem_get_info()
    sz, st = g.sizes(), g.stats()
    print(f"build_multipass {dt:.2f} s: {sz['n_nodes']} nodes, {sz['n_edges']} edges; device time {st['ms_build_total']:.0f} ms "
          f"(extract {st['ms_extract']:.0f}, partition {st['ms_partition']:.0f}, count {st['ms_count']:.0f}, succ {st['ms_succ']:.0f}); "
          f"HBM in use {(total - free) / 1e9:.1f} GB of {total / 1e9:.1f}", flush=True)
    print([g.part_sizes(p)["n_nodes"] for p in range(passes)], flush=True)
