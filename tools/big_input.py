#!/usr/bin/env python3
"""Build + rest of the path on inputs larger than BASELINE configs[1] (GPU box): tools/big_input.py <million reads> ..."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import _dbg

for arg in sys.argv[1:]:
    n = int(float(arg) * 1e6)
    k = int(os.environ.get("BIG_K", "31"))
    g = _dbg.Graph()
    g.synth_reads(1, n * 5, n, 150, 0.01)  # 30x coverage
    t0 = time.perf_counter()
    g.build(k)
    t1 = time.perf_counter()
    g.build(k)
    t2 = time.perf_counter()
    st, sz = g.stats(), g.sizes()
    assert sz["n_kmer_instances"] == n * (150 - k + 1), sz
    out = {"reads": n, "k": k, "first_build_s": round(t1 - t0, 2), "steady_build_ms": round((t2 - t1) * 1e3, 1),
           "kmers_per_s": n * (150 - k + 1) / (t2 - t1), "n_nodes": sz["n_nodes"], "n_edges": sz["n_edges"],
           "buckets": st["n_buckets"], "ms": {key: round(st[key], 1) for key in ("ms_extract", "ms_partition", "ms_count", "ms_succ")}}
    print(json.dumps(out), flush=True)
    t0 = time.perf_counter()
    g.prune(2); g.remove_tips(); g.mark_pull_reads(); g.walk(False, 1 << 20)
    sz = g.sizes()
    print(json.dumps({"rest_of_path_s": round(time.perf_counter() - t0, 2), "n_branch": sz["n_branch"], "n_pulled": sz["n_pulled"],
                      "n_pull_reads": sz["n_pull_reads"], "n_contigs": sz["n_contigs"]}), flush=True)
    g.close()
