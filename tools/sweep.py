#!/usr/bin/env python3
"""Phase timings of dbg_build under different engine options (GPU box)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import _dbg

reads = int(os.environ.get("SWEEP_READS", "10000000"))
err = float(os.environ.get("SWEEP_ERR", "0.01"))
kk = int(os.environ.get("SWEEP_K", "31"))
configs = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [{}]
for cfg in configs:
    g = _dbg.Graph()
    for k, v in cfg.items():
        g.set_option(k, v)
    g.synth_reads(1, int(reads * 150 / 30), reads, 150, err)
    for _ in range(2):
        try:
            g.build(kk)
        except _dbg.DbgError as e:
            if "ablation" not in str(e):
                raise
    st = g.stats()
    sz = g.sizes()
    digest = None
    if os.environ.get("SWEEP_DIGEST"):  # 64-bit digest over (k-mer, stamp, 4 counts) of every node: equal graphs <=> equal digests
        sys.path.insert(0, ROOT)
        import bench
        digest = "%016x" % bench.node_digest_gpu(g)
    print(json.dumps({"cfg": cfg, "k": kk, "compact": round(st["ms_compact"], 2), "extract": round(st["ms_extract"], 2), "partition": round(st["ms_partition"], 2),
                      "count": round(st["ms_count"], 2), "succ": round(st["ms_succ"], 2), "csr": round(st["ms_csr"], 2),
                      "total": round(st["ms_build_total"], 2), "n_nodes": sz["n_nodes"], "buckets": st["n_buckets"],
                      "records": st["n_records"], "queries": st["n_queries"], "n_edges": sz["n_edges"], "digest": digest}), flush=True)
    g.close()
