#!/usr/bin/env python3
"""Whole hot path at BASELINE scale on the GPU: build, prune, tips, pull-out reads, walk (timings + sizes)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import _dbg

reads = int(os.environ.get("SWEEP_READS", "10000000"))
err = float(os.environ.get("SWEEP_ERR", "0.01"))
g = _dbg.Graph()
g.synth_reads(1, int(reads * 150 / 30), reads, 150, err)
g.build(31)
out = {}
for rep in range(int(os.environ.get("REPS", "2"))):  # the first pass pays for the arenas (hipMalloc of GBs), the last is reported
  for name, fn in (("build", lambda: g.build(31)), ("refine_edge_order", g.refine_edge_order), ("prune", lambda: g.prune(2)), ("tips", g.remove_tips),
                 ("pull_reads", g.mark_pull_reads), ("walk_index", lambda: g.walk(False, 1 << 20))):
      print(name, "...", flush=True)
      t0 = time.perf_counter()
      try:
          fn()
          out[name + "_wall_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
      except _dbg.DbgError as e:
          out[name + "_error"] = str(e)
          out[name + "_wall_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
st, sz = g.stats(), g.sizes()
out.update({k: round(st[k], 2) for k in ("ms_prune", "ms_tips", "ms_pull_reads", "ms_walk", "ms_build_total")})
out.update({k: sz[k] for k in ("n_nodes", "n_edges", "n_branch", "n_pulled", "tip_rounds", "n_pull_reads", "n_starts",
                               "n_contigs", "contig_chars")})
print(json.dumps(out))
