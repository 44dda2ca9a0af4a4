#!/usr/bin/env python3
"""One-rank sharded build vs single build: node / edge totals (GPU box; READS in millions, K)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
for key, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29547"), ("RANK", "0"), ("WORLD_SIZE", "1")):
    os.environ.setdefault(key, val)
import torch
import torch.distributed as dist
import _dbg
import multi_gpu as mg

if os.environ.get("SHARD_MAX_MSG"):
    mg.MAX_MESSAGE_BYTES = int(os.environ["SHARD_MAX_MSG"])
torch.cuda.set_device(0)
if os.environ.get("BACKEND", "nccl") == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo")
k = int(os.environ.get("K", "63"))
for m in [float(x) for x in os.environ.get("READS", "1,4,10").split(",")]:
    n = int(m * 1e6)
    g = _dbg.Graph(device=0)
    g.synth_reads(1, n * 5, n, 150, 0.01)
    g.build(k)
    a = g.sizes()
    s = mg.sharded_build(g, k, dist)
    b = s.sizes()
    print(n, "single", a["n_nodes"], a["n_edges"], "sharded", b["n_nodes"], b["n_edges"],
          "OK" if (a["n_nodes"], a["n_edges"]) == (b["n_nodes"], b["n_edges"]) else "MISMATCH", flush=True)
    g.close()
dist.destroy_process_group()
