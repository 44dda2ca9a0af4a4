#!/usr/bin/env python3
"""One rank of the sharded step (nccl, self-exchange; GPU box): multi_gpu.sharded_build (ids tagged with the owner in bits
31:29) against multi_gpu.sharded_build_multipass with P passes (owner byte beside a 32-bit local id) -- wall time per step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29545")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
import _dbg
import multi_gpu as mg

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n, k, P = int(float(os.environ.get("READS", "10")) * 1e6), int(os.environ.get("K", "31")), int(os.environ.get("PASSES", "1"))
check = os.environ.get("CHECK", "1") == "1"
g = _dbg.Graph(device=0)
g.synth_reads(1, n * 5, n, 150, 0.01)
if os.environ.get("CK64"):
    g.set_option("count_kernel_u64", int(os.environ["CK64"]))  # 1 = k_sk_count, 2 = k_sk_count2, 3 = k_sk_count3 for the shards' 64-bit stamps
for name, fn in (("sharded_build (tagged ids)", lambda: mg.sharded_build(g, k, dist, check=check)),
                 (f"sharded_build_multipass P={P} (owner bytes)", lambda: mg.sharded_build_multipass(g, k, dist, P, check=check)),
                 (f"sharded_build_multipass P={P}, records in 2 parts", lambda: mg.sharded_build_multipass(g, k, dist, P, check=check, chunks=2)),
                 (f"sharded_build_multipass P={P}, records in 4 parts", lambda: mg.sharded_build_multipass(g, k, dist, P, check=check, chunks=4))):
    ts = []
    for it in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    st, sz = g.stats(), g.sizes()
    print(name, "ms per step:", ts, {x: round(st[x], 2) for x in ("ms_extract", "ms_partition", "ms_count", "ms_succ")},
          "nodes", sz["n_nodes"], flush=True)
dist.destroy_process_group()
