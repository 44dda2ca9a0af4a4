#!/usr/bin/env python3
"""One rank of the strong-scaling bench's N = 1 leg (40 M reads, sharded path) -- for rocprofv3 --hip-trace --stats."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import _dbg, multi_gpu as mg
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
g = _dbg.Graph(device=0)
n = int(float(os.environ.get("READS", "40")) * 1e6)
g.synth_reads(1, n * 5, n, 150, 0.01)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mg.sharded_build_multipass(g, 31, dist, 1)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    print("step", it, round((time.perf_counter() - t0) * 1e3, 1), "ms; free GB", round(free / 1e9, 1), flush=True)
dist.destroy_process_group()
