#!/usr/bin/env python3
"""Where a sharded step spends its time (one rank, nccl; GPU box): wall time per stage of multi_gpu.sharded_build."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist
import _dbg
import multi_gpu as mg

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n, k = int(float(os.environ.get("READS", "10")) * 1e6), 31
g = _dbg.Graph(device=0)
g.synth_reads(1, n * 5, n, 150, 0.01)
for it in range(3):
    T = {}
    def lap(name, t0):
        torch.cuda.synchronize()
        T[name] = round((time.perf_counter() - t0) * 1e3, 2)
    w, me = 1, 0
    t_all = time.perf_counter()
    t0 = time.perf_counter(); send_counts, (w0, w1, st) = g.shard_extract(k, w); lap("shard_extract", t0)
    device = w0.device
    t0 = time.perf_counter(); recv_counts = mg.exchange_counts(dist, send_counts, device); lap("exchange_counts", t0)
    t0 = time.perf_counter()
    r_w0 = mg.alltoallv(dist, w0, send_counts, recv_counts); r_w1 = mg.alltoallv(dist, w1, send_counts, recv_counts)
    r_st = mg.alltoallv(dist, st, send_counts, recv_counts); lap("alltoallv_records", t0)
    t0 = time.perf_counter(); bases = mg.stamp_bases(dist, g.sizes()["n_bytes"], device); lap("stamp_bases", t0)
    t0 = time.perf_counter(); q_starts, q_counts, q_keys = g.shard_build(k, w, me, r_w0, r_w1, r_st, recv_counts, bases); lap("shard_build", t0)
    st_ = g.stats()
    T["  build phases"] = {x: round(st_[x], 2) for x in ("ms_partition", "ms_count", "ms_succ", "ms_build_total")}
    t0 = time.perf_counter(); q_recv = mg.exchange_counts(dist, q_counts, device)
    groups = [q_keys[s:s + c] for s, c in zip(q_starts, q_counts)]
    packed = torch.cat(groups) if groups else q_keys[:0]
    keys_in = mg.alltoallv(dist, packed, q_counts, q_recv); lap("query_exchange", t0)
    t0 = time.perf_counter(); answers_out = g.shard_answer(keys_in); lap("shard_answer", t0)
    t0 = time.perf_counter(); back = mg.alltoallv(dist, answers_out, q_recv, q_counts)
    answers = torch.empty(q_keys.numel(), dtype=torch.int32, device=device)
    off = 0
    for s, c in zip(q_starts, q_counts):
        answers[s:s + c] = back[off:off + c]; off += c
    lap("answer_exchange", t0)
    t0 = time.perf_counter(); g.shard_apply(answers); lap("shard_apply", t0)
    T["total"] = round((time.perf_counter() - t_all) * 1e3, 2)
    print(T, flush=True)
dist.destroy_process_group()
