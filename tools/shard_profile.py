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
n, k = int(float(os.environ.get("READS", "10")) * 1e6), int(os.environ.get("K", "31"))
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
    t0 = time.perf_counter()
    words = g.shard_record_layout()[0]
    metas = mg._all_gather_ints(dist, [g.sizes()["n_bytes"], words] + g.shard_bucket_counts(), device)
    bases = [0]
    sender_buckets = [m_r[2:514] for m_r in metas]
    recv_counts = [sum(row) for row in sender_buckets]
    lap("meta_all_gather", t0)
    xc = mg.ExchangeCheck(dist) if os.environ.get("CHECK", "1") == "1" else mg._NoCheck(dist)
    t0 = time.perf_counter()
    big = max(max(send_counts), max(recv_counts))  # (one rank: every pair is this pair)
    r_w0 = xc.alltoallv(w0, [words * c for c in send_counts], [words * c for c in recv_counts], "w0", words * big); r_w1 = xc.alltoallv(w1, send_counts, recv_counts, "w1", big)
    r_st = xc.alltoallv(st, send_counts, recv_counts, "st", big); xc.verify(); lap("alltoallv_records(+digests)", t0)
    pre = sender_buckets if os.environ.get("PRESPLIT", "1") == "1" else None
    t0 = time.perf_counter(); q_starts, q_counts, q_keys = g.shard_build(k, w, me, r_w0, r_w1, r_st, recv_counts, bases, pre); lap("shard_build", t0)
    st_ = g.stats()
    T["  build phases"] = {x: round(st_[x], 2) for x in ("ms_partition", "ms_compact", "ms_count", "ms_succ", "ms_build_total")}
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
