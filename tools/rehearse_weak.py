#!/usr/bin/env python3
"""Rehearsal of bench.py's N > 1 workload at the REAL per-rank size on one GPU: RANKS handles on cuda:0, one thread per
rank, the real multi_gpu.sharded_build_multipass (one pass per rank: what bench.py --gpus N runs) with an in-process
exchange (tests/inproc_dist.py).  Per-rank reads, genome and seeds are bench.py's (12.5 M x 150 bp per rank over a
RANKS x 62.5 Mbp genome: RANKS / 8 of BASELINE.json configs[2]; K=63: configs[4]'s per-rank size), so every shard holds
what a shard of the 8-GPU run holds: ~4.5e8 nodes (5.7e8 at k = 63) against the 2^32 - 16 local ids of a shard whose
successors carry an owner byte.  Prints per-rank sizes, the fill of the id space, device memory, the number of CSR columns
left without a successor (must be 0: every (k+1)-mer's successor is a node of some shard), and size-independent checks
of the union."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _dbg  # noqa: E402
import inproc_dist  # noqa: E402
import multi_gpu  # noqa: E402

RANKS = int(os.environ.get("RANKS", "4"))
READS = int(float(os.environ.get("READS_PER_RANK", "12.5e6")))
K, L, COV, ERR, SEED = int(os.environ.get("K", "31")), 150, 30.0, 0.01, 1
genome_len = int(RANKS * READS * L / COV)
ID_SPACE = (1 << 32) - 16  # local ids a shard can name (the owner travels in a byte of its own)
torch.zeros(1, device="cuda")


def one(dist, rank):
    g = _dbg.Graph(device=0)
    g.synth_reads(SEED, genome_len, READS, L, ERR, first_read=rank * READS)
    multi_gpu.sharded_build_multipass(g, K, dist, 1)  # warm-up: arenas
    dist.barrier()
    t0 = time.perf_counter()
    multi_gpu.sharded_build_multipass(g, K, dist, 1)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    sz, st = g.sizes(), g.stats()
    free, total = torch.cuda.mem_get_info()
    part = g.part_tensors(0)
    open_cols = int((part["col"] == -1).sum().item())                      # DBG_NO_NODE as int32
    foreign = int((part["col_part"] != rank).sum().item())                 # successors another rank owns
    out = {"rank": rank, "open_columns": open_cols, "successors_owned_elsewhere": foreign, "n_nodes": sz["n_nodes"], "n_edges": sz["n_edges"], "n_kmer_instances": sz["n_kmer_instances"],
           "n_edge_instances": sz["n_edge_instances"], "id_space_fill": round(sz["n_nodes"] / ID_SPACE, 4),
           "ms_count": round(st["ms_count"], 2), "ms_build_total": round(st["ms_build_total"], 2),
           "n_buckets": st["n_buckets"], "wall_ms_all_ranks_time_sliced": round(dt * 1e3, 1),
           "device_mem_used_GB": round((total - free) / 1e9, 1)}
    dist.barrier()
    g.close()
    return out


res = inproc_dist.run_ranks(RANKS, one)
for r in res:
    print(json.dumps(r), flush=True)
n_k = RANKS * READS * (L - K + 1)
tot_nodes = sum(r["n_nodes"] for r in res)
assert sum(r["n_kmer_instances"] for r in res) == n_k, "k-mer instances of all shards != instances in the reads"
assert sum(r["n_edge_instances"] for r in res) == RANKS * READS * (L - K)
assert all(r["open_columns"] == 0 for r in res), "a CSR column was left without its successor"
print(json.dumps({"ranks": RANKS, "reads_per_rank": READS, "k": K, "total_nodes": tot_nodes,
                  "id_space": ID_SPACE, "max_shard_fill_of_id_space": max(r["id_space_fill"] for r in res),
                  "shard_imbalance_max_over_mean": round(max(r["n_nodes"] for r in res) / (tot_nodes / RANKS), 4)}))
