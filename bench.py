#!/usr/bin/env python3
"""Headline benchmark: k-mers/s hashed + graph-built at k=31 (BASELINE.json metric).

One "step" = one dbg_build() (encode -> hash/count -> compaction -> 4-way successors -> CSR;
debruijn.py:98-147 + :213-222) over a synthetic read set that is already resident in HBM.
Workload at N=1: BASELINE.json configs[1] -- 10M x 150 bp reads, k=31, one MI355X.
N>1 (driver: torch.distributed.run, one rank per GPU): see DESIGN.md section "Multi-GPU".

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def b_alg(read_len, k):
    """Algorithmic bytes per k-mer instance: each base read once + one key-slot read
    (8 bytes for k <= 31, 16 for the two-word k-mers of k = 32..63) + one 4-byte counter read and write."""
    return read_len / (read_len - k + 1) + (8 if k <= 31 else 16) + 8


def pmc_traffic(kernel="k_sk_count"):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (collected in its own
    rocprofv3 --pmc passes, FETCH_SIZE x2 per the gfx950 correction); None when the profile is absent."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")
    try:
        with open(path) as fh:
            for row in json.load(fh):
                if kernel in row["kernel"]:
                    return (row["hbm_read_GB_corrected_x2"] + row["hbm_write_GB"]) * 1e9
    except (OSError, ValueError, KeyError):
        pass
    return None


def extras(g, args, k, L, genome_len):
    """Untimed extras on the graph the last step built: the rest of the hot path (debruijn.py:150-347) at the
    same scale, and the same workload without substitution errors.  Not part of `value`."""
    import _dbg
    out = {}
    t = {}
    if k > 31:
        return out  # the extras describe the headline configuration
    for rep in range(2):  # the first pass pays for buffers that later passes reuse (hipMalloc of GBs); the second is reported
        for name, fn in (("refine_edge_order", g.refine_edge_order), ("prune", lambda: g.prune(2)), ("remove_tips", g.remove_tips),
                         ("pull_out_reads", g.mark_pull_reads), ("walk_index_nonfinal", lambda: g.walk(False, 1 << 20))):
            t0 = time.perf_counter()
            fn()
            t[name] = round((time.perf_counter() - t0) * 1e3, 2)
    sz = g.sizes()
    out["rest_of_path_ms"] = t
    out["rest_of_path_sizes"] = {key: sz[key] for key in ("n_branch", "n_pulled", "tip_rounds", "n_pull_reads", "n_starts",
                                                          "n_contigs", "contig_chars")}
    if args.err > 0:
        g0 = _dbg.Graph(device=int(os.environ.get("LOCAL_RANK", "0")))
        g0.synth_reads(args.seed, genome_len, args.reads, L, 0.0)
        g0.build(k)
        t0 = time.perf_counter()
        for _ in range(3):
            g0.build(k)
        dt = (time.perf_counter() - t0) / 3
        ms_count = g0.stats()["ms_count"]
        out["error_free_variant"] = {"value": args.reads * (L - k + 1) / dt, "unit": "k-mers/s", "ms_per_step": dt * 1e3,
                                     "n_nodes": g0.sizes()["n_nodes"], "count_kernel_ms": round(ms_count, 3),
                                     "roofline_frac": args.reads * (L - k + 1) * b_alg(L, k) / (ms_count * 1e-3) / 1e9 / HBM_PEAK_GBS}
        g0.close()
    return out


def cpu_baseline(seed, genome_len, read_len, k, err, sample_reads):
    """oracle/dbg_oracle.c (single-threaded port of the reference's algorithm) on a bounded sample."""
    import numpy as np
    import synth
    from oracle import orc_c
    reads = synth.reads_ascii(seed, genome_len, sample_reads, read_len, err)
    off = np.arange(0, reads.size + 1, read_len, dtype=np.uint64)
    orc_c.lib()
    t0 = time.perf_counter()
    res = orc_c.build(reads.reshape(-1), off, k, export=False)
    dt = time.perf_counter() - t0
    return {"value": res["n_kmer_instances"] / dt, "unit": "k-mers/s", "cores": 1, "kind": "port",
            "sample": f"first {sample_reads} reads of the same synthetic set ({res['n_kmer_instances']} k-mer "
                      f"instances, {res['n_nodes']} distinct), oracle/dbg_oracle.c, {dt:.1f} s, "
                      f"host has {os.cpu_count()} cpus"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--err", type=float, default=0.01, help="per-base substitution rate")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-sample-reads", type=int, default=1_500_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--table-hint", type=int, default=0)
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extras (rest of the path, error-free variant)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        sys.exit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch
    import _dbg
    dist = None
    force_sharded = os.environ.get("BENCH_FORCE_SHARDED", "") == "1"  # 1-rank rehearsal of the RCCL path
    if world > 1 or force_sharded:
        import torch.distributed as dist
        # BENCH_BACKEND=gloo BENCH_SAME_GPU=1: rehearsal of the N>1 path on a one-GPU box (not a measurement)
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if os.environ.get("BENCH_SAME_GPU", "") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        import multi_gpu  # hash-prefix sharded build (RCCL all-to-all)

    L, k = args.read_len, args.k
    n_total = args.reads * world
    genome_len = int(n_total * L / args.coverage)
    g = _dbg.Graph(device=local_rank)
    g.synth_reads(args.seed, genome_len, args.reads, L, args.err, first_read=rank * args.reads)

    def step():
        if dist is None:
            g.build(k, args.table_hint)
            return g
        return multi_gpu.sharded_build(g, k, dist)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ms_count, ms_phases = [], []
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        built = step()
        st = built.stats()
        ms_count.append(st["ms_count"])
        ms_phases.append(st)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    sz = built.sizes()
    n_k_rank = args.reads * (L - k + 1)  # k-mer instances this rank's reads hold
    assert dist is not None or sz["n_kmer_instances"] == n_k_rank, (sz, n_k_rank)
    n_k_total = n_k_rank * world
    value = n_k_total * args.steps / dt
    mean_count_ms = sum(ms_count) / len(ms_count)
    alg_bytes = n_k_rank * b_alg(L, k)  # per launch of the dominant kernel (one launch per step per rank)
    achieved = alg_bytes / (mean_count_ms * 1e-3) / 1e9

    if rank == 0:
        phases = {key: round(sum(p[key] for p in ms_phases) / len(ms_phases), 3)
                  for key in ("ms_extract", "ms_partition", "ms_count", "ms_compact", "ms_succ", "ms_csr", "ms_build_total")}
        out = {
            "metric": f"k-mers/s hashed+graph-built at k={k}", "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{n_total} x {L} bp synthetic reads ({args.reads} per GPU), k={k}, "
                                   f"{args.coverage:g}x coverage of a {genome_len} bp uniform genome, "
                                   f"{args.err * 100:g}% substitutions, seed {args.seed} (BASELINE.json configs[1] at N=1)",
                       "k": k, "reads_per_gpu": args.reads, "read_len": L, "err_rate": args.err,
                       "parallelism": "single table" if world == 1 else f"hash-prefix shard x{world} (RCCL alltoallv)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic() if (world == 1 and args.reads == 10_000_000 and args.err == 0.01
                                                      and k == 31) else None,
                         "kernel": "k_sk_count" if k <= 31 else "k_wcount", "ms_per_launch": mean_count_ms,
                         "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_kmer": b_alg(L, k)},
            "phases_ms": phases,
            "graph": {"n_nodes": sz["n_nodes"], "n_edges": sz["n_edges"], "n_records": st["n_records"],
                      "n_buckets": st["n_buckets"], "n_cross_bucket_successors": st["n_queries"]},
        }
        # the untimed additions must never cost the headline line
        if dist is None and not args.no_extras:
            try:
                out["extras"] = extras(g, args, k, L, genome_len)
            except Exception as e:  # noqa: BLE001
                out["extras_error"] = f"{type(e).__name__}: {e}"
        if not args.no_cpu_baseline and world == 1:  # contract: rank 0 at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(args.seed, genome_len, L, k, args.err,
                                                   min(args.cpu_sample_reads, args.reads))
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline_error"] = f"{type(e).__name__}: {e}"
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
