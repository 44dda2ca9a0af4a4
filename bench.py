#!/usr/bin/env python3
"""Headline benchmark: k-mers/s hashed + graph-built at k=31 (BASELINE.json metric).

One "step" = one graph build (encode -> super-k-mer records -> bucket partition -> per-bucket LDS hash/count ->
node arrays + 4-way successors as a CSR; debruijn.py:98-147 + :213-222) over a synthetic read set that is already
resident in HBM.

Workloads:
  N = 1              BASELINE.json configs[1]: 10 M x 150 bp reads, k = 31, one MI355X, dbg_build.
  N > 1 (default)    BASELINE.json configs[2] scaled by N/8: 12.5 M reads per rank over an N x 62.5 Mbp genome, hash-prefix
                     sharded build with RCCL all-to-all (multi_gpu.sharded_build_multipass with one pass: every successor
                     is (owner byte, 32-bit local id), so a shard names 2^32 - 16 nodes -- configs[2] fills a tenth of
                     that, configs[4] (--k 63) an eighth); N = 8 is configs[2] itself (100 M x 150 bp).  Per-GPU work is
                     fixed: "scaling": "weak".
  --scaling strong   a FIXED total (--total-reads, default 40 M: 1.45e9 nodes, which the one shard of the N = 1 leg holds)
                     split over the ranks: "scaling": "strong".  N = 1 runs the sharded path on one rank so that every N
                     runs the same code.
The driver launches N > 1 as  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel AND whole step) and
`cpu_baseline` (the oracle's multi-threaded C restatement on the host cores this process may use, full input).
"""
import argparse
import hashlib
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


def kernel_source_hash():
    """Identifies the kernels a stored PMC profile was measured on: sha256 over the device sources."""
    h = hashlib.sha256()
    src = os.path.join(ROOT, "py-debruijn_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(src, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (rocprofv3 --pmc passes of their own,
    FETCH_SIZE x2 per the gfx950 correction) -- only when that profile was collected on exactly these kernel sources
    (it records their hash); otherwise None: a stale figure next to a fresh time would be worse than none."""
    path = os.path.join(ROOT, "profiles", "r03_hbm_traffic_pmc.json")
    try:
        with open(path) as fh:
            prof = json.load(fh)
        if prof.get("kernel_source_hash") != kernel_source_hash():
            return None, None
        for row in prof["kernels"]:
            if kernel in row["kernel"]:
                return (row["hbm_read_GB_corrected_x2"] + row["hbm_write_GB"]) * 1e9, \
                    f"profiles/r03_hbm_traffic_pmc.json (kernel sources {prof['kernel_source_hash']})"
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def usable_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box shows 256 CPUs
    and grants 16)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def extras(g, args, k, L, genome_len, reads_per_rank):
    """Untimed extras on the graph the last step built: the rest of the hot path (debruijn.py:150-347) at the
    same scale, and the same workload without substitution errors.  Not part of `value`."""
    import _dbg
    out = {}
    t = {}
    if k > 31:
        return out  # the extras describe the headline configuration
    t0 = time.perf_counter()
    g.node_tensors()  # the dense per-base views the traversal kernels read are derived from the CSR on first use
    out["dense_views_from_csr_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
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
    # the same reads as a FASTA file image in host memory -> graph: one pageable H2D copy + the parse
    # kernels (dbg_set_reads_fasta), then the build.  Never part of `value` (the timed steps start from resident reads).
    try:
        import numpy as np
        bases, _ = g.copy_reads()
        n = reads_per_rank
        rec = np.empty((n, 3 + L + 1), dtype=np.uint8)
        rec[:, :3] = np.frombuffer(b">r\n", dtype=np.uint8)
        rec[:, 3:3 + L] = bases.reshape(n, L)
        rec[:, -1] = 10
        text = rec.reshape(-1)
        want_sum = g.reads_checksum()
        gf = _dbg.Graph(device=int(os.environ.get("LOCAL_RANK", "0")))
        gf.set_reads_fasta(text)
        gf.build(k)  # arenas
        t0 = time.perf_counter()
        gf.set_reads_fasta(text)
        t1 = time.perf_counter()
        gf.build(k)
        t2 = time.perf_counter()
        out["fasta_ingest"] = {"image_bytes": int(text.size), "set_reads_fasta_ms": round((t1 - t0) * 1e3, 2),
                               "image_GB_per_s": round(text.size / (t1 - t0) / 1e9, 1), "then_build_ms": round((t2 - t1) * 1e3, 2),
                               "image_to_graph_ms": round((t2 - t0) * 1e3, 2),
                               "same_reads_and_graph": bool(gf.reads_checksum() == want_sum and gf.sizes()["n_nodes"] == g.sizes()["n_nodes"]),
                               "note": "host image -> H2D (pageable) + newline scan, header filter, rstrip, compaction on the device; "
                                       "the copy is the PCIe floor (DESIGN.md section 6)"}
        gf.close()
        del text, rec, bases
    except MemoryError:
        out["fasta_ingest"] = None
    # the same graph built in four parts and traversed part by part (part_traversal.py): what a rank of configs[2..4] runs
    # after its build, here with the parts of one GPU -- equal results to the path above are asserted in
    # tests/test_part_traversal.py at this very size
    try:
        import part_traversal
        gp = _dbg.Graph(device=int(os.environ.get("LOCAL_RANK", "0")))
        gp.synth_reads(args.seed, genome_len, reads_per_rank, L, args.err)
        tp = {}
        for rep in range(2):
            t0 = time.perf_counter(); gp.build_multipass(k, 4); tp["build_multipass_4"] = round((time.perf_counter() - t0) * 1e3, 2)
            pt = part_traversal.PartTraversal(gp, k)
            for name, fn in (("prune_and_branch_list", lambda: pt.prune(2)), ("pull_out_reads_and_counter_order", pt.pull_out_reads),
                             ("remove_tips", pt.remove_tips), ("walk_index_nonfinal", pt.walk_index)):
                t0 = time.perf_counter()
                r = fn()
                tp[name] = round((time.perf_counter() - t0) * 1e3, 2)
            n_ctg = int(r["stamp"].size)
        out["traversal_in_parts_ms"] = tp
        out["traversal_in_parts_sizes"] = {"n_branch": int(pt.branch["gid"].size), "n_pulled": int(pt.pulled["gid"].size),
                                           "n_pull_reads": int(pt.read_flags.sum()), "n_contigs": n_ctg,
                                           "same_as_single_graph": bool(int(pt.branch["gid"].size) == sz["n_branch"] and
                                                                        int(pt.pulled["gid"].size) == sz["n_pulled"] and
                                                                        int(pt.read_flags.sum()) == sz["n_pull_reads"] and n_ctg == sz["n_contigs"])}
        gp.close()
    except Exception as e:  # noqa: BLE001
        out["traversal_in_parts_error"] = f"{type(e).__name__}: {e}"
    if args.err > 0:
        g0 = _dbg.Graph(device=int(os.environ.get("LOCAL_RANK", "0")))
        g0.synth_reads(args.seed, genome_len, reads_per_rank, L, 0.0)
        g0.build(k)
        t0 = time.perf_counter()
        for _ in range(3):
            g0.build(k)
        dt = (time.perf_counter() - t0) / 3
        ms_count = g0.stats()["ms_count"]
        n_k = reads_per_rank * (L - k + 1)
        out["error_free_variant"] = {"value": n_k / dt, "unit": "k-mers/s", "ms_per_step": dt * 1e3,
                                     "n_nodes": g0.sizes()["n_nodes"], "count_kernel_ms": round(ms_count, 3),
                                     "roofline_frac_kernel": n_k * b_alg(L, k) / (ms_count * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "roofline_frac_step": n_k * b_alg(L, k) / dt / 1e9 / HBM_PEAK_GBS}
        g0.close()
    return out


def node_digest_gpu(g):
    """The node digest of oracle/orc_c.digest, computed on the device with torch (int64 arithmetic wraps like uint64)."""
    import torch
    nodes = g.node_tensors()
    keys, stamps, counts = nodes["keys"], nodes["stamps"], nodes["counts"].view(-1, 4).to(torch.int64) & 0xFFFFFFFF

    def c64(v):  # a 64-bit constant as the int64 with the same bits
        return v - (1 << 64) if v >= (1 << 63) else v

    def srl33(x):  # logical shift right by 33 on int64
        return (x >> 33) & ((1 << 31) - 1)

    def mix(x):
        x = x ^ srl33(x)
        x = x * c64(0xff51afd7ed558ccd)
        x = x ^ srl33(x)
        x = x * c64(0xc4ceb9fe1a85ec53)
        return x ^ srl33(x)

    w = counts[:, 0] + 3 * counts[:, 1] + 5 * counts[:, 2] + 7 * counts[:, 3] + 1
    return int(mix(keys ^ mix(stamps) ^ mix(w)).sum().item()) & ((1 << 64) - 1)


def cpu_baseline(g, read_len, k, n_reads, gpu_sizes, gpu_digest):
    """oracle/dbg_oracle.c, multi-threaded (orc_build_mt_partitioned: every k-mer window is hashed once and handed to the
    thread that owns its hash slice; one table per thread, no locks), on the FULL input of the N = 1 workload and on every
    core this process may use; orc_build_mt (every thread scans all reads, round 2's baseline) beside it; plus the Python restatement of the reference's
    algorithm (oracle/dbg_oracle.py, one core: Python is single-threaded) timed at BASELINE.json configs[0]."""
    import numpy as np
    import synth
    from oracle import orc_c
    cores = usable_cores()
    reads, off = g.copy_reads()  # the very bytes the GPU built from (the device generator is bit-identical to synth.py)
    orc_c.lib()
    t0 = time.perf_counter()
    res = orc_c.build_mt(reads, off, k, cores, partition_once=True)
    dt = time.perf_counter() - t0
    # round 2's baseline beside it, on a quarter of the reads (its rate does not depend on the input size; the whole default run
    # stays within ~25 s of CPU work)
    nq = max(1, n_reads // 4)
    t0 = time.perf_counter()
    res_scan = orc_c.build_mt(reads[:int(off[nq])], off[:nq + 1], k, cores)
    dt_scan = time.perf_counter() - t0
    with open("/proc/cpuinfo") as fh:
        model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "?")
    same = (res["n_nodes"] == gpu_sizes["n_nodes"] and res["n_edges"] == gpu_sizes["n_edges"] and
            res["n_kmer_instances"] == gpu_sizes["n_kmer_instances"] and
            (gpu_digest is None or res["digest"] == gpu_digest))
    # one thread on a sixteenth of the reads
    one = None
    try:
        n1 = max(1, n_reads // 16)
        r1, o1 = reads[:int(off[n1])], off[:n1 + 1]
        t1 = time.perf_counter()
        res1 = orc_c.build_mt(r1, o1, k, 1)
        dt1 = time.perf_counter() - t1
        one = {"value": res1["n_kmer_instances"] / dt1, "unit": "k-mers/s", "cores": 1, "reads": int(n1), "seconds": round(dt1, 2)}
    except Exception as e:  # noqa: BLE001
        one = {"error": f"{type(e).__name__}: {e}"}
    out = {"value": res["n_kmer_instances"] / dt, "unit": "k-mers/s", "cores": cores, "kind": "port",
           "sample": f"the full input of this run ({n_reads} reads, {res['n_kmer_instances']} k-mer instances, "
                     f"{res['n_nodes']} distinct), oracle/dbg_oracle.c orc_build_mt_partitioned with {cores} threads, {dt:.1f} s: "
                     f"each thread rolls and hashes its share of the reads once and hands every k-mer to the thread owning "
                     f"its hash slice, which builds that slice's table (16 B per instance held between the phases); "
                     f"{model}, {os.cpu_count()} cpus visible, {cores} usable (affinity / cgroup quota)",
           "one_thread": one,
           "every_thread_scans_all_reads": {"value": res_scan["n_kmer_instances"] / dt_scan, "unit": "k-mers/s", "cores": cores,
                                            "seconds": round(dt_scan, 1), "reads": int(nq),
                                            "what": "orc_build_mt, round 2's baseline, on a quarter of the reads: no "
                                            "list between the phases, the scan is repeated per thread"},
           "same_graph_as_gpu": bool(same),
           "compared": "node, edge and k-mer instance totals" + ("" if gpu_digest is None else
                                                                    " + 64-bit digest over (k-mer, stamp, 4 counts) of every node")}
    # BASELINE.json configs[0] through the Python restatement (same dict/str algorithm as the reference, one core)
    try:
        import contextlib
        import io
        from oracle import dbg_oracle as orc
        r0 = synth.reads_list(1, 100_000, 10_000, 100, 0.01)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            (verts, _edges), *_ = orc.construct_graph(r0, 21, threshold=2)
        dt0 = time.perf_counter() - t0
        out["python_restatement_config0"] = {"value": 10_000 * (100 - 21 + 1) / dt0, "unit": "k-mers/s", "cores": 1,
                                             "seconds": round(dt0, 2), "n_nodes": len(verts),
                                             "what": "oracle/dbg_oracle.py construct_graph (graph + pruning + tips + pull-out), "
                                                     "10k x 100 bp, k=21, 1% errors"}
    except Exception as e:  # noqa: BLE001
        out["python_restatement_config0_error"] = f"{type(e).__name__}: {e}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: 10 M at N = 1, 12.5 M at N > 1)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--total-reads", type=int, default=40_000_000, help="--scaling strong: the fixed total")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--err", type=float, default=0.01, help="per-base substitution rate")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exchange-check", action="store_true", help="sharded builds: skip the per-message digests")
    ap.add_argument("--chunks", type=int, default=1, help="sharded builds: cut and send the records in this many parts, the "
                    "exchange of one under the extraction of the next (multi_gpu._exchange_records_in_parts); 1 = one exchange")
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
    sharded = world > 1 or force_sharded or args.scaling == "strong"
    if sharded:
        import torch.distributed as dist
        # BENCH_BACKEND=gloo BENCH_SAME_GPU=1: rehearsal of the N>1 path on a one-GPU box (not a measurement)
        backend = os.environ.get("BENCH_BACKEND", "nccl")
        if os.environ.get("BENCH_SAME_GPU", "") == "1":
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29531")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        import multi_gpu  # hash-prefix sharded build (RCCL all-to-all)

    L, k = args.read_len, args.k
    if args.scaling == "strong":
        n_total = args.total_reads // world * world
        reads_per_rank = n_total // world
        workload = (f"strong scaling: a fixed total of {n_total} x {L} bp synthetic reads split over {world} rank(s), "
                    f"{reads_per_rank} per GPU")
    else:
        reads_per_rank = args.reads or (10_000_000 if world == 1 else 12_500_000)
        n_total = reads_per_rank * world
        which = ("the read set of BASELINE.json configs[1] at another k" if world == 1 and reads_per_rank == 10_000_000 and k != 31 else
                 "BASELINE.json configs[1]" if world == 1 and reads_per_rank == 10_000_000 else
                 "BASELINE.json configs[2]" if n_total == 100_000_000 and world == 8 else
                 f"BASELINE.json configs[2] scaled to {world}/8 of its reads" if reads_per_rank == 12_500_000 else "custom size")
        workload = f"{n_total} x {L} bp synthetic reads ({reads_per_rank} per GPU): {which}"
    genome_len = int(n_total * L / args.coverage)
    workload += (f"; k={k}, {args.coverage:g}x coverage of a {genome_len} bp uniform genome, "
                 f"{args.err * 100:g}% substitutions, seed {args.seed}")
    g = _dbg.Graph(device=local_rank)
    g.synth_reads(args.seed, genome_len, reads_per_rank, L, args.err, first_read=rank * reads_per_rank)

    def step():
        if not sharded:
            g.build(k, args.table_hint)
            return g
        # one pass per rank: the plain sharded build with (owner byte, 32-bit local id) successors
        return multi_gpu.sharded_build_multipass(g, k, dist, 1, check=not args.no_exchange_check, chunks=args.chunks)

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

    # the first build on a read set also scans it for its alphabet (cached per read set): what the multi-k driver pays
    # after every `sequences.extend(...)`; the timed steps above are the steady state
    first_build_ms = None
    if not sharded:
        g.synth_reads(args.seed, genome_len, reads_per_rank, L, args.err, first_read=rank * reads_per_rank)
        sync()
        t1 = time.perf_counter()
        step()
        sync()
        first_build_ms = (time.perf_counter() - t1) * 1e3
    sz = built.sizes()
    n_k_rank = reads_per_rank * (L - k + 1)  # k-mer instances this rank's reads hold
    assert sharded or sz["n_kmer_instances"] == n_k_rank, (sz, n_k_rank)
    n_k_total = n_k_rank * world
    value = n_k_total * args.steps / dt
    mean_count_ms = sum(ms_count) / len(ms_count)
    # the dominant kernel: one launch per step and rank over the k-mer instances of the buckets this rank owns
    n_k_launch = sz["n_kmer_instances"] if sharded else n_k_rank
    alg_bytes = n_k_launch * b_alg(L, k)
    achieved = alg_bytes / (mean_count_ms * 1e-3) / 1e9
    # the whole step, as SURVEY.md 8d defines the fraction: N_k x B_alg / t (first encode kernel -> CSR complete), per GPU
    achieved_step = n_k_rank * b_alg(L, k) / (dt / args.steps) / 1e9

    if rank == 0:
        # 32-bit stamps (a single GPU below 2 GiB of reads): k_sk_count2 / k_wsk_count2 (k > 31); 64-bit stamps (shards):
        # k_sk_count3 / k_wsk_count
        narrow = not sharded and reads_per_rank * L < (1 << 31)
        kernel = ("k_sk_count2" if narrow else "k_sk_count3") if k <= 31 else ("k_wsk_count2" if narrow else "k_wsk_count")
        phases = {key: round(sum(p[key] for p in ms_phases) / len(ms_phases), 3)
                  for key in ("ms_extract", "ms_partition", "ms_count", "ms_compact", "ms_succ", "ms_csr", "ms_build_total")}
        traffic, traffic_src = (pmc_traffic(kernel) if (not sharded and reads_per_rank == 10_000_000 and args.err == 0.01 and
                                                        k == 31 and L == 150) else (None, None))
        out = {
            "metric": f"k-mers/s hashed+graph-built at k={k}", "value": value, "unit": "k-mers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "first_build_ms": first_build_ms, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": workload, "k": k, "reads_per_gpu": reads_per_rank, "reads_total": n_total, "read_len": L,
                       "err_rate": args.err,
                       "parallelism": "single table" if not sharded else f"hash-prefix shard x{world} (RCCL alltoallv)" + (f", records in {args.chunks} parts" if args.chunks > 1 else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_kernel": achieved / HBM_PEAK_GBS,
                         "frac_step": achieved_step / HBM_PEAK_GBS, "achieved_step": achieved_step,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel, "ms_per_launch": mean_count_ms,
                         "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_kmer": b_alg(L, k),
                         "note": "frac / frac_kernel: dominant kernel (HIP events on the library's stream); frac_step: "
                                 "N_k x B_alg over the whole step per GPU (SURVEY.md 8d)"},
            "phases_ms": phases,
            "graph": {"n_nodes": sz["n_nodes"], "n_edges": sz["n_edges"], "n_records": st["n_records"],
                      "n_buckets": st["n_buckets"], "n_cross_bucket_successors": st["n_queries"]},
        }
        gpu_digest = None
        # the untimed additions must never cost the headline line
        if not sharded and not args.no_extras:
            try:
                out["extras"] = extras(g, args, k, L, genome_len, reads_per_rank)
            except Exception as e:  # noqa: BLE001
                out["extras_error"] = f"{type(e).__name__}: {e}"
        if not sharded and not args.no_cpu_baseline and k <= 31:
            try:
                gpu_digest = node_digest_gpu(g)
            except Exception as e:  # noqa: BLE001
                out["gpu_digest_error"] = f"{type(e).__name__}: {e}"
        if not args.no_cpu_baseline and world == 1 and k <= 31:  # contract: rank 0 at N=1 only
            try:
                out["cpu_baseline"] = cpu_baseline(g, L, k, reads_per_rank,
                                                   sz if not sharded else {"n_nodes": sz["n_nodes"], "n_edges": sz["n_edges"],
                                                                           "n_kmer_instances": sz["n_kmer_instances"]},
                                                   gpu_digest)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline_error"] = f"{type(e).__name__}: {e}"
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
