#!/usr/bin/env python3
"""Randomised soak of the sharded paths on one GPU (in-process ranks, tests/inproc_dist.py): tools/soak_shard.py [n] [seed].

Every case: random rank count (1..8), k (13..63), read set, stamp widths per rank, and -- for k <= 31 -- either the plain
sharded build or ranks x passes (multi_gpu.sharded_build_multipass); the union of what the ranks hold must equal the C
oracle (keys, stamps, counts), and for ranks x passes every successor (virtual shard, id) must be the shifted k-mer."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _dbg  # noqa: E402
import inproc_dist  # noqa: E402
import multi_gpu  # noqa: E402
import synth  # noqa: E402
from oracle import orc_c  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
only = {int(x) for x in os.environ["SOAK_CASES"].split(",")} if os.environ.get("SOAK_CASES") else None  # re-run single cases


def dense_counts(d):
    n = d["keys"].size
    counts = np.zeros((n, 4), dtype=np.uint32)
    e = d["row_ptr"][:-1].astype(np.int64).copy()
    for code in range(4):
        has = ((d["flags"] >> (1 + code)) & 1).astype(bool)
        counts[has, code] = d["cnt"][e[has]]
        e[has] += 1
    return counts


for case in range(n_cases):
    ranks = int(rng.choice([1, 2, 4, 8]))
    k = int(rng.choice([13, 17, 21, 27, 31, 31, 33, 47, 63, int(rng.integers(13, 64))]))
    read_len = int(rng.choice([k + 3, 100, 150]))
    per = int(rng.choice([40, 500, 3000]))
    err = float(rng.choice([0.0, 0.01, 0.05]))
    seed = int(rng.integers(1, 1 << 30))
    genome = max(4 * read_len, ranks * per * read_len // int(rng.choice([3, 20])))
    passes = int(rng.choice([0, 0, 1, 2, 4, 8])) if k <= 31 else 0
    while passes and ranks * passes > 64:
        passes //= 2
    wide = [r for r in range(ranks) if k <= 31 and rng.random() < 0.3]
    desc = f"ranks={ranks} passes={passes} k={k} L={read_len} per={per} err={err} stamp64={wide} seed={seed}"
    if only is not None and case not in only:
        continue
    if os.environ.get("SOAK_VERBOSE"):
        print(f"case {case}: {desc}", flush=True)

    def reads_of(r):
        return synth.reads_ascii(seed, genome, per, read_len, err, first_read=r * per)

    def one(dist, rank):
        g = _dbg.Graph(device=0)
        if rank in wide:
            g.set_option("shard_stamp64", 1)
        rd = reads_of(rank)
        g.set_reads(rd.reshape(-1), np.arange(0, rd.size + 1, read_len, dtype=np.uint64))
        if passes:
            multi_gpu.sharded_build_multipass(g, k, dist, passes)
            out = [g.export_part(p) for p in range(g.part_count())]
        else:
            multi_gpu.sharded_build(g, k, dist)
            keys, stamps, counts, _ = g.export_nodes()
            out = {"keys": keys, "keys_hi": g.export_keys_hi(), "stamps": stamps, "counts": counts}
        g.close()
        return out

    try:
        got = inproc_dist.run_ranks(ranks, one)
        allr = np.concatenate([reads_of(r) for r in range(ranks)])
        want = orc_c.build(allr.reshape(-1), np.arange(0, allr.size + 1, read_len, dtype=np.uint64), k)
        if passes:
            parts = [d for rp in got for d in rp]
            keys = np.concatenate([d["keys"] for d in parts]); stamps = np.concatenate([d["stamps"] for d in parts])
            counts = np.concatenate([dense_counts(d) for d in parts])
            mask = np.uint64((1 << (2 * k)) - 1)
            for d in parts:
                e = d["row_ptr"][:-1].astype(np.int64).copy()
                for code in range(4):
                    has = ((d["flags"] >> (1 + code)) & 1).astype(bool)
                    cols, owners = d["col"][e[has]], d["col_part"][e[has]]
                    gk = np.empty(cols.size, dtype=np.uint64)
                    for q, dq in enumerate(parts):
                        sel = owners == q
                        gk[sel] = dq["keys"][cols[sel]]
                    assert np.array_equal(gk, ((d["keys"][has] << np.uint64(2)) | np.uint64(code)) & mask), "successor"
                    e[has] += 1
            hi = None
        else:
            keys = np.concatenate([d["keys"] for d in got]); stamps = np.concatenate([d["stamps"] for d in got])
            counts = np.concatenate([d["counts"] for d in got]); hi = np.concatenate([d["keys_hi"] for d in got])
        o = np.argsort(stamps, kind="stable")
        assert np.array_equal(keys[o], want["keys"]) and np.array_equal(stamps[o], want["stamps"]), "nodes"
        assert np.array_equal(counts[o], want["counts"]), "counts"
        if hi is not None and k > 32:
            assert np.array_equal(hi[o], want["keys_hi"]), "keys_hi"
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"case {case} FAILED ({type(e).__name__}: {str(e)[:200]}): {desc}", flush=True)
    if (case + 1) % 10 == 0:
        print(f".. {case + 1} cases, {bad} bad", flush=True)
print(f"soak_shard: {n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
