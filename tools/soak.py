#!/usr/bin/env python3
"""Randomised soak of the build against the C oracle (GPU box): tools/soak.py [n_cases] [seed].

Random k (1..63), ragged read lengths, genome sizes from "everything repeats" to "nothing repeats", error rates,
forced bucket geometries (one- and two-level multisplit, LDS overflow splits, 64-bit stamps are out of reach here).
Prints one line per failure and a summary; exit code 1 on any mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import _dbg
from oracle import orc_c

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(n_cases):
    k = int(rng.choice([1, 2, 3, 5, 8, 11, 12, 13, 16, 21, 27, 31, 32, 33, 40, 47, 63, int(rng.integers(1, 64))]))
    G = int(rng.choice([40, 300, 5000, 100000]))
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=G)
    if rng.random() < 0.2:  # low complexity
        unit = genome[:int(rng.integers(1, 9))]
        genome = np.tile(unit, G // unit.size + 1)[:G]
    n_reads = int(rng.choice([1, 7, 200, 5000, 40000]))
    max_len = int(rng.choice([k + 1, k + 5, 80, 150, 300]))
    lens = rng.integers(0, max_len + 1, size=n_reads)
    lens = np.minimum(lens, G)
    starts = (rng.random(n_reads) * (G - lens + 1)).astype(np.int64)
    err = float(rng.choice([0.0, 0.0, 0.01, 0.05]))
    parts = []
    for s, L in zip(starts, lens):
        r = genome[s:s + L].copy()
        if err and L:
            m = rng.random(L) < err
            r[m] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(m.sum()))
        parts.append(r)
    blob = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    off = np.zeros(n_reads + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    want = orc_c.build(blob, off, k)
    opts = {}
    if k <= 31 and rng.random() < 0.6:
        opts = dict(rng.choice([dict(bucket_bits=1, lds_slots=2048), dict(bucket_bits=4), dict(bucket_bits=11),
                                dict(bucket_bits=18), dict(engine=1), dict(bucket_bits=2, lds_slots=2048)]))
    g = _dbg.Graph()
    for name, v in opts.items():
        g.set_option(name, int(v))
    g.set_reads(blob, off)
    try:
        g.build(k)
        keys, stamps, counts, flags = g.export_nodes()
        hi = g.export_keys_hi()
        o = np.argsort(stamps, kind="stable")
        ok = (g.sizes()["n_nodes"] == want["n_nodes"] and np.array_equal(keys[o], want["keys"]) and np.array_equal(hi[o], want["keys_hi"])
              and np.array_equal(stamps[o], want["stamps"]) and np.array_equal(counts[o], want["counts"])
              and g.sizes()["n_kmer_instances"] == want["n_kmer_instances"])
        if ok and want["n_nodes"]:
            succ = g.export_succ()
            v = (keys.astype(object) | (hi.astype(object) << 64)) if k > 31 else None
            for code in range(4):
                has = counts[:, code] != 0
                if not np.all(succ[has, code] != _dbg.NO_NODE) or not np.all(succ[~has, code] == _dbg.NO_NODE):
                    ok = False
                    break
                if k <= 31:
                    mask = np.uint64((1 << (2 * k)) - 1)
                    ok = ok and np.array_equal(keys[succ[has, code]], ((keys[has] << np.uint64(2)) | np.uint64(code)) & mask)
                elif has.any():
                    m = (1 << (2 * k)) - 1
                    idx = np.nonzero(has)[0][:2000]
                    ok = ok and all(int(v[succ[i, code]]) == ((int(v[i]) << 2) | code) & m for i in idx)
            rp, col, cnt = g.export_csr()
            ok = ok and int(rp[-1]) == int((counts != 0).sum()) and np.array_equal(cnt, counts[counts != 0])
        # rest of the path must at least run and agree on basic counts
        if ok and want["n_nodes"]:
            g.refine_edge_order(); g.prune(2); g.remove_tips(); g.mark_pull_reads()
            try:
                g.walk(False, 1 << 26)
            except _dbg.DbgError as e:  # documented: a small graph whose contig text exceeds max_chars (sizes stay valid)
                if "max_chars" not in str(e):
                    raise
            sz = g.sizes()
            ok = sz["n_starts"] == int(((stamps & np.uint64(1)) == 0).sum()) and sz["n_contigs"] <= sz["n_starts"]
    except _dbg.DbgError as e:
        ok = False
        print("ERROR", e)
    if not ok:
        bad += 1
        print("MISMATCH case", case, dict(k=k, G=G, n_reads=n_reads, max_len=max_len, err=err, opts=opts), flush=True)
    g.close()
    if case % 25 == 24:
        print("..", case + 1, "cases,", bad, "bad", flush=True)
print("soak:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
