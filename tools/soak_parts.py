#!/usr/bin/env python3
"""Randomised soak of the traversal in parts (GPU box): tools/soak_parts.py [n] [seed].

Every case: random rank count (1..8, in-process: tests/inproc_dist.py), passes, k (13..63), read set (synthetic genome reads,
sometimes with tandem repeats so that chains run into cycles inside and across parts), threshold; the whole path on the
parts -- branch_kmer, already_pull_out, pull_out_read, contigs (text and getScore) -- must equal the Python restatement of
the reference on all reads (oracle/dbg_oracle.py, iterative, any size)."""
import contextlib
import io
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
import part_traversal  # noqa: E402
from oracle import dbg_oracle as orc  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
only = {int(x) for x in os.environ["SOAK_CASES"].split(",")} if os.environ.get("SOAK_CASES") else None
bad = 0
for case in range(n_cases):
    ranks = int(rng.choice([1, 1, 2, 4, 8]))
    passes = int(rng.choice([1, 1, 2, 4]))
    k = int(rng.choice([13, 17, 21, 31, 33, 47, 63, int(rng.integers(13, 64))]))
    L = int(rng.choice([k + 2, 80, 120]))
    n_reads = int(rng.choice([60, 400, 1500])) // ranks * ranks
    err = float(rng.choice([0.0, 0.01, 0.03]))
    thr = int(rng.choice([1, 2, 3]))
    G = max(3 * L, n_reads * L // int(rng.choice([4, 15])))
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=G)
    if rng.random() < 0.4:   # a tandem repeat longer than k: the walk meets `current in vec`
        unit = genome[:int(rng.integers(3, 25))]
        at = int(rng.integers(0, max(1, G - 3 * k)))
        rep = np.tile(unit, 3 * k // unit.size + 2)[:min(3 * k, G - at)]
        genome[at:at + rep.size] = rep
    starts = rng.integers(0, G - L + 1, size=n_reads)
    reads = np.stack([genome[s:s + L] for s in starts]).copy()
    m = rng.random(reads.shape) < err
    reads[m] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(m.sum()))
    desc = f"ranks={ranks} passes={passes} k={k} L={L} reads={n_reads} err={err} thr={thr} G={G}"
    if only is not None and case not in only:
        continue
    if os.environ.get("SOAK_VERBOSE"):
        print(f"case {case}: {desc}", flush=True)
    per = n_reads // ranks
    strs = [row.tobytes().decode() for row in reads]
    with contextlib.redirect_stdout(io.StringIO()):
        og, opull, obranch, opulled, oect = orc.construct_graph(strs, k, threshold=thr)
        octg = orc.output_contigs(og, obranch, opulled)
    oscores = [orc.get_score(oect, c, k) for c in octg]

    def one(dist, rank):
        rd = reads[rank * per:(rank + 1) * per]
        g = _dbg.Graph(device=0)
        g.set_reads(rd.reshape(-1), np.arange(0, rd.size + 1, L, dtype=np.uint64))
        if dist is None:
            g.build_multipass(k, passes)
        else:
            multi_gpu.sharded_build_multipass(g, k, dist, passes)
        t, flags, br, pu = part_traversal.construct_graph(g, k, thr, dist)
        ctg = part_traversal.output_contigs(t)
        out = (br, pu, flags, ctg.texts(range(len(ctg))), ctg.scores.tolist())
        g.close()
        return out

    try:
        got = [one(None, 0)] if ranks == 1 else inproc_dist.run_ranks(ranks, one)
        flags = np.concatenate([r[2] for r in got]).astype(bool)
        ok = [s for s, f in zip(strs, flags) if f] == list(opull)
        for br, pu, _, texts, scores in got:
            ok = ok and br == list(obranch) and pu == list(opulled) and texts == list(octg) and scores == oscores
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: {desc}", flush=True)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"ERROR case {case}: {desc}: {type(e).__name__}: {e}", flush=True)
print(f"{n_cases} cases, {bad} bad")
sys.exit(1 if bad else 0)
