#!/usr/bin/env python3
"""Randomised soak of the whole drop-in surface against the Python oracle (GPU box): tools/soak_path.py [n] [seed].

Small random inputs over several alphabets, k = 2..63, repeats longer than k, tandem repeats, errors near read ends,
duplicates; half of the cases force the list-ranking walk (DBG_WALK_JUMP_MIN=0) and the lazy Mapping views."""
import contextlib
import io
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import debruijn as prod
from golden_util import canonical
from oracle import dbg_oracle as orc

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
aa = "ACDEFGHIKLMNPQRSTVWY"
bad = 0
for case in range(n_cases):
    alpha = rng.choice(["ACGT", "ACGT", "AC", "ACG", aa, aa[:5], "ACGTN", "acgtACGT"])
    k = rng.choice([2, 3, 4, 5, 7, 9, 11, 12, 13, 16, 21, 31, 32, 33, 40, 63, rng.randint(2, 63)])
    rnd = lambda n: "".join(rng.choice(alpha) for _ in range(n))
    kind = rng.random()
    if kind < 0.4:
        R = rnd(k + rng.randint(1, 25))
        G = rnd(rng.randint(3, 40)) + R + rnd(rng.randint(3, 50)) + R + rnd(rng.randint(3, 40))
    elif kind < 0.6:
        unit = rnd(rng.randint(1, 15))
        G = rnd(rng.randint(0, 9)) + unit * ((k + 40) // len(unit) + 2) + rnd(rng.randint(0, 9))
    else:
        G = rnd(rng.randint(k + 3, k + 150))
    reads = []
    for _ in range(rng.randint(1, 25)):
        L = rng.randint(max(1, k - 3), min(len(G), k + 50))
        st = rng.randint(0, len(G) - L)
        r = list(G[st:st + L])
        if rng.random() < 0.45:
            j = L - 1 - rng.randint(0, 5) if rng.random() < 0.6 else rng.randrange(L)
            r[max(j, 0)] = rng.choice(alpha)
        reads.append("".join(r))
    if rng.random() < 0.3:
        reads += reads[:rng.randint(1, 3)]
    thr = rng.choice([1, 2, 2, 3, 3, 5])
    final = rng.random() < 0.35
    force = rng.random() < 0.5
    os.environ["DBG_WALK_JUMP_MIN"] = "0" if force else "1048576"
    prod.LAZY_MIN_NODES = 0 if force else 2_000_000
    res = []
    try:
        for mod in (prod, orc):
            with contextlib.redirect_stdout(io.StringIO()) as buf:
                g, pull, branch, pulled, ect = mod.construct_graph(list(reads), k, threshold=thr, final=final)
                contigs = mod.output_contigs(g, branch, pulled)
            r = canonical(g, pull, branch, pulled, ect, contigs)
            r["stdout"] = buf.getvalue()
            res.append(r)
        ok = all(res[0][f] == res[1][f] for f in res[1])
        if ok:
            scores = [orc.get_score(dict(map(tuple, res[1]["edge_count_table"])), c, k) for c in res[1]["contigs"]]
            ok = list(contigs.scores) == scores if hasattr(contigs, "scores") else True
    except Exception as e:  # noqa: BLE001
        ok = False
        print("ERROR", type(e).__name__, e)
    if not ok:
        bad += 1
        print("MISMATCH case", case, dict(alpha=alpha, k=k, thr=thr, final=final, force=force, reads=reads), flush=True)
    if case % 50 == 49:
        print("..", case + 1, "cases,", bad, "bad", flush=True)
print("soak_path:", n_cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
