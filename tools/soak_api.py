#!/usr/bin/env python3
"""Randomised sequences of calls on ONE handle (GPU box): tools/soak_api.py [n_sequences] [seed].

The handle keeps arenas, parts, shard state and walk tables across builds; this soak mixes single builds (one- and
two-word k-mers, peptides), multi-pass builds, one-rank sharded builds, the traversal calls and exports in random order,
with new reads in between.  Every build is checked against the C oracle (node, edge and instance totals), every
traversal against a fresh handle that runs the same calls from scratch; calls that must be refused have to raise
DbgError, not crash."""
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

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0


def new_reads():
    L = int(rng.choice([40, 100, 150]))
    n = int(rng.choice([1, 30, 800, 6000]))
    r = synth.reads_ascii(int(rng.integers(1, 1 << 30)), max(4 * L, n * L // int(rng.choice([2, 25]))), n, L, float(rng.choice([0, 0.01, 0.04])))
    return r, L


def path(g):
    g.refine_edge_order(); g.prune(2); g.remove_tips(); g.mark_pull_reads(); g.walk(False)
    off, chars, score, stamp, seq = g.export_contigs()
    co = np.lexsort((seq, stamp))
    text = chars.tobytes()
    return [text[int(off[i]):int(off[i + 1])] for i in co], score[co].tolist(), g.sizes()["n_pulled"], g.sizes()["n_pull_reads"]


for sq in range(n_seq):
    g = _dbg.Graph()
    reads, L = new_reads()
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, L, dtype=np.uint64))
    log = []
    try:
        for step in range(int(rng.integers(3, 9))):
            op = str(rng.choice(["build", "build", "multipass", "shard1", "path", "reads", "export", "refused"]))
            k = int(rng.choice([13, 21, 31, 33, 47, 63, int(rng.integers(5, 64))]))
            k = min(k, L - 2)
            log.append((op, k))
            off = np.arange(0, reads.size + 1, L, dtype=np.uint64)
            if op == "reads":
                reads, L = new_reads()
                g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, L, dtype=np.uint64))
            elif op == "build":
                g.build(k)
                want = orc_c.build(reads.reshape(-1), off, k, export=False)
                sz = g.sizes()
                assert (sz["n_nodes"], sz["n_kmer_instances"], sz["n_edge_instances"]) == (want["n_nodes"], want["n_kmer_instances"], want["n_edge_instances"]), "build totals"
            elif op == "multipass":
                kk = min(k, 31)
                g.build_multipass(kk, int(rng.choice([1, 2, 8, 64])))
                want = orc_c.build(reads.reshape(-1), off, kk, export=False)
                assert g.sizes()["n_nodes"] == want["n_nodes"], "multipass totals"
                for p in range(g.part_count()):
                    g.export_part(p)
            elif op == "shard1":
                dist = inproc_dist.InProcDist(inproc_dist._World(1), 0)
                multi_gpu.sharded_build(g, k, dist)
                want = orc_c.build(reads.reshape(-1), off, k, export=False)
                assert g.sizes()["n_nodes"] == want["n_nodes"], "shard totals"
            elif op == "path":
                g.build(k)
                a = path(g)
                h = _dbg.Graph()
                h.set_reads(reads.reshape(-1), off)
                h.build(k)
                b = path(h)
                h.close()
                assert a == b, "path differs from a fresh handle"
            elif op == "export":
                g.build(k)
                g.export_nodes(); g.export_succ(); g.export_csr()
                g.refine_edge_order(); g.export_orders()
            elif op == "refused":
                g.build_multipass(min(k, 31), 4)
                for fn in (lambda: g.prune(2), g.export_nodes, g.remove_tips, lambda: g.walk(False)):
                    try:
                        fn()
                        raise AssertionError("a multi-pass graph must refuse the traversal calls")
                    except _dbg.DbgError:
                        pass
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"sequence {sq} FAILED ({type(e).__name__}: {str(e)[:160]}): L={L} n={reads.shape[0]} ops={log}", flush=True)
    g.close()
    if (sq + 1) % 10 == 0:
        print(f".. {sq + 1} sequences, {bad} bad", flush=True)
print(f"soak_api: {n_seq} sequences, {bad} failures")
sys.exit(1 if bad else 0)
