"""Host logic of the drop-in (CPU only): the read-only Mapping views that replace the dicts for large graphs.

The arrays a GPU build would export are derived here from the oracle, put into a shuffled "table order" next to the
dict-order permutation (what dbg_export_dict_order returns), and the views over them must equal the oracle's dicts --
order, lookups, membership -- without ever seeing the arrays in dict order."""
import contextlib
import io
from collections import Counter

import numpy as np
import pytest

import _dbg
import debruijn as prod
from conftest import case_reads, load_golden
from oracle import dbg_oracle as orc
from oracle import orc_c

CODE_CHAR = "ACTG"  # code -> base of the 2-bit path


def pack(reads):
    blob = "".join(reads).encode("ascii")
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    return np.frombuffer(blob, dtype=np.uint8), off


@pytest.mark.parametrize("name", ["dna_small_k5_e1_t2", "dna_small_k9_e1_t1", "dna_med_k21_e1_t2", "dna_small_k33_e1_t2",
                                  "hand_tips_order_t2_nonfinal", "hand_cycle_rho_t1_nonfinal"])
def test_views_over_table_order_arrays_equal_the_oracle_dicts(name):
    case = load_golden(name)
    reads, inp = case_reads(case), case["inputs"]
    k, thr = inp["k"], inp["threshold"]
    with contextlib.redirect_stdout(io.StringIO()):
        V0, E0 = orc.graph_from_reads(list(reads), k)                       # unpruned: successor lists with repeats
        (V, E), pull, branch, pulled, ect = orc.construct_graph(list(reads), k, threshold=thr, final=False)
    res = orc_c.build(*pack(reads), k)
    n = res["n_nodes"]
    labels = list(V.keys())
    assert n == len(labels)
    code = {c: i for i, c in enumerate(CODE_CHAR)}
    rank_mc = np.full((n, 4), 0, dtype=np.uint8)
    rank_fs = np.full((n, 4), 0, dtype=np.uint8)
    keep = np.zeros(n, dtype=np.uint32)
    flags = (res["stamps"] & np.uint64(1)).astype(np.uint8)
    pulled_set = set(pulled)
    for i, lab in enumerate(labels):
        cnt = Counter(E0.get(lab, []))
        mc = [code[s[-1]] for s, _ in cnt.most_common()]
        fs = [code[s[-1]] for s in cnt]
        rank_mc[i] = mc + [c for c in range(4) if c not in mc]              # DNA ranks all four codes
        rank_fs[i] = fs + [c for c in range(4) if c not in fs]
        if lab in pulled_set:
            flags[i] |= _dbg.F_PULLED
        else:
            for s in E[lab]:
                keep[i] |= 1 << code[s[-1]]
    rng = np.random.default_rng(3)
    perm = rng.permutation(n)                                               # table row r holds dict-order node perm[r]
    order = np.argsort(res["stamps"][perm], kind="stable").astype(np.int64)  # == dbg_export_dict_order
    wide = 2 * k > 64
    loads = []

    def loader():  # what construct_graph hands the store: the arrays leave the device on first use, once
        loads.append(1)
        return {"order": order, "keys": res["keys"][perm], "keys_hi": res["keys_hi"][perm] if wide else None,
                "counts": res["counts"][perm], "rank_mc": rank_mc[perm], "rank_fs": rank_fs[perm], "flags": flags[perm],
                "keep": keep[perm]}

    store = prod._NodeStore(k, CODE_CHAR.encode(), 2, n, loader)
    assert len(prod._LazyVertices(store)) == n and not loads  # the size is known without the arrays
    Vv, Ev, Cv = prod._LazyVertices(store), prod._LazyEdges(store), prod._LazyEdgeCounts(store)
    assert list(Vv) == labels and len(Vv) == n
    assert [(Vv[v].indegree, Vv[v].outdegree) for v in labels] == [(V[v].indegree, V[v].outdegree) for v in labels]
    assert list(Ev) == list(E) and dict(Ev) == E and len(Ev) == len(E)
    assert list(Cv.items()) == list(ect.items()) and len(Cv) == len(ect)
    for v in labels[:40]:
        assert v in Vv and (v in Ev) == (v in E)
    for bogus in ("", "N" * k, "A" * (k + 1), None, 3):
        assert bogus not in Vv and bogus not in Ev and bogus not in Cv
    some_edge = next(iter(ect))
    assert Cv[some_edge] == ect[some_edge] and (some_edge[:-1] + "N") not in Cv
    assert loads == [1]  # one load, however much was asked


def test_lazy_contigs_sequence_protocol():
    """LazyContigs fetches texts by index in the reference's order; here against a stand-in for the device handle."""
    texts = {0: b"ACGT", 1: b"TTTTT", 2: b"GG"}

    class Handle:
        def export_contig_text(self, index, length):
            assert length == len(texts[index])
            return texts[index]

    off = np.array([0, 4, 9, 11], dtype=np.uint64)
    score = np.array([7, 9, 1], dtype=np.uint64)
    order = np.array([2, 0, 1])
    lazy = prod.LazyContigs(Handle(), order, off, score)
    assert len(lazy) == 3 and list(lazy) == ["GG", "ACGT", "TTTTT"] and lazy.scores == [1, 7, 9] and lazy.lengths == [2, 4, 5]
    assert lazy[-1] == "TTTTT" and lazy[0:2] == ["GG", "ACGT"] and "ACGT" in lazy
    with pytest.raises(IndexError):
        lazy[3]


def test_lazy_contigs_survive_the_reference_driver_lines():
    """II_assembleFromReads.py:64 and :74 on a LazyContigs: ``sequences.sort(key=lambda x: getScore(...), reverse=True)``
    permutes by the device scores exactly as list.sort would (stable), ``sequences.extend(pull_out_read)`` appends."""
    rng = np.random.default_rng(3)
    n = 40
    texts = {i: bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(12, 24)))) for i in range(n)}
    lens = np.array([len(texts[i]) for i in range(n)], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    score = rng.integers(0, 6, size=n).astype(np.uint64)  # many ties: stability matters

    class Handle:
        def export_contig_text(self, index, length):
            return texts[index]

    order = rng.permutation(n)
    table = {texts[i].decode(): int(score[i]) for i in range(n)}  # stand-in for getScore(edge_count_table, x, k)
    if len(table) < n:  # duplicate texts would make the stand-in ambiguous
        pytest.skip("random texts collided")
    lazy = prod.LazyContigs(Handle(), order, off, score)
    want = list(lazy)
    want.sort(key=lambda x: table[x], reverse=True)
    lazy.sort(key=lambda x: table[x], reverse=True)
    assert list(lazy) == want and lazy.scores == [table[x] for x in want] and lazy.lengths == [len(x) for x in want]
    lazy.sort(key=lambda x: table[x])
    want.sort(key=lambda x: table[x])
    assert list(lazy) == want
    with pytest.raises(NotImplementedError):
        lazy.sort(key=len)  # not the contig score: would need every text
    lazy.extend(["TTT", "GGGG"])
    assert len(lazy) == n + 2 and lazy[-1] == "GGGG" and lazy[n] == "TTT" and list(lazy)[:n] == want


@pytest.mark.parametrize("name", ["peptide_k3_nonfinal", "peptide_k4_nonfinal"])
def test_views_for_a_generic_alphabet(name):
    """The same for the generic engine's layout: 5 bits per character, [n][32] count and rank arrays, 0xFF beyond the
    out-degree."""
    case = load_golden(name)
    reads, inp = case_reads(case), case["inputs"]
    k, thr = inp["k"], inp["threshold"]
    with contextlib.redirect_stdout(io.StringIO()):
        V0, E0 = orc.graph_from_reads(list(reads), k)
        (V, E), pull, branch, pulled, ect = orc.construct_graph(list(reads), k, threshold=thr, final=False)
    alphabet = "".join(sorted(set("".join(reads))))  # codes in byte order, like dbg_get_alphabet
    code = {c: i for i, c in enumerate(alphabet)}
    labels = list(V.keys())
    n = len(labels)
    keys = np.zeros(n, dtype=np.uint64)
    counts = np.zeros((n, 32), dtype=np.uint32)
    rank_mc = np.full((n, 32), 0xFF, dtype=np.uint8)
    rank_fs = np.full((n, 32), 0xFF, dtype=np.uint8)
    keep = np.zeros(n, dtype=np.uint32)
    flags = np.array([V[v].indegree for v in labels], dtype=np.uint8)
    pulled_set = set(pulled)
    for i, lab in enumerate(labels):
        v = 0
        for ch in lab:
            v = (v << 5) | code[ch]
        keys[i] = v
        cnt = Counter(E0.get(lab, []))
        for s, c in cnt.items():
            counts[i, code[s[-1]]] = c
        mc = [code[s[-1]] for s, _ in cnt.most_common()]
        fs = [code[s[-1]] for s in cnt]
        rank_mc[i, :len(mc)] = mc
        rank_fs[i, :len(fs)] = fs
        if lab in pulled_set:
            flags[i] |= _dbg.F_PULLED
        else:
            for s in E[lab]:
                keep[i] |= 1 << code[s[-1]]
    take = np.random.default_rng(5).permutation(n)  # table row r holds dict-order node take[r]
    order = np.argsort(take).astype(np.int64)        # row of dict-order node i (what dbg_export_dict_order returns)
    store = prod._NodeStore(k, alphabet.encode(), 5, n, lambda: {"order": order, "keys": keys[take], "keys_hi": None,
                                                                "counts": counts[take], "rank_mc": rank_mc[take],
                                                                "rank_fs": rank_fs[take], "flags": flags[take], "keep": keep[take]})
    Vv, Ev, Cv = prod._LazyVertices(store), prod._LazyEdges(store), prod._LazyEdgeCounts(store)
    assert list(Vv) == labels
    assert [(Vv[v].indegree, Vv[v].outdegree) for v in labels] == [(V[v].indegree, V[v].outdegree) for v in labels]
    assert dict(Ev) == E and list(Ev) == list(E)
    assert list(Cv.items()) == list(ect.items())
    assert ("?" * k) not in Vv and (labels[0] + "?") not in Cv


def test_support_score_table_cache_follows_the_dict_and_its_content():
    """IV_sortOutputs._pack_table (no GPU needed): the packed table is reused only for the SAME dict with the SAME items.
    A temporary dict of equal length (CPython hands its id to the next one), an in-place update and an int score
    replaced by the equal float must all be packed afresh -- the reference reads the dict on every call."""
    import IV_sortOutputs as iv

    def values(t):
        return iv._pack_table(t)[2].tolist(), iv._pack_table(t)[3].tolist()

    seen = []
    for a, b in ((1.5, 2.5), (10.0, 20.0), (7.0, 8.0)):   # same length, each freed before the next is made
        seen.append(values({"ACGT": a, "TTGA": b})[0])
    assert seen == [[1.5, 2.5], [10.0, 20.0], [7.0, 8.0]]
    t = {"ACGT": 1.5, "TTGA": 2.5}
    first = iv._pack_table(t)
    assert iv._pack_table(t) is first                      # unchanged: reused
    t["TTGA"] = 4.0                                        # value updated in place
    assert values(t)[0] == [1.5, 4.0]
    t["GGG"] = 1                                           # key added
    assert values(t) == ([1.5, 4.0, 1.0], [1, 1, 0])
    t["GGG"] = 1.0                                         # 1 == 1.0 and hash alike; the result type differs
    assert values(t) == ([1.5, 4.0, 1.0], [1, 1, 1])


def test_skeleton_ranking_equals_a_step_by_step_walk():
    """part_traversal.rank_skeleton (pointer jumping over the segment skeleton of a graph in parts; no GPU needed) against a
    plain walk from every start, on random skeletons with chains, pulled entries, cycles inside a part and across parts."""
    import torch
    import part_traversal as pt
    rng = np.random.default_rng(11)
    for trial in range(60):
        n = int(rng.integers(1, 400))
        gid = np.sort(rng.choice(1 << 20, size=n, replace=False)).astype(np.int64) | (rng.integers(0, 8, size=n).astype(np.int64) << 32)
        gid = np.unique(gid)
        n = gid.size
        kind = rng.choice([pt.K_EMIT, pt.K_PULLED, pt.K_REMOTE, pt.K_REMOTE, pt.K_REMOTE, pt.K_CYCLE, pt.K_NEXT_PULLED], size=n)
        nxt = gid[rng.integers(0, n, size=n)]                       # any entry, cycles included
        hops = rng.integers(0, 30, size=n).astype(np.int64)
        score = rng.integers(0, 1000, size=n).astype(np.int64)
        exit_cnt = rng.integers(1, 50, size=n).astype(np.int64)
        start = (rng.random(n) < 0.3).astype(np.int64)
        stamp = rng.permutation(n).astype(np.int64) * 2
        perm = rng.permutation(n)                                    # rows arrive in any order
        e = {"gid": gid[perm], "kind": kind[perm].astype(np.int64), "next": nxt[perm], "hops": hops[perm], "score": score[perm],
             "exit": exit_cnt[perm], "stamp": stamp[perm], "start": start[perm]}
        got, sk = pt.rank_skeleton({c: torch.from_numpy(v.copy()) for c, v in e.items()}, 21, keep=True)
        pos = {int(g_): i for i, g_ in enumerate(gid)}
        want = []
        for s in np.nonzero(start)[0]:
            if kind[s] == pt.K_PULLED:
                continue
            i, h, sc, seen, ok = int(s), 0, 0, set(), True
            while True:
                if i in seen or kind[i] == pt.K_CYCLE:
                    ok = False
                    break
                seen.add(i)
                h += int(hops[i]); sc += int(score[i])
                if kind[i] != pt.K_REMOTE:
                    break
                j = pos[int(nxt[i])]
                if kind[j] == pt.K_PULLED:
                    break
                h += 1; sc += int(exit_cnt[i])
                i = j
            if ok:
                want.append((int(stamp[s]), h + 21, sc))
        want.sort()
        assert list(zip(got["stamp"].tolist(), got["length"].tolist(), got["score"].tolist())) == want
        assert sk["emit"].size == len(want)
