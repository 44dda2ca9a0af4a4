"""Traversal of a graph in parts (part_traversal.py over the dbg_part_* primitives): branch_kmer, already_pull_out,
pull_out_read and the contig index must equal the single-GPU path on the same reads -- for a multi-pass build on one
handle and for the ranks of a sharded build (8 handles on cuda:0, in-process exchange) -- without ever gathering the graph."""
import numpy as np
import pytest

import _dbg
import synth

pytestmark = pytest.mark.gpu


def single_gpu_reference(reads, read_len, k, threshold):
    """The same path on one handle (the kernels the golden vectors pin): branch list, pulled list, pull-out reads, contig index."""
    g = _dbg.Graph()
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64))
    g.build(k)
    g.refine_edge_order()
    g.prune(threshold)
    g.remove_tips()
    g.mark_pull_reads()
    keys, stamps, _, flags = g.export_nodes(counts=False)
    hi = g.export_keys_hi()
    br = np.nonzero(flags & _dbg.F_BRANCH)[0]
    br = br[np.argsort(stamps[br], kind="stable")]
    ranks = g.export_pull_ranks()
    pu = np.nonzero(flags & _dbg.F_PULLED)[0]
    pu = pu[np.argsort(ranks[pu], kind="stable")]
    read_flags = g.export_pull_reads()
    g.set_option("walk_jump_min_nodes", 0)   # the list-ranking walk: index only needs no text
    g.walk(False, 1)
    off, score, stamp, seq = g.export_contig_index()
    o = np.lexsort((seq, stamp))
    out = {"branch": (keys[br], hi[br]), "pulled": (keys[pu], hi[pu]), "read_flags": read_flags,
           "contigs": (stamp[o], (off[1:] - off[:-1])[o].astype(np.int64), score[o].astype(np.int64)),
           "tip_rounds": g.sizes()["tip_rounds"], "n_pulled": int(pu.size)}
    g.close()
    return out


def check(res, want, read_flags):
    assert np.array_equal(res["branch"]["keys"].astype(np.uint64), want["branch"][0])
    assert np.array_equal(res["branch"]["keys_hi"].astype(np.uint64), want["branch"][1])
    assert np.array_equal(res["pulled"]["keys"], want["pulled"][0]) and np.array_equal(res["pulled"]["keys_hi"], want["pulled"][1])
    assert np.array_equal(read_flags, want["read_flags"])
    st, ln, sc = want["contigs"]
    assert np.array_equal(res["contigs"]["stamp"], st)
    assert np.array_equal(res["contigs"]["length"], ln) and np.array_equal(res["contigs"]["score"], sc)


@pytest.mark.parametrize("k,n_reads,read_len,n_passes,err", [(21, 6000, 100, 4, 0.01), (31, 8000, 150, 4, 0.01), (63, 6000, 150, 4, 0.01),
                                                            (21, 6000, 100, 1, 0.02), (31, 6000, 150, 8, 0.02), (40, 4000, 120, 2, 0.01)])
def test_multipass_traversal_equals_the_single_gpu_path(k, n_reads, read_len, n_passes, err):
    import part_traversal
    reads = synth.reads_ascii(21, n_reads * read_len // 20, n_reads, read_len, err)
    want = single_gpu_reference(reads, read_len, k, 2)
    assert want["branch"][0].size > 0 and want["n_pulled"] > 0 and want["contigs"][0].size > 0
    g = _dbg.Graph()
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64))
    g.build_multipass(k, n_passes)
    res = part_traversal.traverse(g, k, 2)
    check(res, want, res["read_flags"])
    g.close()


@pytest.mark.parametrize("k,n_reads,read_len,ranks,n_passes", [(31, 8000, 150, 8, 1), (21, 8000, 100, 4, 2), (63, 8000, 150, 8, 1), (31, 6000, 150, 2, 4)])
def test_sharded_traversal_equals_the_single_gpu_path(k, n_reads, read_len, ranks, n_passes):
    """ranks handles on cuda:0, one thread each, the real multi_gpu.sharded_build_multipass and part_traversal with an
    in-process exchange: no rank ever holds another rank's nodes, the results are those of one GPU on all reads."""
    import inproc_dist
    import multi_gpu
    import part_traversal
    per = n_reads // ranks

    def rank_reads(r):
        return synth.reads_ascii(22, n_reads * read_len // 20, per, read_len, 0.01, first_read=r * per)

    allr = np.concatenate([rank_reads(r) for r in range(ranks)])
    want = single_gpu_reference(allr, read_len, k, 2)
    assert want["branch"][0].size > 0 and want["n_pulled"] > 0

    def one(dist, rank):
        reads = rank_reads(rank)
        g = _dbg.Graph(device=0)
        g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64))
        multi_gpu.sharded_build_multipass(g, k, dist, n_passes)
        res = part_traversal.traverse(g, k, 2, dist)
        g.close()
        return res

    got = inproc_dist.run_ranks(ranks, one)
    flags = np.concatenate([r["read_flags"] for r in got])   # rank-major == the order of the concatenated reads
    for r in got:                                            # every rank ends with the same global lists
        check(r, want, flags)


def test_traversal_refuses_what_it_cannot_do():
    import part_traversal
    reads = synth.reads_ascii(23, 3000, 300, 80, 0.01)
    g = _dbg.Graph()
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, 80, dtype=np.uint64))
    g.build(21)
    with pytest.raises(_dbg.DbgError):
        part_traversal.PartTraversal(g, 21)          # a single-piece graph has no parts
    g.build_multipass(21, 2)
    t = part_traversal.PartTraversal(g, 21)
    with pytest.raises(ValueError):
        t.prune(0.5)                                  # below 1 the kept successor depends on the Counter order
    with pytest.raises(_dbg.DbgError):
        g.part_select(0, 0x20, 0x20)                  # before dbg_part_prune
    g.close()


def rewalk(g, k, starts_v, starts_local, max_steps=100000):
    """Independent restatement of one contig walk per start (debruijn.py:288-316, non-final) over the parts' rows
    (dbg_part_gather only): follows (virtual shard, local id) step by step.  -> (emits, length, score) numpy arrays."""
    import torch
    n = len(starts_v)
    v = np.asarray(starts_v, dtype=np.int64).copy()
    loc = np.asarray(starts_local, dtype=np.int64).copy()
    hops = np.zeros(n, dtype=np.int64)
    score = np.zeros(n, dtype=np.int64)
    emits = np.zeros(n, dtype=bool)
    active = np.ones(n, dtype=bool)
    seen = [set() for _ in range(n)]
    last_cnt = np.zeros(n, dtype=np.int64)
    for _ in range(max_steps):
        if not active.any():
            break
        rows = {}
        for p in np.unique(v[active]):
            sel = np.nonzero(active & (v == p))[0]
            ids = torch.from_numpy(loc[sel].astype(np.uint32).view(np.int32).copy()).cuda()
            r = g.part_gather(int(p), ids, what=("counts", "succ_owner", "succ_local", "pflags"))
            rows[int(p)] = (sel, r["counts"].cpu().numpy().view(np.uint32).reshape(-1, 4), r["succ_owner"].cpu().numpy().reshape(-1, 4),
                            r["succ_local"].cpu().numpy().view(np.uint32).reshape(-1, 4), r["pflags"].cpu().numpy())
        for p, (sel, cnt, so, sl, pf) in rows.items():
            for j, i in enumerate(sel):
                node = (int(v[i]), int(loc[i]))
                if node in seen[i]:                     # current in vec: nothing is emitted
                    active[i] = False
                    emits[i] = False
                    continue
                if pf[j] & 0x40:                        # current in already_pull_out: the path ended at the previous node
                    active[i] = False
                    emits[i] = len(seen[i]) > 0         # len(vec) == 1 (the start itself is pulled): nothing
                    if emits[i]:
                        hops[i] -= 1                    # ... so the edge that entered it does not count
                        score[i] -= last_cnt[i]
                    continue
                seen[i].add(node)
                keep = (pf[j] >> 1) & 15
                if (pf[j] & 0x20) or keep == 0:         # branch node, or no successor left
                    active[i] = False
                    emits[i] = True
                    continue
                b = (int(keep) & -int(keep)).bit_length() - 1
                last_cnt[i] = int(cnt[j, b])
                hops[i] += 1
                score[i] += int(cnt[j, b])
                v[i], loc[i] = int(so[j, b]), int(sl[j, b])
    assert not active.any(), "a sampled walk did not end"
    return emits, hops + k, score



@pytest.mark.parametrize("k,n_passes", [(31, 4)])
def test_rewalk_of_sampled_starts_equals_the_index(k, n_passes):
    """The contig index of the segment skeleton against an independent step-by-step walk over dbg_part_gather rows."""
    import part_traversal
    import torch
    reads = synth.reads_ascii(29, 40000, 8000, 150, 0.01)
    g = _dbg.Graph()
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, 150, dtype=np.uint64))
    g.build_multipass(k, n_passes)
    t = part_traversal.PartTraversal(g, k)
    t.prune(2); t.pull_out_reads(); t.remove_tips()
    idx = t.walk_index()
    by_stamp = {int(s): (int(l), int(c)) for s, l, c in zip(idx["stamp"], idx["length"], idx["score"])}
    rng = np.random.default_rng(3)
    sv, sl, ss = [], [], []
    for p in range(n_passes):
        ids = g.part_select(p, 0x01, 0)
        pick = ids[torch.from_numpy(rng.choice(ids.numel(), size=min(60, ids.numel()), replace=False)).cuda()].contiguous()
        st = g.part_gather(p, pick, what=("stamps",))["stamps"].cpu().numpy()
        sv += [p] * pick.numel(); sl += (pick.cpu().numpy().view(np.uint32)).tolist(); ss += st.tolist()
    emits, length, score = rewalk(g, k, sv, sl)
    assert emits.sum() > 100
    for e, l, c, s in zip(emits, length, score, ss):
        assert (int(s) in by_stamp) == bool(e)
        if e:
            assert by_stamp[int(s)] == (int(l), int(c))
    g.close()


def test_traversal_in_parts_at_the_baseline_size():
    """BASELINE.json configs[1] (10 M x 150 bp, k = 31, 1 % substitutions): the graph built in four parts and traversed
    part by part equals the single-GPU path -- branch list, pulled list, pull-out reads, and the 3 million contigs' (start
    stamp, length, score) -- without the four parts ever being one graph."""
    import part_traversal
    n, L, k = 10_000_000, 150, 31
    g1 = _dbg.Graph()
    g1.synth_reads(1, n * L // 30, n, L, 0.01)
    g1.build(k); g1.refine_edge_order(); g1.prune(2); g1.remove_tips(); g1.mark_pull_reads()
    sz = g1.sizes()
    br_rows, br_keys, _ = g1.export_marked(_dbg.F_BRANCH)
    pu_rows, pu_keys, _ = g1.export_marked(_dbg.F_PULLED)
    rf = g1.export_pull_reads()
    g1.walk(False, 1)
    off, score, stamp, seq = g1.export_contig_index()
    o = np.lexsort((seq, stamp))
    want = (stamp[o], (off[1:] - off[:-1])[o].astype(np.int64), score[o].astype(np.int64))
    bases, offsets = g1.copy_reads()
    g1.close()
    g = _dbg.Graph()
    g.set_reads(bases, offsets)
    g.build_multipass(k, 4)
    res = part_traversal.traverse(g, k, 2)
    assert sz["n_branch"] == res["branch"]["keys"].size and np.array_equal(res["branch"]["keys"].astype(np.uint64), br_keys)
    assert sz["n_pulled"] == res["pulled"]["keys"].size and np.array_equal(res["pulled"]["keys"], pu_keys)
    assert np.array_equal(res["read_flags"], rf)
    assert np.array_equal(res["contigs"]["stamp"], want[0])
    assert np.array_equal(res["contigs"]["length"], want[1]) and np.array_equal(res["contigs"]["score"], want[2])
    g.close()


def test_traversal_of_more_than_two_to_the_32_nodes():
    """BASELINE.json configs[3]'s per-rank size on one GPU: 4.4e9 nodes in 8 parts (45 M x 150 bp, 5 % substitutions).
    The whole path in parts: pruningEdges (1.5e7 branch nodes), pull-out reads (6.75 GB of reads against their k-mers), tip
    removal on the collected neighbourhoods (1.9e6 nodes pulled in 8 reservation rounds) and the contig index through the
    segment skeleton (3.5e8 entries ranked on the device); the index is checked against an independent step-by-step walk of
    sampled starts and through properties that hold at any size.  About a minute with the build."""
    import gc
    import part_traversal
    import torch
    torch.zeros(1, device="cuda")
    gc.collect()                # the build below needs most of the card: handles earlier tests dropped, and the blocks torch's
    torch.cuda.empty_cache()    # caching allocator still holds for them, go back to the driver first
    n, L, k, G = 45_000_000, 150, 31, 225_000_000
    g = _dbg.Graph()
    g.synth_reads(1, G, n, L, 0.05)
    g.build_multipass(k, 8)
    sz = g.sizes()
    assert sz["n_nodes"] > (1 << 32)
    t = part_traversal.PartTraversal(g, k)
    branch = t.prune(2)
    n_branch = branch["gid"].size
    assert n_branch > 0
    # every branch node keeps at least two successors, each at least half as frequent as the most frequent one (threshold 2)
    keep = (branch["pflags"][:, None] >> (1 + np.arange(4)[None, :])) & 1
    assert np.all(keep.sum(axis=1) >= 2)
    mx = branch["counts"].max(axis=1)
    assert np.all(np.where(keep == 1, 2 * branch["counts"] >= mx[:, None], True))
    flags = t.pull_out_reads()
    assert flags.size == n and 0 < int(flags.sum()) < n
    pulled = t.remove_tips()
    assert pulled["gid"].size > 0 and np.unique(pulled["gid"]).size == pulled["gid"].size
    assert np.intersect1d(pulled["gid"], branch["gid"]).size == 0          # a branch node is never pulled (debruijn.py:251)
    for p in range(8):                                                      # the pulled nodes carry the flag in their parts
        mine = pulled["gid"][(pulled["gid"] >> 32) == p] & 0xFFFFFFFF
        got = g.part_select(p, 0x40, 0x40).cpu().numpy().view(np.uint32)
        assert np.array_equal(np.sort(mine.astype(np.uint32)), got)
    idx = t.walk_index()
    n_starts = sz["n_starts"]
    assert 0 < idx["stamp"].size <= n_starts
    assert np.all(idx["length"] >= k) and np.all(idx["score"] >= 0) and np.all(idx["stamp"][1:] > idx["stamp"][:-1])
    assert np.all((idx["stamp"] & np.uint64(1)) == 0) and np.all((idx["stamp"] >> np.uint64(1)) % np.uint64(L) == 0)  # starts: position 0 of a read
    by_stamp = dict(zip(idx["stamp"].tolist(), zip(idx["length"].tolist(), idx["score"].tolist())))
    rng = np.random.default_rng(5)
    sv, sl, ss = [], [], []
    for p in range(8):
        ids = g.part_select(p, 0x01, 0)
        pick = ids[torch.from_numpy(rng.choice(ids.numel(), size=40, replace=False)).cuda()].contiguous()
        st = g.part_gather(p, pick, what=("stamps",))["stamps"].cpu().numpy()
        sv += [p] * pick.numel(); sl += (pick.cpu().numpy().view(np.uint32)).tolist(); ss += st.tolist()
        del ids
    emits, length, score = rewalk(g, k, sv, sl)
    assert emits.sum() > 100
    for e, l, c, s in zip(emits, length, score, ss):
        assert (int(s) in by_stamp) == bool(e)
        if e:
            assert by_stamp[int(s)] == (int(l), int(c))
    g.close()


@pytest.mark.parametrize("k,n_passes,ranks", [(21, 4, 1), (63, 2, 1), (31, 1, 4)])
def test_contig_texts_from_the_parts_equal_the_single_gpu_contigs(k, n_passes, ranks):
    """Identical contig sets: every contig spelled from the parts (dbg_part_segment_text along the segment skeleton) equals
    the text of the single-GPU walk, contig by contig in dict order of the starts."""
    import contextlib
    import io
    import debruijn as prod
    import inproc_dist
    import multi_gpu
    import part_traversal
    n_reads, L = 4000, 120
    per = n_reads // ranks
    reads_of = lambda r: synth.reads_ascii(31, n_reads * L // 20, per, L, 0.01, first_read=r * per)  # noqa: E731
    allr = np.concatenate([reads_of(r) for r in range(ranks)])
    with contextlib.redirect_stdout(io.StringIO()):
        gg, pull, branch, pulled, ect = prod.construct_graph([row.tobytes().decode() for row in allr], k, threshold=2)
        wc = prod.output_contigs(gg, branch, pulled)
        want, want_scores = list(wc), prod.get_score_device(wc)
    assert len(want) > 50

    def one(dist, rank):
        reads = reads_of(rank)
        g = _dbg.Graph(device=0)
        g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, L, dtype=np.uint64))
        if dist is None:
            g.build_multipass(k, n_passes)
        else:
            multi_gpu.sharded_build_multipass(g, k, dist, n_passes)
        t, flags, br, pu = part_traversal.construct_graph(g, k, 2, dist)      # the reference-shaped surface
        ctg = part_traversal.output_contigs(t)
        texts = ctg.texts(range(len(ctg)))
        assert ctg[0] == texts[0] and ctg[-1] == texts[-1]
        out = (ctg.lengths.tolist(), ctg.scores.tolist(), texts, br, pu, flags)
        g.close()
        return out

    got = [one(None, 0)] if ranks == 1 else inproc_dist.run_ranks(ranks, one)
    want_pull = [row.tobytes().decode() for row in allr[np.concatenate([r[5] for r in got]).astype(bool)]]
    assert want_pull == list(pull)
    for lengths, scores, texts, br, pu, _ in got:
        assert texts == want and [len(x) for x in texts] == lengths
        assert scores == want_scores
        assert br == list(branch) and pu == list(pulled)
