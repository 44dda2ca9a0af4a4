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
