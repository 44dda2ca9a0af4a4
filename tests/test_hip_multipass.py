"""Multi-pass build (BASELINE.json configs[3]: hash-prefix passes, finished parts parked in HBM, node ids beyond 2^32).

Small sizes: every part export, gathered, equals the C oracle and every successor (part, local id) points at the
shifted k-mer.  Full size: one GPU, more than 2^32 nodes across the parts, checked through properties that hold
whatever the size."""
import numpy as np
import pytest

import _dbg
import synth
from oracle import orc_c

pytestmark = pytest.mark.gpu


def gather_parts(g):
    parts = [g.export_part(p) for p in range(g.part_count())]
    for p, d in enumerate(parts):
        assert g.part_sizes(p)["n_nodes"] == d["keys"].size and int(d["row_ptr"][-1]) == d["col"].size
    return parts


def dense_counts(d):
    """[n, 4] counts by base code from a part's CSR: the columns of a row follow the base codes set in flags."""
    n = d["keys"].size
    counts = np.zeros((n, 4), dtype=np.uint32)
    e = d["row_ptr"][:-1].astype(np.int64).copy()
    for code in range(4):
        has = ((d["flags"] >> (1 + code)) & 1).astype(bool)
        counts[has, code] = d["cnt"][e[has]]
        e[has] += 1
    assert np.array_equal(e, d["row_ptr"][1:].astype(np.int64))
    return counts


def shifted(keys, keys_hi, code, k):
    """(lo, hi) of the successor k-mers: the k-mer shifted by one base `code` (two words for k > 31)."""
    u64 = np.uint64
    lo_mask = u64((1 << (2 * k)) - 1) if 2 * k < 64 else u64(0xFFFFFFFFFFFFFFFF)
    hi_mask = u64((1 << (2 * k - 64)) - 1) if 2 * k > 64 else u64(0)
    return ((keys << u64(2)) | u64(code)) & lo_mask, ((keys_hi << u64(2)) | (keys >> u64(62))) & hi_mask


def check_against_oracle(g, reads, read_len, k):
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64), k)
    parts = gather_parts(g)
    keys = np.concatenate([d["keys"] for d in parts])
    stamps = np.concatenate([d["stamps"] for d in parts])
    counts = np.concatenate([dense_counts(d) for d in parts])
    flags = np.concatenate([d["flags"] for d in parts])
    sz = g.sizes()
    assert keys.size == want["n_nodes"] == sz["n_nodes"] and sz["n_edges"] == int((want["counts"] != 0).sum())
    assert sz["n_kmer_instances"] == want["n_kmer_instances"] and sz["n_edge_instances"] == want["n_edge_instances"]
    o = np.argsort(stamps, kind="stable")
    keys_hi = np.concatenate([d["keys_hi"] for d in parts])
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(stamps[o], want["stamps"])
    assert np.array_equal(keys_hi[o], want["keys_hi"])
    assert np.array_equal(counts[o], want["counts"])
    assert np.array_equal(flags & 1, (stamps & np.uint64(1)).astype(np.uint8))
    assert sz["n_starts"] == int(((want["stamps"] & np.uint64(1)) == 0).sum())
    check_successors(parts, k)


def check_successors(parts, k, n_passes=None):
    """every successor (part, local id) -> the shifted k-mer; returns how many cross a rank (parts // n_passes differ)"""
    crossing = 0
    for v, d in enumerate(parts):
        e = d["row_ptr"][:-1].astype(np.int64).copy()
        for code in range(4):
            has = ((d["flags"] >> (1 + code)) & 1).astype(bool)
            cols, owners = d["col"][e[has]], d["col_part"][e[has]]
            got, got_hi = np.empty(cols.size, dtype=np.uint64), np.empty(cols.size, dtype=np.uint64)
            for q, dq in enumerate(parts):
                sel = owners == q
                assert np.all(cols[sel] < dq["keys"].size)
                got[sel] = dq["keys"][cols[sel]]
                got_hi[sel] = dq["keys_hi"][cols[sel]]
            assert owners.size == 0 or int(owners.max()) < len(parts)
            lo, hi = shifted(d["keys"][has], d["keys_hi"][has], code, k)
            assert np.array_equal(got, lo) and np.array_equal(got_hi, hi)
            if n_passes:
                crossing += int((owners // n_passes != v // n_passes).sum())
            e[has] += 1
    return crossing


@pytest.mark.parametrize("stamp64", [0, 1])
@pytest.mark.parametrize("n_passes", [1, 2, 4, 8, 64])
@pytest.mark.parametrize("k,n_reads,read_len", [(31, 6000, 150), (21, 6000, 100), (63, 6000, 150), (40, 4000, 120)])
def test_multipass_equals_the_oracle(n_passes, k, n_reads, read_len, stamp64):
    """stamp64: the 64-bit stamps that reads of 2 GiB and more get (two-word k-mers: refused before round 3)."""
    if stamp64 and n_passes in (2, 64):
        pytest.skip("a sample of the pass counts")
    reads = synth.reads_ascii(11, n_reads * read_len // 20, n_reads, read_len, 0.01)
    g = _dbg.Graph()
    g.set_option("stamp64", stamp64)
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64))
    g.build_multipass(k, n_passes)
    assert g.part_count() == n_passes
    check_against_oracle(g, reads, read_len, k)
    with pytest.raises(_dbg.DbgError, match="multi-pass"):
        g.prune(2)  # 32-bit node ids: the traversal refuses a graph in parts
    with pytest.raises(_dbg.DbgError, match="multi-pass"):
        g.export_nodes()
    g.build(k)  # the same handle builds a single-pass graph again
    assert g.part_count() == 0 and g.sizes()["n_nodes"] > 0


def test_multipass_with_forced_geometry_and_overflowing_buckets():
    """bucket_bits = 9: one final bucket per level-1 group, so tables overflow and are counted in hash sub-ranges
    (the directory then has extra ranges): successors across parts and across sub-ranges still resolve."""
    reads = synth.reads_ascii(12, 400_000, 60_000, 100, 0.01)
    g = _dbg.Graph()
    g.set_option("bucket_bits", 9)
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, 100, dtype=np.uint64))
    g.build_multipass(31, 4)
    check_against_oracle(g, reads, 100, 31)


def test_more_than_two_to_the_32_nodes_on_one_gpu():
    """45 M x 150 bp reads with 5 % substitutions: about 4.5e9 distinct 31-mers -- more than a 32-bit node id can
    name -- built in 8 passes on one GPU (6.75 GB of reads, 64-bit stamps).  No host export at this size: the
    invariants are computed on the device (torch) over the parts' arrays."""
    import torch
    torch.zeros(1, device="cuda")  # bring torch's context up before the build fills most of the HBM
    n, L, k, G = 45_000_000, 150, 31, 225_000_000
    g = _dbg.Graph()
    g.synth_reads(1, G, n, L, 0.05)
    g.build_multipass(k, 8)
    sz = g.sizes()
    assert sz["n_nodes"] > (1 << 32)
    assert sz["n_kmer_instances"] == n * (L - k + 1) and sz["n_edge_instances"] == n * (L - k)
    parts = [g.part_tensors(p) for p in range(8)]
    psz = [g.part_sizes(p) for p in range(8)]
    assert sum(s["n_nodes"] for s in psz) == sz["n_nodes"] and sum(s["n_edges"] for s in psz) == sz["n_edges"]
    assert all(s["n_nodes"] < (1 << 32) - 16 for s in psz)
    assert [s["first_node_id"] for s in psz] == [sum(x["n_nodes"] for x in psz[:p]) for p in range(8)]
    bases, _ = g.reads_tensors()
    mask = (1 << (2 * k)) - 1
    total_cnt = total_starts = 0
    gen = torch.Generator(device="cuda").manual_seed(5)
    for p, d in enumerate(parts):
        nn, ne = psz[p]["n_nodes"], psz[p]["n_edges"]
        rp = d["row_ptr"].to(torch.int64) & 0xFFFFFFFF
        assert int(rp[-1]) == ne and int(rp[0]) == 0
        deg = rp[1:] - rp[:-1]
        present = (d["flags"] >> 1) & 15
        popc = ((present & 1) + ((present >> 1) & 1) + ((present >> 2) & 1) + ((present >> 3) & 1)).to(torch.int64)
        assert bool((deg == popc).all())                                    # a row holds one column per base that occurs
        total_cnt += int((d["cnt"].to(torch.int64) & 0xFFFFFFFF).sum())
        assert bool((d["cnt"] != 0).all())
        stamps = d["stamps"].to(torch.int64)
        assert bool(((d["flags"] & 1).to(torch.int64) == (stamps & 1)).all())
        pos = stamps >> 1
        assert bool((((stamps & 1) == 0) == (pos % L == 0)).all())          # indegree 0 <=> position 0 of a read
        total_starts += int(((stamps & 1) == 0).sum())
        for q in range(8):                                                   # every successor id names a node of its part
            sel = d["col_part"] == q
            if bool(sel.any()):
                assert int((d["col"][sel].to(torch.int64) & 0xFFFFFFFF).max()) < psz[q]["n_nodes"]
        assert int(d["col_part"].max()) < 8
        # a sample of nodes: the first occurrence holds the k-mer, and every successor is the shifted k-mer
        idx = torch.randint(0, nn, (200_000,), device="cuda", generator=gen)
        window = bases[(pos[idx][:, None] + torch.arange(k, device="cuda")[None, :])].to(torch.int64)
        code = (window >> 1) & 3
        shifts = 2 * (k - 1 - torch.arange(k, device="cuda"))
        assert bool(((code << shifts[None, :]).sum(dim=1) == d["keys"][idx]).all())
        e = rp[idx].clone()
        for c in range(4):
            has = ((present[idx] >> c) & 1).bool()
            ee = e[has]
            tgt_part, tgt = d["col_part"][ee], d["col"][ee].to(torch.int64) & 0xFFFFFFFF
            got = torch.empty_like(tgt)
            for q in range(8):
                sel = tgt_part == q
                got[sel] = parts[q]["keys"][tgt[sel]]
            assert bool((got == (((d["keys"][idx][has] << 2) | c) & mask)).all())
            e[has] += 1
        del rp, deg, present, popc, stamps, pos
    assert total_cnt == sz["n_edge_instances"]                               # every (k+1)-mer instance counted once
    assert total_starts == sz["n_starts"]


def rank_reads(world, rank, n_reads, read_len, seed=31):
    per = n_reads // world
    return synth.reads_ascii(seed, max(4 * read_len, n_reads * read_len // 20), per, read_len, 0.01, first_read=rank * per)


@pytest.mark.parametrize("ranks,n_passes,k,n_reads,read_len,wide_stamp_ranks",
                         [(2, 4, 31, 8000, 150, ()), (4, 4, 21, 8000, 100, ()), (8, 8, 31, 16000, 150, ()),
                          (8, 1, 31, 8000, 150, ()), (1, 4, 31, 6000, 150, ()), (4, 2, 31, 8000, 150, (0, 3)),
                          (2, 2, 21, 6000, 100, (0, 1)),
                          # two-word k-mers (BASELINE.json configs[4]): records by value, (lo, hi) queries, owner bytes
                          (8, 1, 63, 8000, 150, ()), (4, 2, 63, 8000, 150, ()), (2, 4, 40, 6000, 120, ())])
@pytest.mark.parametrize("chunks", [1, 2, 3])
def test_ranks_times_passes_equal_the_oracle(ranks, n_passes, k, n_reads, read_len, wide_stamp_ranks, chunks):
    """BASELINE.json configs[3] in miniature: a sharded build whose ranks build their shards in passes
    (multi_gpu.sharded_build_multipass through the C ABI; `ranks` handles on cuda:0, one thread each, in-process
    exchange).  The union of all parts of all ranks == the C oracle, and every successor (virtual shard = rank * passes +
    part, local id) -- resolved inside a part, across the parts of a rank, or across ranks -- is the shifted k-mer.
    chunks > 1: every rank cuts and sends its records in that many parts (dbg_shard_extract_part; the receiver sees
    chunks x ranks senders) -- the same graph."""
    import inproc_dist
    import multi_gpu
    if chunks > 1 and (n_passes, ranks) not in ((4, 2), (1, 8), (2, 4), (4, 1)):
        pytest.skip("parts of the records: a sample of the geometries")

    def one(dist, rank):
        reads = rank_reads(ranks, rank, n_reads, read_len)
        g = _dbg.Graph(device=0)
        if rank in wide_stamp_ranks:
            g.set_option("shard_stamp64", 1)
        g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64))
        multi_gpu.sharded_build_multipass(g, k, dist, n_passes, chunks=chunks)
        assert g.part_count() == n_passes
        parts = gather_parts(g)
        sz = g.sizes()
        with pytest.raises(_dbg.DbgError, match="multi-pass"):
            g.prune(2)
        g.close()
        return parts, sz

    got = inproc_dist.run_ranks(ranks, one)
    parts = [d for rank_parts, _ in got for d in rank_parts]      # index = virtual shard
    assert len(parts) == ranks * n_passes
    all_reads = np.concatenate([rank_reads(ranks, r, n_reads, read_len) for r in range(ranks)])
    want = orc_c.build(all_reads.reshape(-1), np.arange(0, all_reads.size + 1, read_len, dtype=np.uint64), k)
    keys = np.concatenate([d["keys"] for d in parts])
    stamps = np.concatenate([d["stamps"] for d in parts])
    counts = np.concatenate([dense_counts(d) for d in parts])
    assert keys.size == want["n_nodes"] == sum(sz["n_nodes"] for _, sz in got)
    assert sum(sz["n_kmer_instances"] for _, sz in got) == want["n_kmer_instances"]
    assert sum(sz["n_edge_instances"] for _, sz in got) == want["n_edge_instances"]
    o = np.argsort(stamps, kind="stable")
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(stamps[o], want["stamps"])
    assert np.array_equal(np.concatenate([d["keys_hi"] for d in parts])[o], want["keys_hi"])
    assert np.array_equal(counts[o], want["counts"])
    crossing = check_successors(parts, k, n_passes)
    assert ranks == 1 or crossing > 0  # the exchange between ranks really carried successors


@pytest.mark.parametrize("n_reads,n_passes", [(3, 64), (40, 64), (1, 8), (0, 4)])
def test_multipass_with_empty_parts(n_reads, n_passes):
    """Fewer records than level-1 groups: most parts hold nothing (no arrays at all) -- exports, sizes and the device
    views must cope, and the non-empty parts still equal the oracle."""
    reads = synth.reads_ascii(13, 2000, max(n_reads, 1), 60, 0.0)[:n_reads]
    g = _dbg.Graph()
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, 60, dtype=np.uint64))
    g.build_multipass(31, n_passes)
    assert g.part_count() == n_passes
    empty = [p for p in range(n_passes) if g.part_sizes(p)["n_nodes"] == 0]
    assert len(empty) > 0
    for p in empty:
        d = g.export_part(p)
        assert d["keys"].size == 0 and d["row_ptr"].tolist() == [0] and d["col"].size == 0
        t = g.part_tensors(p)
        assert t["keys"].numel() == 0 and t["row_ptr"].tolist() == [0]
    if n_reads:
        check_against_oracle(g, reads, 60, 31)
    else:
        assert g.sizes()["n_nodes"] == 0


def test_ranks_times_passes_with_empty_parts():
    """4 ranks x 8 passes over 160 short reads: 32 virtual shards, some without a single record."""
    import inproc_dist
    import multi_gpu
    k, read_len, per, ranks, n_passes = 31, 34, 40, 4, 8

    def one(dist, rank):
        reads = synth.reads_ascii(862236413, 272, per, read_len, 0.01, first_read=rank * per)
        g = _dbg.Graph(device=0)
        if rank == 0:
            g.set_option("shard_stamp64", 1)
        g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64))
        multi_gpu.sharded_build_multipass(g, k, dist, n_passes)
        parts = gather_parts(g)
        g.close()
        return parts

    parts = [d for rp in inproc_dist.run_ranks(ranks, one) for d in rp]
    allr = np.concatenate([synth.reads_ascii(862236413, 272, per, read_len, 0.01, first_read=r * per) for r in range(ranks)])
    want = orc_c.build(allr.reshape(-1), np.arange(0, allr.size + 1, read_len, dtype=np.uint64), k)
    keys = np.concatenate([d["keys"] for d in parts])
    stamps = np.concatenate([d["stamps"] for d in parts])
    o = np.argsort(stamps, kind="stable")
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(stamps[o], want["stamps"])
    assert np.array_equal(np.concatenate([dense_counts(d) for d in parts])[o], want["counts"])


def test_part_entry_points_refuse_bad_arguments():
    """dbg_part_* / dbg_multipass_finish / dbg_shard_build_multipass on handles and arguments that cannot be served:
    an error code and text, never a crash."""
    import torch
    reads = synth.reads_ascii(14, 3000, 200, 80, 0.01)
    g = _dbg.Graph()
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, 80, dtype=np.uint64))
    with pytest.raises(_dbg.DbgError):
        g.multipass_finish()                      # no multi-pass build yet
    with pytest.raises(_dbg.DbgError):
        g.part_queries(0)
    g.build(31)                                   # a single-pass graph has no parts either
    assert g.part_count() == 0
    with pytest.raises(_dbg.DbgError):
        g.export_part(0)
    g.build_multipass(31, 4)
    for bad in (-1, 4, 1000):
        with pytest.raises(_dbg.DbgError):
            g.part_sizes(bad)
        with pytest.raises(_dbg.DbgError):
            g.part_queries(bad)
        with pytest.raises(_dbg.DbgError):
            g.part_answer(bad, torch.zeros(4, dtype=torch.int64, device="cuda"))
    with pytest.raises(_dbg.DbgError):
        g.part_apply(0, 99, torch.zeros(0, dtype=torch.int32, device="cuda"))   # no such owner
    g.multipass_finish()                          # nothing open on one GPU: fine, and twice as well
    g.multipass_finish()
    # ranks x passes with impossible shapes
    dummy = torch.zeros(16, dtype=torch.int64, device="cuda")
    st = torch.zeros(16, dtype=torch.int32, device="cuda")
    rows = [[0] * 256, [0] * 256]
    for n_shards, me, passes in ((2, 0, 3), (2, 0, 64), (2, 5, 2), (3, 0, 2)):
        with pytest.raises(_dbg.DbgError):
            g.shard_build_multipass(31, n_shards, me, passes, dummy, dummy, st, [0] * n_shards, [0] * n_shards,
                                    rows if n_shards == 2 else [[0] * 170] * 3 + [[0] * 2])
    with pytest.raises(_dbg.DbgError, match="4-byte stamps"):  # two-word records carry 32-bit rank-local stamps
        g.shard_build_multipass(40, 2, 0, 2, dummy, dummy, dummy, [0, 0], [0, 0], rows)
    g.build(31)                                   # the handle still works
    assert g.sizes()["n_nodes"] > 0
