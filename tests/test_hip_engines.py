"""Node table / successor parity of the two build engines and the C oracle (GPU, through the C ABI).

engine 0: super-k-mer partitioned build (default); engine 1: single global hash table.
Bucket geometries are forced so that the one- and two-level multisplit paths and the LDS
overflow split (a bucket that does not fit the table) are all exercised at test sizes.
"""
import numpy as np
import pytest

import _dbg
import synth
from oracle import orc_c

pytestmark = pytest.mark.gpu


def table(g, k):
    keys, stamps, counts, flags = g.export_nodes()
    succ = g.export_succ()
    o = np.argsort(stamps, kind="stable")
    return keys[o], stamps[o], counts[o], flags[o], succ, keys


def check_succ(keys_raw, counts_raw, succ, k):
    mask = np.uint64((1 << (2 * k)) - 1)
    for code in range(4):
        has = counts_raw[:, code] != 0
        assert np.all(succ[~has, code] == _dbg.NO_NODE)
        assert np.all(succ[has, code] != _dbg.NO_NODE), "unresolved successor"
        want = ((keys_raw[has] << np.uint64(2)) | np.uint64(code)) & mask
        assert np.array_equal(keys_raw[succ[has, code]], want)


def build(reads2d, k, **opts):
    g = _dbg.Graph()
    for name, v in opts.items():
        g.set_option(name, v)
    L = reads2d.shape[1]
    g.set_reads(reads2d.reshape(-1), np.arange(0, reads2d.size + 1, L, dtype=np.uint64))
    g.build(k)
    return g


@pytest.mark.parametrize("k", [5, 13, 21, 31])
@pytest.mark.parametrize("opts", [
    dict(engine=1),
    dict(engine=0),
    dict(engine=0, bucket_bits=3),
    dict(engine=0, bucket_bits=9),
    dict(engine=0, bucket_bits=12, lds_slots=2048),
    dict(engine=0, bucket_bits=18),
    dict(engine=0, bucket_bits=22),                  # three multisplit levels: 10 + 10 + 2 bits, all 22 of the bucket hash
    dict(engine=0, bucket_bits=1, lds_slots=2048),   # forces LDS overflow splits
    dict(engine=0, stamp64=1),                       # 64-bit stamps (reads of 2 GiB and more, shards)
    dict(engine=0, stamp64=1, bucket_bits=3),
    dict(engine=0, stamp64=1, count_kernel_u64=1),   # ... on k_sk_count (lookups after the insert)
    dict(engine=0, stamp64=1, count_kernel_u64=3),   # ... on k_sk_count3 (one hint per slot, deferred lookups)
    dict(engine=0, stamp64=1, count_kernel_u64=3, bucket_bits=3),
    dict(engine=0, stamp64=1, count_kernel_u64=3, bucket_bits=9),
    dict(engine=0, stamp64=1, count_kernel_u64=2),   # ... on k_sk_count2 (320 staged records)
])
def test_engine_matches_c_oracle(k, opts):
    reads = synth.reads_ascii(7, 60000, 6000, 100, 0.01)
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, 100, dtype=np.uint64), k)
    g = build(reads, k, **opts)
    sz = g.sizes()
    assert sz["n_kmer_instances"] == want["n_kmer_instances"]
    assert sz["n_edge_instances"] == want["n_edge_instances"]
    assert sz["n_nodes"] == want["n_nodes"]
    keys, stamps, counts, flags, succ, keys_raw = table(g, k)
    assert np.array_equal(keys, want["keys"])
    assert np.array_equal(stamps, want["stamps"])
    assert np.array_equal(counts, want["counts"])
    assert np.array_equal(flags & 1, (want["stamps"] & np.uint64(1)).astype(np.uint8))
    _, _, counts_raw, _ = g.export_nodes()
    check_succ(keys_raw, counts_raw, succ, k)
    rp, col, cnt = g.export_csr()
    assert int(rp[-1]) == sz["n_edges"] == int((counts_raw != 0).sum())
    assert np.array_equal(col, succ[counts_raw != 0]) and np.array_equal(cnt, counts_raw[counts_raw != 0])


def test_ragged_reads_and_boundaries():
    """Variable-length reads (including len <= k and empty) packed back to back."""
    rng = np.random.default_rng(5)
    genome = synth.reads_ascii(9, 5000, 1, 5000, 0.0)[0]
    reads, lens = [], []
    for _ in range(3000):
        L = int(rng.integers(0, 90))
        s = int(rng.integers(0, 5000 - L + 1))
        reads.append(genome[s:s + L])
        lens.append(L)
    blob = np.concatenate(reads) if reads else np.zeros(0, np.uint8)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    for k in (4, 17, 31, 40, 63):   # above 31: the two-word path (the options do not apply)
        want = orc_c.build(blob, off, k)
        for opts in ((dict(engine=1), dict(engine=0), dict(engine=0, bucket_bits=7)) if k <= 31 else (dict(),)):
            g = _dbg.Graph()
            for name, v in opts.items():
                g.set_option(name, v)
            g.set_reads(blob, off)
            g.build(k)
            keys, stamps, counts, flags, succ, keys_raw = table(g, k)
            assert np.array_equal(keys, want["keys"]) and np.array_equal(stamps, want["stamps"])
            assert np.array_equal(counts, want["counts"])
            if k <= 31:
                check_succ(keys_raw, g.export_nodes()[2], succ, k)
            else:
                assert np.array_equal(g.export_keys_hi()[np.argsort(g.export_nodes()[1], kind="stable")], want["keys_hi"])


@pytest.mark.parametrize("k", [9, 31, 47])
def test_skewed_low_complexity_reads(k):
    """Homopolymers, tandem repeats and heavy duplication next to ordinary reads: a handful of buckets receive
    tens of thousands of identical records (several staging chunks, multiplicities far above the coverage)."""
    L = 120
    rng = np.random.default_rng(11)
    rows = [np.frombuffer(("A" * L).encode(), dtype=np.uint8)] * 9000
    rows += [np.frombuffer(("AC" * L)[:L].encode(), dtype=np.uint8)] * 7000
    rows += [np.frombuffer(("ACGGT" * L)[:L].encode(), dtype=np.uint8)] * 5000
    rows += [np.frombuffer(("T" * 50 + "G" * 70).encode(), dtype=np.uint8)] * 3000
    rows += list(synth.reads_ascii(13, 20000, 4000, L, 0.01))
    order = rng.permutation(len(rows))
    reads = np.stack([rows[i] for i in order])
    off = np.arange(0, reads.size + 1, L, dtype=np.uint64)
    want = orc_c.build(reads.reshape(-1), off, k)
    for opts in ((dict(engine=0), dict(engine=0, bucket_bits=2, lds_slots=2048), dict(engine=1)) if k <= 31 else (dict(),)):
        g = build(reads, k, **opts)
        keys, stamps, counts, flags, succ, keys_raw = table(g, k)
        assert np.array_equal(keys, want["keys"]) and np.array_equal(stamps, want["stamps"])
        assert np.array_equal(counts, want["counts"])
        assert int(counts.max()) >= 9000 * (L - k)
        if k <= 31:
            check_succ(keys_raw, g.export_nodes()[2], succ, k)


def test_rebuild_on_same_handle_reuses_buffers():
    reads = synth.reads_ascii(3, 30000, 3000, 100, 0.01)
    g = build(reads, 21)
    a = table(g, 21)
    g.build(31)
    g.build(21)
    b = table(g, 21)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)


def test_capacity_retry_when_the_estimate_is_low():
    """The node arrays are sized from a sampled distinct-k-mer estimate; if it is too low the count kernel says so
    and the build runs it once more with the worst-case size -- same graph."""
    reads = synth.reads_ascii(21, 400000, 60000, 100, 0.01)
    off = np.arange(0, reads.size + 1, 100, dtype=np.uint64)
    want = orc_c.build(reads.reshape(-1), off, 31)
    g = _dbg.Graph()
    g.set_option("estimate_scale_pct", 5)
    g.set_reads(reads.reshape(-1), off)
    g.build(31)
    assert g.stats()["count_launches"] == 2
    keys, stamps, counts, flags, succ, keys_raw = table(g, 31)
    assert np.array_equal(keys, want["keys"]) and np.array_equal(stamps, want["stamps"]) and np.array_equal(counts, want["counts"])
    check_succ(keys_raw, g.export_nodes()[2], succ, 31)
    g2 = _dbg.Graph()
    g2.set_reads(reads.reshape(-1), off)
    g2.build(31)
    assert g2.stats()["count_launches"] == 1


def test_two_word_counter_overflow_falls_back_to_32_bit_counters():
    """k > 31 keeps the four successor counters of a k-mer as 16-bit fields of its table slot; an edge seen more than
    65 535 times reports the overflow and the build runs again with separate 32-bit counters -- same graph."""
    rng = np.random.default_rng(5)
    one = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 40)
    reads = np.tile(one, (70000, 1))
    reads[:500] = synth.reads_ascii(9, 2000, 500, 40, 0.02)  # some ordinary reads next to the 69 500 copies
    off = np.arange(0, reads.size + 1, 40, dtype=np.uint64)
    for k in (33, 39):
        want = orc_c.build(reads.reshape(-1), off, k)
        assert want["counts"].max() > 65535
        g = _dbg.Graph()
        g.set_reads(reads.reshape(-1), off)
        g.build(k)
        keys, stamps, counts, flags, succ, keys_raw = table(g, k)
        o = np.argsort(g.export_nodes()[1], kind="stable")
        assert np.array_equal(keys, want["keys"]) and np.array_equal(g.export_keys_hi()[o], want["keys_hi"])
        assert np.array_equal(stamps, want["stamps"]) and np.array_equal(counts, want["counts"])


@pytest.mark.parametrize("k,err", [(21, 0.02), (31, 0.01), (13, 0.03)])
def test_edge_order_and_pull_reads_from_bucket_records_equal_the_pass_over_the_reads(k, err):
    """dbg_refine_edge_order has two implementations for the partitioned build (per range from the bucket's records;
    streaming the reads against a global set): same rank bytes for every node with two or more successors."""
    reads = synth.reads_ascii(31, 60000, 30000, 100, err)
    off = np.arange(0, reads.size + 1, 100, dtype=np.uint64)
    got = []
    for streaming in (0, 1):
        g = _dbg.Graph()
        g.set_option("refine_streaming", streaming)
        g.set_reads(reads.reshape(-1), off)
        g.build(k)
        g.refine_edge_order()
        _, stamps, counts, _ = g.export_nodes()
        o = np.argsort(stamps, kind="stable")
        mc, fs = g.export_orders()
        multi = (counts[o] != 0).sum(axis=1) >= 2
        g.prune(2)
        g.remove_tips()
        g.mark_pull_reads()  # the same two ways: per range from the records / all k-mers of the reads against a global set
        got.append((mc[o][multi], fs[o][multi], g.export_pull_reads(), g.sizes()["n_pull_reads"]))
    assert got[0][0].shape[0] > 100 and got[0][3] > 10
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][1], got[1][1])
    assert got[0][3] == got[1][3] and np.array_equal(got[0][2], got[1][2])


@pytest.mark.parametrize("k", [32, 40, 63])
@pytest.mark.parametrize("opts", [
    dict(),                                   # the super-k-mer / LDS engine of dbg_wsk.h, automatic geometry; k_wsk_count2 (one hint per slot)
    dict(bucket_bits=3),                      # 8 buckets: every table overflows and is counted in hash sub-ranges
    dict(wcount_kernel=1),                    # k_wsk_count: successor lookups after the insert (what 64-bit stamps run)
    dict(wcount_kernel=1, bucket_bits=3),
    dict(stamp64=1),                          # 64-bit stamps: what reads of 2 GiB and more get (k_wsk_count<uint64_t>)
    dict(stamp64=1, bucket_bits=3),
    dict(bucket_bits=9),
    dict(bucket_bits=14),
    dict(bucket_bits=21),                     # three multisplit levels
    dict(wide_engine=0),                      # the global reference-keyed table of dbg_wide.h
])
def test_two_word_engine_matches_c_oracle(k, opts):
    reads = synth.reads_ascii(8, 40000, 4000, 120, 0.01)
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, 120, dtype=np.uint64), k)
    g = build(reads, k, **opts)
    sz = g.sizes()
    assert sz["n_kmer_instances"] == want["n_kmer_instances"] and sz["n_edge_instances"] == want["n_edge_instances"]
    assert sz["n_nodes"] == want["n_nodes"]
    keys, stamps, counts, flags = g.export_nodes()
    hi = g.export_keys_hi()
    o = np.argsort(stamps, kind="stable")
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(hi[o], want["keys_hi"])
    assert np.array_equal(stamps[o], want["stamps"]) and np.array_equal(counts[o], want["counts"])
    assert np.array_equal(flags & 1, (stamps & np.uint64(1)).astype(np.uint8))
    succ = g.export_succ()
    rp, col, cnt = g.export_csr()
    assert int(rp[-1]) == sz["n_edges"] == int((counts != 0).sum())
    assert np.array_equal(col, succ[counts != 0]) and np.array_equal(cnt, counts[counts != 0])
    # successors: the (2k)-bit shifted k-mer
    full = (hi.astype(object) << 64) | keys.astype(object)
    mask = (1 << (2 * k)) - 1
    for code in range(4):
        has = counts[:, code] != 0
        assert np.all(succ[has, code] != 0xFFFFFFFF) and np.all(succ[~has, code] == 0xFFFFFFFF)
        got = full[succ[has, code]]
        assert np.array_equal(got, ((full[has] << 2) | code) & mask)


def test_two_word_engine_counter_overflow_falls_back():
    """The LDS engine for k > 31 keeps 16-bit successor counters; an edge seen more than 65 535 times must send the
    build to the global-table engine, with the same result."""
    read = synth.reads_ascii(3, 200, 1, 60, 0.0)[0]
    reads = np.tile(read, (70000, 1))
    k = 33
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, 60, dtype=np.uint64), k)
    assert int(want["counts"].max()) == 70000
    g = build(reads, k)
    keys, stamps, counts, flags = g.export_nodes()
    o = np.argsort(stamps, kind="stable")
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(g.export_keys_hi()[o], want["keys_hi"])
    assert np.array_equal(counts[o], want["counts"]) and np.array_equal(stamps[o], want["stamps"])
    assert g.stats()["n_records"] == 0  # the fallback engine has no records


@pytest.mark.parametrize("k", [40, 63])
def test_two_word_engine_ragged_reads(k):
    """Variable-length reads (including len < k, len == k, len == k + 1 and empty) packed back to back; k = 40 goes
    through the doubling-table extraction, k = 63 through the register kernel (k_wsk_extract_w<51>)."""
    rng = np.random.default_rng(6 + k)
    genome = synth.reads_ascii(10, 6000, 1, 6000, 0.0)[0]
    reads = []
    for i in range(4000):
        L = int(rng.integers(0, 130)) if i % 7 else int(rng.choice([0, k - 1, k, k + 1, 2 * k]))
        s = int(rng.integers(0, 6000 - 130))
        reads.append(genome[s:s + L])
    blob = np.concatenate(reads) if reads else np.zeros(0, np.uint8)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum([r.size for r in reads], out=off[1:])
    want = orc_c.build(blob, off, k)
    g = _dbg.Graph()
    g.set_reads(blob, off)
    g.build(k)
    assert g.sizes()["n_nodes"] == want["n_nodes"] and g.sizes()["n_kmer_instances"] == want["n_kmer_instances"]
    keys, stamps, counts, flags = g.export_nodes()
    o = np.argsort(stamps, kind="stable")
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(g.export_keys_hi()[o], want["keys_hi"])
    assert np.array_equal(counts[o], want["counts"]) and np.array_equal(stamps[o], want["stamps"])


@pytest.mark.parametrize("k", list(range(13, 64)))
def test_every_window_width_both_extraction_kernels(k):
    """One register extraction kernel per window w = k - 12 (k_sk_extract_w<1..19>, k_wsk_extract_w<20..51>) and the
    generic kernels that take the window minimum through LDS ("extract_generic"): both against the C oracle, on ragged
    reads (empty, shorter than k, exactly k, k + 1) so that read borders fall everywhere in a lane's 32 positions."""
    rng = np.random.default_rng(100 + k)
    genome = synth.reads_ascii(20 + k, 9000, 1, 9000, 0.0)[0]
    reads = []
    for i in range(1500):
        L = int(rng.integers(0, 160)) if i % 5 else int(rng.choice([0, k - 1, k, k + 1, 3 * k]))
        s = int(rng.integers(0, 9000 - 200))
        r = genome[s:s + L].copy()
        if L > 3 and i % 3 == 0:
            r[int(rng.integers(0, L))] = b"ACGT"[int(rng.integers(0, 4))]
        reads.append(r)
    blob = np.concatenate(reads)
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum([r.size for r in reads], out=off[1:])
    want = orc_c.build(blob, off, k)
    n_rec = []
    for generic in (0, 1):
        g = _dbg.Graph()
        g.set_option("extract_generic", generic)
        g.set_reads(blob, off)
        g.build(k)
        assert g.sizes()["n_nodes"] == want["n_nodes"] and g.sizes()["n_kmer_instances"] == want["n_kmer_instances"]
        assert g.sizes()["n_edge_instances"] == want["n_edge_instances"]
        keys, stamps, counts, flags = g.export_nodes()
        o = np.argsort(stamps, kind="stable")
        assert np.array_equal(keys[o], want["keys"]) and np.array_equal(stamps[o], want["stamps"])
        assert np.array_equal(counts[o], want["counts"])
        if k > 32:
            assert np.array_equal(g.export_keys_hi()[o], want["keys_hi"])
        n_rec.append(g.stats()["n_records"])
        g.close()
    assert n_rec[0] == n_rec[1] and n_rec[0] > 0  # the same super-k-mers either way


@pytest.mark.parametrize("k", [21, 31])
def test_count_kernel_16_bit_counters_fall_back_to_32_bit(k):
    """k_sk_count2 keeps 16-bit edge counters beside the successor hints: an edge seen more than 65 535 times must repeat
    the build with k_sk_count (32-bit counters) -- same result, two launches of the dominant kernel."""
    read = synth.reads_ascii(4, 200, 1, 60, 0.0)[0]
    rng = np.random.default_rng(1)
    other = synth.reads_ascii(5, 3000, 300, 60, 0.01)
    reads = np.concatenate([np.tile(read, (70000, 1)), other])
    reads = reads[rng.permutation(reads.shape[0])]
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, 60, dtype=np.uint64), k)
    assert int(want["counts"].max()) >= 70000
    g = build(reads, k)
    assert g.stats()["count_launches"] == 2
    keys, stamps, counts, flags, succ, keys_raw = table(g, k)
    assert np.array_equal(keys, want["keys"]) and np.array_equal(stamps, want["stamps"]) and np.array_equal(counts, want["counts"])
    _, _, counts_raw, _ = g.export_nodes()
    check_succ(keys_raw, counts_raw, succ, k)
    g1 = build(reads, k, count_kernel=1)   # the first kernel needs one launch
    assert g1.stats()["count_launches"] == 1 and np.array_equal(table(g1, k)[2], want["counts"])
    g3 = build(reads, k, stamp64=1, count_kernel_u64=3)   # k_sk_count3 (64-bit stamps) has 16-bit counters too
    assert g3.stats()["count_launches"] == 2
    t3 = table(g3, k)
    assert np.array_equal(t3[0], want["keys"]) and np.array_equal(t3[1], want["stamps"]) and np.array_equal(t3[2], want["counts"])


@pytest.mark.parametrize("k,opts", [(31, dict()), (21, dict(bucket_bits=9)), (31, dict(bucket_bits=3)), (13, dict(bucket_bits=14))])
def test_resolver_with_queries_grouped_by_target(k, opts):
    """Option "resolve_sorted" (off by default: measured slower): the cross-bucket queries are split by the 512
    level-1 groups of their target and the resolver takes the bucket hash from the split's key words -- same graph."""
    reads = synth.reads_ascii(23, 60000, 6000, 100, 0.01)
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, 100, dtype=np.uint64), k)
    g = build(reads, k, resolve_sorted=2, **opts)
    assert g.stats()["n_queries"] > 0
    keys, stamps, counts, flags, succ, keys_raw = table(g, k)
    assert np.array_equal(keys, want["keys"]) and np.array_equal(stamps, want["stamps"]) and np.array_equal(counts, want["counts"])
    check_succ(keys_raw, g.export_nodes()[2], g.export_succ(), k)


@pytest.mark.parametrize("opts", [dict(), dict(bucket_bits=9), dict(bucket_bits=3), dict(bucket_bits=14)])
def test_both_count_kernels_write_the_same_graph(opts):
    """k_sk_count (lookups after the insert) and k_sk_count2 (successor hints, pending list) on the same reads and geometry:
    the same nodes, counts, flags and successors (compared in dict order) and the same cross-bucket query count.  bucket_bits = 3: eight buckets, every table overflows into hash
    sub-ranges, so in-record successors land in other ranges (pending lookups that miss, queries beyond the LDS staging)."""
    reads = synth.reads_ascii(17, 80000, 8000, 100, 0.01)
    out = []
    for ck in (1, 2):
        g = build(reads, 31, count_kernel=ck, **opts)
        keys, stamps, counts, flags = g.export_nodes()
        succ = g.export_succ()
        rp, col, cnt = g.export_csr()
        assert np.array_equal(col, succ[counts != 0]) and np.array_equal(cnt, counts[counts != 0])
        o = np.argsort(stamps, kind="stable")     # node ids follow the order in which buckets reserve them: compare in dict order
        succ_keys = np.where(succ == _dbg.NO_NODE, np.uint64(0xFFFFFFFFFFFFFFFF), keys[np.minimum(succ, keys.size - 1)])
        out.append((keys[o], stamps[o], counts[o], flags[o], succ_keys[o], g.stats()["n_queries"], g.stats()["n_buckets"]))
    a, b = out
    for x, y in zip(a[:5], b[:5]):
        assert np.array_equal(x, y)
    assert a[5] == b[5] and a[6] == b[6]
