"""Sharded (multi-rank) build: union of the shards == the single-rank oracle result.

CPU (gloo, world_size 2 and 4): multi_gpu.py's exchange logic around a numpy model of the device steps,
then the gather for traversal.
GPU (-m gpu): the real library, two ranks sharing the test box's single GPU over gloo: sharded build, gather to
rank 0, and the whole rest of the path (refine, prune, tips, pull-out reads, walk) equal to a single-GPU build.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import inproc_dist
import synth
from oracle import orc_c

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(mode, world, tmp_path, k, n_reads, read_len, max_msg=None, chunks=None):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        if max_msg:
            env["SHARD_MAX_MSG"] = str(max_msg)
        if chunks:
            env["SHARD_CHUNKS"] = str(chunks)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "shard_worker.py"), mode, str(tmp_path), str(k),
                                       str(n_reads), str(read_len)], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]


def check(shards, world, k, n_reads, read_len):
    per = n_reads // world
    reads = np.concatenate([synth.reads_ascii(77, max(4 * read_len, n_reads * read_len // 20), per, read_len, 0.01,
                                              first_read=r * per) for r in range(world)])
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64), k)
    keys = np.concatenate([s["keys"] for s in shards])
    keys_hi = np.concatenate([s["keys_hi"] for s in shards])
    stamps = np.concatenate([s["stamps"] for s in shards])
    counts = np.concatenate([s["counts"] for s in shards])
    assert keys.size == want["n_nodes"], "every k-mer is owned by exactly one shard"
    o = np.argsort(stamps, kind="stable")
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(keys_hi[o], want["keys_hi"])
    assert np.array_equal(stamps[o], want["stamps"])       # global first-occurrence stamps
    assert np.array_equal(counts[o], want["counts"])
    # successors: (owner << 29) | id on the owner; the successor's k-mer is this k-mer shifted by the base (two words for k > 31)
    u64 = np.uint64
    lo_mask = u64((1 << (2 * k)) - 1) if 2 * k < 64 else u64(0xFFFFFFFFFFFFFFFF)
    hi_mask = u64((1 << (2 * k - 64)) - 1) if 2 * k > 64 else u64(0)
    for r, s in enumerate(shards):
        for code in range(4):
            has = s["counts"][:, code] != 0
            ref = s["succ"][has, code]
            assert np.all(ref != 0xFFFFFFFF)
            owner, idx = ref >> 29, ref & ((1 << 29) - 1)
            got = np.empty(ref.size, dtype=np.uint64)
            got_hi = np.empty(ref.size, dtype=np.uint64)
            for d in range(world):
                sel = owner == d
                got[sel] = shards[d]["keys"][idx[sel]]
                got_hi[sel] = shards[d]["keys_hi"][idx[sel]]
            lo, hi = s["keys"][has], s["keys_hi"][has]
            assert np.array_equal(got, ((lo << u64(2)) | u64(code)) & lo_mask)
            assert np.array_equal(got_hi, ((hi << u64(2)) | (lo >> u64(62))) & hi_mask)
            assert np.all(s["succ"][~has, code] == 0xFFFFFFFF)


@pytest.mark.parametrize("world,max_msg", [(2, None), (4, None), (4, 200)])  # 200 B per message: transfers go in several rounds
def test_exchange_logic_on_cpu_gloo(world, max_msg, tmp_path):
    shards = run_ranks("fake", world, tmp_path, 9, 64, 40, max_msg)
    check(shards, world, 9, 64, 40)
    # gather for traversal: rank 0 holds the whole graph over the rank-major concatenation of the reads
    m = np.load(os.path.join(tmp_path, "merged.npz"))
    k, per = 9, 64 // world
    reads = np.concatenate([synth.reads_ascii(77, max(4 * 40, 64 * 40 // 20), per, 40, 0.01, first_read=r * per)
                            for r in range(world)])
    assert np.array_equal(m["bases"], reads.reshape(-1)) and np.array_equal(m["offsets"], np.arange(0, reads.size + 1, 40))
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, 40, dtype=np.uint64), k)
    o = np.argsort(m["stamps"], kind="stable")
    assert np.array_equal(m["keys"][o], want["keys"]) and np.array_equal(m["counts"][o], want["counts"])
    mask = np.uint64((1 << (2 * k)) - 1)
    for code in range(4):  # successor ids are positions in the merged arrays
        has = m["counts"][:, code] != 0
        assert np.array_equal(m["keys"][m["succ"][has, code]], ((m["keys"][has] << np.uint64(2)) | np.uint64(code)) & mask)
        assert np.all(m["succ"][~has, code] == 0xFFFFFFFF)


@pytest.mark.parametrize("chunks", [1, 2])
def test_ranks_times_passes_on_cpu_gloo(tmp_path, chunks):
    """multi_gpu.sharded_build_multipass with real torch.distributed (gloo, 2 processes x 2 passes; rank 0 hands out
    64-bit stamps): same checks as the in-process run below.  chunks = 2: the records cut and sent in two parts."""
    world, n_passes, k, n_reads, read_len = 2, 2, 9, 64, 40
    files = run_ranks("fake_mp", world, tmp_path, k, n_reads, read_len, chunks=chunks)
    parts = [{key: f[f"{key}{p}"] for key in ("keys", "stamps", "counts", "succ_part", "succ_id")} for f in files for p in range(n_passes)]
    reads = np.concatenate([rank_reads(world, r, n_reads, read_len) for r in range(world)])
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64), k)
    keys = np.concatenate([d["keys"] for d in parts])
    stamps = np.concatenate([d["stamps"] for d in parts])
    o = np.argsort(stamps, kind="stable")
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(stamps[o], want["stamps"])
    assert np.array_equal(np.concatenate([d["counts"] for d in parts])[o], want["counts"])
    mask = np.uint64((1 << (2 * k)) - 1)
    for d in parts:
        for code in range(4):
            has = d["counts"][:, code] != 0
            got = np.array([parts[q]["keys"][i] for q, i in zip(d["succ_part"][has, code], d["succ_id"][has, code])], dtype=np.uint64)
            assert np.array_equal(got, ((d["keys"][has] << np.uint64(2)) | np.uint64(code)) & mask)


def rank_reads(world, rank, n_reads, read_len):
    per = n_reads // world
    return synth.reads_ascii(77, max(4 * read_len, n_reads * read_len // 20), per, read_len, 0.01, first_read=rank * per)


@pytest.mark.parametrize("wide_stamp_ranks", [(), (2, 5), tuple(range(8))])
def test_eight_ranks_in_process_exchange_logic_on_cpu(wide_stamp_ranks):
    """Three owner bits (8 logical shards, host-side exchange): the real multi_gpu.sharded_build
    on eight threads of this process around the numpy model of the device steps, checksummed exchanges included.
    wide_stamp_ranks: ranks whose rank-local stamps are 64-bit (they hold 2 GiB of reads or more) -- the others widen
    theirs before the exchange, a receiver sees one width."""
    import multi_gpu
    import shard_worker
    k, n_reads, read_len = 9, 128, 40

    def one(dist, rank):
        g = shard_worker.NumpyShardGraph([row.tobytes().decode() for row in rank_reads(8, rank, n_reads, read_len)], k,
                                         stamp64=rank in wide_stamp_ranks)
        multi_gpu.sharded_build(g, k, dist)
        keys, stamps, counts, succ = g.export()
        return {"keys": keys, "keys_hi": np.zeros_like(keys), "stamps": stamps, "counts": counts, "succ": succ}

    check(inproc_dist.run_ranks(8, one), 8, k, n_reads, read_len)


@pytest.mark.parametrize("ranks,n_passes,wide_stamp_ranks,chunks", [(4, 2, (), 1), (2, 4, (1,), 1), (8, 1, (), 1), (1, 4, (), 1),
                                                                   # the records cut and sent in parts (dbg_shard_extract_part)
                                                                   (8, 1, (), 2), (4, 2, (2,), 3), (1, 1, (), 4)])
def test_ranks_times_passes_exchange_logic_on_cpu(ranks, n_passes, wide_stamp_ranks, chunks):
    """multi_gpu.sharded_build_multipass (BASELINE.json configs[3]: ranks x passes) around the numpy model of
    dbg_shard_build_multipass / dbg_part_*: the parts of all ranks together hold every k-mer of the reads once, with
    the counts of a single table, and every successor -- same part, another part of the rank, another rank -- names
    (virtual shard, local id) of the shifted k-mer."""
    import multi_gpu
    import shard_worker
    k, n_reads, read_len = 9, 96, 40

    def one(dist, rank):
        g = shard_worker.NumpyMultipassGraph([row.tobytes().decode() for row in rank_reads(ranks, rank, n_reads, read_len)], k,
                                             stamp64=rank in wide_stamp_ranks)
        multi_gpu.sharded_build_multipass(g, k, dist, n_passes, chunks=chunks)
        return g.parts

    parts = [d for rank_parts in inproc_dist.run_ranks(ranks, one) for d in rank_parts]  # index = virtual shard
    assert len(parts) == ranks * n_passes
    reads = np.concatenate([rank_reads(ranks, r, n_reads, read_len) for r in range(ranks)])
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64), k)
    keys = np.array([key for d in parts for key in d["keys"]], dtype=np.uint64)
    stamps = np.concatenate([d["stamps"] for d in parts])
    counts = np.concatenate([d["counts"].reshape(-1, 4) for d in parts])
    o = np.argsort(stamps, kind="stable")
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(stamps[o], want["stamps"])
    assert np.array_equal(counts[o], want["counts"])
    mask, crossing = (1 << (2 * k)) - 1, 0
    for v, d in enumerate(parts):
        for j, key in enumerate(d["keys"]):
            for code in range(4):
                if d["counts"][j, code]:
                    q, i = int(d["succ_part"][j, code]), int(d["succ_id"][j, code])
                    assert parts[q]["keys"][i] == ((key << 2) | code) & mask
                    crossing += q // n_passes != v // n_passes
                else:
                    assert d["succ_id"][j, code] == 0xFFFFFFFF
    assert ranks == 1 or crossing > 0


def test_records_in_parts_when_the_receive_buffers_must_grow(monkeypatch):
    """_exchange_records_in_parts sizes its receive buffers from the first part; a later part that does not fit makes it wait for
    the posted exchanges, allocate larger buffers and copy what has arrived.  PART_SLACK = 0.3 forces that at every part."""
    import multi_gpu
    import shard_worker
    monkeypatch.setattr(multi_gpu, "PART_SLACK", 0.3)
    ranks, n_passes, chunks, k, n_reads, read_len = 4, 1, 4, 9, 96, 40

    def one(dist, rank):
        g = shard_worker.NumpyMultipassGraph([row.tobytes().decode() for row in rank_reads(ranks, rank, n_reads, read_len)], k)
        multi_gpu.sharded_build_multipass(g, k, dist, n_passes, chunks=chunks)
        return g.parts

    parts = [d for rank_parts in inproc_dist.run_ranks(ranks, one) for d in rank_parts]
    reads = np.concatenate([rank_reads(ranks, r, n_reads, read_len) for r in range(ranks)])
    want = orc_c.build(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64), k)
    keys = np.array([key for d in parts for key in d["keys"]], dtype=np.uint64)
    stamps = np.concatenate([d["stamps"] for d in parts])
    counts = np.concatenate([d["counts"].reshape(-1, 4) for d in parts])
    o = np.argsort(stamps, kind="stable")
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(stamps[o], want["stamps"]) and np.array_equal(counts[o], want["counts"])


@pytest.mark.parametrize("posted", [False, True])
def test_damaged_exchange_is_detected(posted):
    """multi_gpu.ExchangeCheck: a message that arrives with one wrong word must raise, not build a wrong graph.
    posted: the exchange that does not wait (ExchangeCheck.post / wait: records sent in parts)."""
    import multi_gpu
    import torch

    class Corrupting(inproc_dist.InProcDist):
        def all_to_all_single(self, out, inp, out_splits=None, in_splits=None, async_op=False):
            work = super().all_to_all_single(out, inp, out_splits, in_splits, async_op)
            if out_splits is not None and self.get_rank() == 1 and out.numel() > 3:
                out[3] += 1  # one flipped value in what rank 1 received
            return work

    def one(dist, rank):
        dist.__class__ = Corrupting
        xc = multi_gpu.ExchangeCheck(dist)
        t = torch.arange(10, dtype=torch.int64) + 100 * rank
        if posted:
            out = torch.empty(10, dtype=torch.int64)
            xc.post(t, [5, 5], out, [5, 5], "test", 5)
            xc.wait()
            if rank == 0:
                assert out.tolist() == [0, 1, 2, 3, 4, 100, 101, 102, 103, 104]
        else:
            xc.alltoallv(t, [5, 5], [5, 5], "test")
        try:
            xc.verify()
        except RuntimeError as e:
            return str(e)
        return None

    got = inproc_dist.run_ranks(2, one)
    # the check is collective: BOTH ranks raise (a rank that carried on alone would hang in the next collective), and
    # both name the receiver and the sender of the damaged message
    assert all(x is not None and "rank 1 received damaged 'test' messages from ranks [0]" in x for x in got), got


@pytest.mark.gpu
@pytest.mark.parametrize("k,n_reads,read_len,bucket_bits,wide_stamp_ranks",
                         [(31, 24000, 150, 0, ()), (21, 8000, 100, 0, ()), (63, 8000, 150, 0, ()),
                          (31, 8000, 150, 21, ()),  # 21 bits: 9 (senders) + 10 + 2, third level
                          (31, 24000, 150, 0, (1, 6)), (21, 8000, 100, 0, tuple(range(8)))])
def test_eight_shards_in_process_on_one_gpu(k, n_reads, read_len, bucket_bits, wide_stamp_ranks):
    """shard_bits = 3 through the C ABI before a real 8-GPU node sees it: eight handles on cuda:0, one thread per rank,
    the real multi_gpu.sharded_build with an in-process exchange; union of the shards == the C oracle.
    wide_stamp_ranks: ranks that hand out 64-bit rank-local stamps ("shard_stamp64": what a rank holding 2 GiB of reads or
    more does by itself)."""
    import _dbg
    import multi_gpu

    def one(dist, rank):
        reads = rank_reads(8, rank, n_reads, read_len)
        g = _dbg.Graph(device=0)
        if bucket_bits:
            g.set_option("bucket_bits", bucket_bits)
        if rank in wide_stamp_ranks:
            g.set_option("shard_stamp64", 1)
        g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64))
        multi_gpu.sharded_build(g, k, dist)
        keys, stamps, counts, _ = g.export_nodes()
        out = {"keys": keys, "keys_hi": g.export_keys_hi(), "stamps": stamps, "counts": counts, "succ": g.export_succ()}
        rp, col, cnt = g.export_csr()
        assert np.array_equal(col, out["succ"][counts != 0]) and np.array_equal(cnt, counts[counts != 0])
        g.close()
        return out

    check(inproc_dist.run_ranks(8, one), 8, k, n_reads, read_len)


@pytest.mark.gpu
@pytest.mark.parametrize("k,n_reads,read_len", [(21, 4000, 100), (31, 20000, 150), (5, 400, 30), (40, 6000, 120), (63, 8000, 150)])
def test_two_ranks_on_one_gpu(k, n_reads, read_len, tmp_path):
    shards = run_ranks("gpu", 2, tmp_path, k, n_reads, read_len)
    check(shards, 2, k, n_reads, read_len)


@pytest.mark.gpu
@pytest.mark.parametrize("k,world,n_passes,chunks", [(31, 2, 2, 1), (63, 4, 1, 1), (31, 2, 1, 2), (40, 2, 1, 3)])
def test_traversal_in_parts_over_real_processes(k, world, n_passes, chunks, tmp_path):
    """part_traversal with one PROCESS per rank (gloo; the ranks share the box's GPU): every rank ends with the same
    branch_kmer / already_pull_out / contigs as the reference restatement on all reads, and the pull-out flags of the ranks,
    in rank order, are the reference's pull_out_read."""
    import contextlib
    import io
    from oracle import dbg_oracle as orc
    n_reads, read_len = 4000, 120
    os.environ["SHARD_PASSES"] = str(n_passes)
    try:
        files = run_ranks("gpu_parts", world, tmp_path, k, n_reads, read_len, chunks=chunks)  # chunks = 2: records sent in two parts
    finally:
        del os.environ["SHARD_PASSES"]
    per = n_reads // world
    reads = np.concatenate([synth.reads_ascii(77, max(4 * read_len, n_reads * read_len // 20), per, read_len, 0.01,
                                              first_read=r * per) for r in range(world)])
    strs = [row.tobytes().decode() for row in reads]
    with contextlib.redirect_stdout(io.StringIO()):
        og, opull, obranch, opulled, oect = orc.construct_graph(strs, k, threshold=2)
        octg = orc.output_contigs(og, obranch, opulled)
    assert len(obranch) > 0 and len(octg) > 0
    flags = np.concatenate([f["read_flags"] for f in files]).astype(bool)
    assert [s for s, fl in zip(strs, flags) if fl] == list(opull)
    for f in files:
        assert f["branch"].tolist() == list(obranch) and f["pulled"].tolist() == list(opulled)
        assert f["contigs"].tolist() == list(octg)
        assert f["scores"].tolist() == [orc.get_score(oect, c, k) for c in octg]


@pytest.mark.gpu
def test_four_ranks_on_one_gpu(tmp_path):
    """Two owner bits (the box allows at most six processes on its GPU, so eight ranks run only on the real node)."""
    shards = run_ranks("gpu", 4, tmp_path, 21, 8000, 100)
    check(shards, 4, 21, 8000, 100)


@pytest.mark.gpu
def test_four_ranks_on_one_gpu_two_word_kmers(tmp_path):
    """BASELINE.json configs[4] in miniature: k = 63 over four ranks (the k-mer instances travel)."""
    shards = run_ranks("gpu", 4, tmp_path, 63, 8000, 150)
    check(shards, 4, 63, 8000, 150)


@pytest.mark.gpu
def test_one_rank_over_rccl_equals_the_single_build_with_multi_round_exchange():
    """The exchange itself over nccl (= RCCL), at a size where a (source, destination) message exceeds
    MAX_MESSAGE_BYTES and goes in rounds: 4 M reads at k = 63 are 3.5e8 k-mer instance tuples, 2.8 GB per array.
    (A 2 GiB self-copy over nccl arrived damaged; the node and edge totals then differ from the single build.)"""
    env = dict(os.environ, READS="4", K="63", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "shard_check.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("4000000 ")]
    assert lines and lines[-1].endswith(" OK"), out.stdout[-2000:]


@pytest.mark.gpu
def test_a_shard_that_outgrows_its_id_space_fails_loudly():
    """Sharded successor ids are (owner << 29) | local id: a shard that would hold more nodes than its id space must
    end in DBG_E_CAPACITY with a message, never in ids that alias another shard (the ceiling is lowered with the
    "shard_node_limit" option so that a few thousand reads reach it)."""
    import _dbg
    import multi_gpu

    def one(dist, rank):
        reads = rank_reads(2, rank, 8000, 100)
        g = _dbg.Graph(device=0)
        g.set_option("shard_node_limit", 5000)
        g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, 100, dtype=np.uint64))
        try:
            multi_gpu.sharded_build(g, 21, dist)
        except _dbg.DbgError as e:
            return (e.code, str(e))
        finally:
            g.close()
        return None

    # both ranks fail at the same step (each owns half of ~60 000 nodes), so neither waits for the other
    got = inproc_dist.run_ranks(2, one)
    assert all(x is not None and x[0] == _dbg.DBG_E_CAPACITY and "capacity" in x[1] for x in got), got


@pytest.mark.gpu
@pytest.mark.parametrize("k,n_passes", [(31, 0), (63, 0), (21, 4)])
def test_a_rank_without_reads(k, n_passes):
    """An uneven split can leave a rank with no reads at all: it still takes part in every exchange and owns its share
    of the k-mers of the others (sharded build, two-word records, ranks x passes)."""
    import _dbg
    import multi_gpu
    L = 100

    def reads_of(r):
        return synth.reads_ascii(5, 4000, 300, L, 0.01) if r != 1 else np.zeros((0, L), dtype=np.uint8)

    def one(dist, rank):
        g = _dbg.Graph(device=0)
        rd = reads_of(rank)
        g.set_reads(rd.reshape(-1), np.arange(0, rd.size + 1, L, dtype=np.uint64))
        if n_passes:
            multi_gpu.sharded_build_multipass(g, k, dist, n_passes)
            n = sum(g.part_sizes(p)["n_nodes"] for p in range(n_passes))
        else:
            multi_gpu.sharded_build(g, k, dist)
            n = g.sizes()["n_nodes"]
        g.close()
        return n

    got = inproc_dist.run_ranks(4, one)
    allr = np.concatenate([reads_of(r) for r in range(4)])
    want = orc_c.build(allr.reshape(-1), np.arange(0, allr.size + 1, L, dtype=np.uint64), k, export=False)
    assert sum(got) == want["n_nodes"] and min(got) > 0


@pytest.mark.parametrize("world", [2, 4])
def test_part_traversal_routing_on_cpu(world):
    """part_traversal._Net -- how id lists and rows reach the rank that owns them -- on CPU tensors, `world` in-process
    ranks: route() delivers every row to its destination and nothing else, gather_all() gives every rank the same
    rank-ordered table (columns of several elements per row included), min_u64() is the element-wise minimum with the
    all-ones sentinel for 'never seen'."""
    import torch
    import part_traversal as pt

    def one(dist, rank):
        net = pt._Net(dist, "cpu")
        rng = np.random.default_rng(100 + rank)
        n = int(rng.integers(0, 50))
        dest = torch.from_numpy(rng.integers(0, world, size=n))
        val = torch.arange(n, dtype=torch.int64) + 1000 * rank
        got_dest, got_val = net.route(dest, dest.clone(), val)
        assert bool((got_dest == rank).all())
        rows = torch.arange(3, dtype=torch.int64) + 10 * rank
        wide = (torch.arange(12, dtype=torch.int64) + 100 * rank)
        a, b = net.gather_all(rows, wide, widths=[1, 4])
        fs = np.full((5, 4), np.iinfo(np.uint64).max, dtype=np.uint64)
        fs[rank % 5, rank % 4] = 7 + rank
        fs[4, 3] = 100 - rank
        return sorted(got_val.tolist()), (dest.tolist(), val.tolist()), a.tolist(), b.tolist(), net.min_u64(fs)

    got = inproc_dist.run_ranks(world, one)
    for r in range(world):
        want = sorted(v for _, (dest, val), _, _, _ in got for d, v in zip(dest, val) if d == r)
        assert got[r][0] == want
        assert got[r][2] == [x + 10 * q for q in range(world) for x in range(3)]
        assert got[r][3] == [x + 100 * q for q in range(world) for x in range(12)]
        m = got[r][4]
        assert m[4, 3] == 100 - (world - 1)
        for q in range(world):
            assert m[q % 5, q % 4] == min(7 + p for p in range(world) if (p % 5, p % 4) == (q % 5, q % 4))
        assert int((m == np.iinfo(np.uint64).max).sum()) == 20 - len({(p % 5, p % 4) for p in range(world)} | {(4, 3)})
