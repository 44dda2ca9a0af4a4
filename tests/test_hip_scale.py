"""Parity at sizes the Python oracle cannot reach: against the C oracle at 1M reads, and through
size-independent properties at the full BASELINE.json configs[1] size (10M x 150 bp, k = 31)."""
import numpy as np
import pytest

import _dbg
import synth
from oracle import orc_c

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("err,final,scale", [(0.0, False, 1.0), (0.01, False, 1.0), (0.0, True, 0.3)])  # final mode with errors: all simple paths, exponential
def test_config1_whole_path_vs_oracle(err, final, scale):
    """BASELINE.json configs[0]: 10k x 100 bp, k = 21 (10x coverage of a 100 kbp genome): the whole drop-in
    surface against the Python oracle, orders included.  (scale 0.3: the same at 3k reads over 30 kbp -- the Python oracle
    spells the one long contig of error-free reads in a minute at full size, and the non-final case already pays that.)"""
    import contextlib
    import io
    import debruijn as prod
    from golden_util import canonical
    from oracle import dbg_oracle as orc
    reads = synth.reads_list(1, int(100_000 * scale), int(10_000 * scale), 100, err)
    res = []
    for mod in (prod, orc):
        with contextlib.redirect_stdout(io.StringIO()) as buf:
            g, pull, branch, pulled, ect = mod.construct_graph(list(reads), 21, threshold=2, final=final)
            contigs = mod.output_contigs(g, branch, pulled)
        r = canonical(g, pull, branch, pulled, ect, contigs)
        r["stdout"] = buf.getvalue()
        res.append(r)
    for field in res[1]:
        assert res[0][field] == res[1][field], field
    assert len(res[0]["vertices"]) > 90_000 * scale and res[0]["contigs"]


def test_drop_in_surface_at_two_million_reads():
    """construct_graph on 2M x 150 bp (6.7e7 nodes): the Python surface stays usable through the lazy views."""
    import contextlib
    import io
    import debruijn as prod
    n, L, k, G = 2_000_000, 150, 31, 10_000_000
    g0 = _dbg.Graph()
    g0.synth_reads(1, G, n, L, 0.01)
    bases, off = g0.copy_reads()
    g0.close()
    text = bases.tobytes().decode("ascii")

    class Reads:  # a list-like over the packed buffer (2M Python strings would work too, this is cheaper)
        def __len__(self):
            return n

        def __getitem__(self, i):
            return text[int(off[i]):int(off[i + 1])]

        def __iter__(self):
            return (self[i] for i in range(n))

    dev = _dbg.Graph()
    dev.set_reads(bases, off)
    reads = prod.DeviceReads.__new__(prod.DeviceReads)  # adopt the resident reads
    reads._graph, reads._n, reads._host = dev, n, (text, off)
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        (V, E), pull, branch, pulled, ect = prod.construct_graph(reads, k, threshold=2)
    assert not isinstance(V, dict) and len(V) > 5 * 10**7
    assert buf.getvalue().startswith("number of 31mer:  %d\n" % len(V))
    it = iter(V)
    first = [next(it) for _ in range(1000)]
    assert first[0] == text[:k] and V[first[0]].indegree == 0          # dict order starts at read 0, position 0
    for lab in first[:200]:
        nd = V[lab]
        assert nd.label == lab and nd.outdegree == sum((lab + c) in ect for c in "ACGT")
        if lab in E:
            assert all(s in V and s[:-1] == lab[1:] for s in E[lab])
    assert len(E) == len(V) - len(pulled)
    assert len(branch) and all(len(E[b]) > 1 for b in branch[:200])   # branch nodes are never pulled out
    bs = set(branch)
    assert len(pull) and all(any(r[i:i + k] in bs for i in range(len(r) - k + 1)) for r in pull[:20])
    some = text[150 * 777:150 * 777 + k + 1]                           # an edge that certainly exists
    assert ect[some] >= 1 and some in ect


def test_one_million_reads_vs_c_oracle():
    n, L, k = 1_000_000, 150, 31
    reads = synth.reads_ascii(1, 5_000_000, n, L, 0.01)
    off = np.arange(0, reads.size + 1, L, dtype=np.uint64)
    want = orc_c.build(reads.reshape(-1), off, k)
    g = _dbg.Graph()
    g.synth_reads(1, 5_000_000, n, L, 0.01)              # device twin of the generator
    assert g.reads_checksum() == synth.checksum(reads)
    g.build(k)
    keys, stamps, counts, flags = g.export_nodes()
    o = np.argsort(stamps, kind="stable")
    assert g.sizes()["n_nodes"] == want["n_nodes"]
    assert np.array_equal(keys[o], want["keys"])
    assert np.array_equal(stamps[o], want["stamps"])
    assert np.array_equal(counts[o], want["counts"])
    succ = g.export_succ()
    mask = np.uint64((1 << (2 * k)) - 1)
    for code in range(4):
        has = counts[:, code] != 0
        assert np.array_equal(keys[succ[has, code]], ((keys[has] << np.uint64(2)) | np.uint64(code)) & mask)


@pytest.mark.parametrize("k", [32, 47, 63])
def test_two_word_kmers_vs_c_oracle(k):
    """BASELINE.json configs[4] key width (k = 63, 128-bit keys) at a size the C oracle builds in seconds."""
    n, L = 300_000, 150
    reads = synth.reads_ascii(3, 1_500_000, n, L, 0.01)
    off = np.arange(0, reads.size + 1, L, dtype=np.uint64)
    want = orc_c.build(reads.reshape(-1), off, k)
    g = _dbg.Graph()
    g.synth_reads(3, 1_500_000, n, L, 0.01)
    g.build(k)
    sz = g.sizes()
    assert sz["n_nodes"] == want["n_nodes"] and sz["n_kmer_instances"] == want["n_kmer_instances"]
    assert sz["n_edge_instances"] == want["n_edge_instances"]
    keys, stamps, counts, flags = g.export_nodes()
    hi = g.export_keys_hi()
    o = np.argsort(stamps, kind="stable")
    assert np.array_equal(keys[o], want["keys"]) and np.array_equal(hi[o], want["keys_hi"])
    assert np.array_equal(stamps[o], want["stamps"])
    assert np.array_equal(counts[o], want["counts"])
    assert np.array_equal(flags & 1, (stamps & np.uint64(1)).astype(np.uint8))
    # successors: (key << 2 | code) mod 4^k, two words
    succ = g.export_succ()
    hb = 2 * k - 64
    hmask = np.uint64((1 << hb) - 1) if hb else np.uint64(0)
    for code in range(4):
        has = counts[:, code] != 0
        s = succ[has, code]
        assert np.array_equal(keys[s], (keys[has] << np.uint64(2)) | np.uint64(code))
        assert np.array_equal(hi[s], ((hi[has] << np.uint64(2)) | (keys[has] >> np.uint64(62))) & hmask)
    rp, col, cnt = g.export_csr()
    assert rp[-1] == sz["n_edges"] == int((counts != 0).sum())


def test_two_word_kmers_full_size_invariants():
    """10M x 150 bp at k = 63: the sizes and sums that hold whatever the input."""
    n, L, k, G = 10_000_000, 150, 63, 50_000_000
    g = _dbg.Graph()
    g.synth_reads(1, G, n, L, 0.0)
    g.build(k)
    sz = g.sizes()
    assert sz["n_kmer_instances"] == n * (L - k + 1) and sz["n_edge_instances"] == n * (L - k)
    assert sz["n_nodes"] <= G - k + 1  # error-free reads hold only k-mers of the genome
    keys, stamps, counts, flags = g.export_nodes()
    assert int(counts.sum(dtype=np.uint64)) == sz["n_edge_instances"]
    assert np.unique(stamps).size == stamps.size
    pos = stamps >> np.uint64(1)
    assert np.array_equal((stamps & np.uint64(1)) == 0, pos % np.uint64(L) == 0)
    # the first occurrence really holds the k-mer (sample)
    hi = g.export_keys_hi()
    reads, _ = g.copy_reads()
    idx = np.linspace(0, keys.size - 1, 5000).astype(np.int64)
    for i in idx[::50]:
        v = 0
        for ch in reads[int(pos[i]):int(pos[i]) + k]:
            v = (v << 2) | ((int(ch) >> 1) & 3)
        assert v == (int(hi[i]) << 64) | int(keys[i])


@pytest.mark.parametrize("err", [0.0, 0.01])
def test_full_size_invariants(err):
    """10M x 150 bp, k = 31: properties that hold whatever the size."""
    n, L, k, G = 10_000_000, 150, 31, 50_000_000
    g = _dbg.Graph()
    g.synth_reads(1, G, n, L, err)
    g.build(k)
    sz = g.sizes()
    assert sz["n_kmer_instances"] == n * (L - k + 1)
    assert sz["n_edge_instances"] == n * (L - k)
    keys, stamps, counts, flags = g.export_nodes()
    assert int(counts.sum(dtype=np.uint64)) == sz["n_edge_instances"]        # every (k+1)-mer instance counted once
    assert sz["n_edges"] == int((counts != 0).sum())
    import torch

    def dev_sorted(a):  # (3.6e8 values: a numpy sort takes half a minute, the card 50 ms; keys and stamps stay below 2^63)
        v, idx = torch.sort(torch.from_numpy(a.view(np.int64)).cuda())
        return v, idx

    sk, ok_ = dev_sorted(keys)
    assert bool((sk[1:] != sk[:-1]).all())                                    # nodes are distinct k-mers
    ss, _ = dev_sorted(stamps)
    assert bool((ss[1:] != ss[:-1]).all())                                    # first occurrences are distinct positions
    del ss
    pos = stamps >> np.uint64(1)
    assert np.array_equal((stamps & np.uint64(1)) == 0, pos % np.uint64(L) == 0)  # indegree 0 <=> read position 0
    assert sz["n_starts"] == int(((stamps & np.uint64(1)) == 0).sum())
    # the first occurrence really holds the k-mer: re-encode it from the reads on the host
    reads, _ = g.copy_reads()
    idx = np.linspace(0, keys.size - 1, 20000).astype(np.int64)
    code = ((reads[(pos[idx][:, None] + np.arange(k, dtype=np.uint64)[None, :]).astype(np.int64)] >> 1) & 3).astype(np.uint64)
    shifts = (2 * (k - 1 - np.arange(k, dtype=np.uint64))).astype(np.uint64)
    assert np.array_equal((code << shifts[None, :]).sum(axis=1, dtype=np.uint64), keys[idx])
    if err == 0.0:
        # an error-free read set holds exactly the genome's k-mers that the reads cover
        assert sz["n_nodes"] <= G - k + 1
    # successor ids are consistent with the shifted key (all edges)
    succ = g.export_succ()
    mask = np.uint64((1 << (2 * k)) - 1)
    for c in range(4):
        has = counts[:, c] != 0
        assert np.array_equal(keys[succ[has, c]], ((keys[has] << np.uint64(2)) | np.uint64(c)) & mask)
    # building again on the same handle gives the same table (order may differ)
    g.build(k)
    keys2, stamps2, counts2, _ = g.export_nodes()
    sk2, ok2 = dev_sorted(keys2)
    assert torch.equal(sk, sk2)
    del sk, sk2
    assert torch.equal(torch.from_numpy(stamps.view(np.int64)).cuda()[ok_], torch.from_numpy(stamps2.view(np.int64)).cuda()[ok2])
    assert torch.equal(torch.from_numpy(counts.view(np.int32)).cuda()[ok_], torch.from_numpy(counts2.view(np.int32)).cuda()[ok2])


def test_twenty_five_million_reads_fit_one_gpu():
    """2.5x BASELINE configs[1]: node and edge arrays are sized from the distinct-k-mer estimate (the worst case
    would need 3e9 x 58 B), 2^19..2^20 buckets.  Sizes and CSR sums only: the node exports would be 50 GB."""
    n, L, k = 25_000_000, 150, 31
    g = _dbg.Graph()
    g.synth_reads(1, n * 5, n, L, 0.01)
    g.build(k)
    sz = g.sizes()
    assert sz["n_kmer_instances"] == n * (L - k + 1) and sz["n_edge_instances"] == n * (L - k)
    assert 8.5e8 < sz["n_nodes"] < 9.5e8 and sz["n_nodes"] <= sz["n_edges"] + n  # only read-final k-mers lack a successor
    rp, col, cnt = g.export_csr()
    assert rp[-1] == sz["n_edges"] == col.size
    assert int(cnt.sum(dtype=np.uint64)) == sz["n_edge_instances"]  # every (k+1)-mer instance counted exactly once
    assert int(col.max()) < sz["n_nodes"] and np.all(np.diff(rp.astype(np.int64)) <= 4)
    g.prune(2)
    g.remove_tips()
    g.mark_pull_reads()
    g.walk(False, 1 << 20)
    sz = g.sizes()
    assert sz["n_contigs"] == sz["n_starts"] > 0 and sz["n_branch"] > 0


def test_reads_beyond_2_gib_use_64bit_stamps():
    """15M x 150 bp = 2.25 GB of reads: byte offsets no longer fit the 32-bit stamp, the build switches to
    64-bit stamps (smaller LDS staging); same invariants as at the BASELINE size."""
    n, L, k, G = 15_000_000, 150, 31, 75_000_000
    g = _dbg.Graph()
    g.synth_reads(1, G, n, L, 0.0)
    assert g.sizes()["n_bytes"] >= 1 << 31
    g.build(k)
    sz = g.sizes()
    assert sz["n_kmer_instances"] == n * (L - k + 1) and sz["n_edge_instances"] == n * (L - k)
    keys, stamps, counts, flags = g.export_nodes()
    assert int(counts.sum(dtype=np.uint64)) == sz["n_edge_instances"]
    assert np.unique(keys).size == keys.size and sz["n_nodes"] <= G - k + 1
    pos = stamps >> np.uint64(1)
    # (a 32-bit stamp would wrap for offsets >= 2^31 and win the min: the re-encoding check below would fail)
    assert np.array_equal((stamps & np.uint64(1)) == 0, pos % np.uint64(L) == 0)
    rp, col, cnt = g.export_csr()
    assert int(rp[-1]) == sz["n_edges"] == int((counts != 0).sum())
    succ = g.export_succ()
    assert np.array_equal(col, succ[counts != 0]) and np.array_equal(cnt, counts[counts != 0])
    mask = np.uint64((1 << (2 * k)) - 1)
    for c in range(4):
        has = counts[:, c] != 0
        assert np.array_equal(keys[succ[has, c]], ((keys[has] << np.uint64(2)) | np.uint64(c)) & mask)
    # first occurrences re-encoded from the reads
    reads, _ = g.copy_reads()
    idx = np.linspace(0, keys.size - 1, 20000).astype(np.int64)
    code = ((reads[(pos[idx][:, None] + np.arange(k, dtype=np.uint64)[None, :]).astype(np.int64)] >> 1) & 3).astype(np.uint64)
    shifts = (2 * (k - 1 - np.arange(k, dtype=np.uint64))).astype(np.uint64)
    assert np.array_equal((code << shifts[None, :]).sum(axis=1, dtype=np.uint64), keys[idx])


def test_per_range_edge_order_and_pull_reads_with_64bit_stamps():
    """The per-range kernels behind dbg_refine_edge_order / dbg_mark_pull_reads in their 64-bit-stamp form (reads beyond
    2 GiB): same rank bytes and the same pulled reads as the passes over the reads."""
    import ctypes as C
    n, L, k, G = 15_000_000, 150, 31, 75_000_000
    res = []
    for streaming in (0, 1):
        g = _dbg.Graph()
        g.set_option("refine_streaming", streaming)
        g.synth_reads(3, G, n, L, 0.004)
        assert g.sizes()["n_bytes"] >= 1 << 31
        g.build(k)
        g.refine_edge_order()
        nn = g.sizes()["n_nodes"]
        mc, fs = np.empty(nn, dtype=np.uint8), np.empty(nn, dtype=np.uint8)
        g._chk(g._lib.dbg_export_orders(g._h, mc.ctypes.data_as(C.c_void_p), fs.ctypes.data_as(C.c_void_p)))
        order = g.export_dict_order()  # node ids inside a bucket differ from run to run: compare in first-occurrence order
        mc, fs = mc[order], fs[order]
        g.prune(2)
        g.remove_tips()
        g.mark_pull_reads()
        sz = g.sizes()
        res.append((mc, fs, g.export_pull_reads(), sz["n_pull_reads"], sz["n_branch"]))
        g.close()
    assert res[0][4] == res[1][4] > 1000 and res[0][3] == res[1][3] > 1000
    assert np.array_equal(res[0][2], res[1][2])
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_bench_digest_on_the_device_equals_the_oracle_digest():
    """bench.py compares the GPU graph with the multi-threaded CPU baseline at full size through a 64-bit digest over
    (k-mer, stamp, 4 counts) of every node; here the device-side digest (torch int64 arithmetic) against the numpy one
    of the C oracle's export, and against orc_build_mt's own."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    reads = synth.reads_ascii(21, 200_000, 40_000, 100, 0.01)
    off = np.arange(0, reads.size + 1, 100, dtype=np.uint64)
    g = _dbg.Graph()
    g.set_reads(reads.reshape(-1), off)
    g.build(31)
    want = orc_c.build(reads.reshape(-1), off, 31)
    mt = orc_c.build_mt(reads.reshape(-1), off, 31, 4)
    d = bench.node_digest_gpu(g)
    assert d == orc_c.digest(want["keys"], want["stamps"], want["counts"]) == mt["digest"]
    assert g.sizes()["n_edges"] == mt["n_edges"] and g.sizes()["n_nodes"] == mt["n_nodes"]


def test_bench_line_contract_at_a_small_size(tmp_path):
    """bench.py end to end in a child process (200 k reads): ONE JSON line with the driver's contract fields, the
    roofline and cpu_baseline blocks, the CPU/GPU digest comparison and the FASTA-ingest extra."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--reads", "200000", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "k-mers/s" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 200000 * 120 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None  # the committed PMC profile describes the full-size workload only
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["same_graph_as_gpu"] is True and c["value"] > 0
    assert d["extras"]["fasta_ingest"]["same_reads_and_graph"] is True
    assert d["extras"]["traversal_in_parts_sizes"]["same_as_single_graph"] is True
    assert d["first_build_ms"] > 0 and c["one_thread"]["value"] > 0 and c["every_thread_scans_all_reads"]["value"] > 0
    assert "orc_build_mt_partitioned" in c["sample"]


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_rehearsal(scaling):
    """The N > 1 leg of bench.py as the driver launches it (torch.distributed.run, one process per rank), rehearsed on
    this box's one GPU: both ranks on cuda:0, gloo instead of RCCL (RCCL refuses two ranks on one device).  Checks the
    line's fields, not its speed."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BENCH_BACKEND="gloo", BENCH_SAME_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--scaling", scaling] + (["--total-reads", "200000"] if scaling == "strong" else ["--reads", "100000"])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["config"]["reads_total"] == 200000
    assert "shard x2" in d["config"]["parallelism"]
    assert abs(d["value"] - 200000 * 120 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]  # whole-job k-mers / max-over-ranks time
