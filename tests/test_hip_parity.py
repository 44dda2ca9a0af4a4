"""HIP path vs the oracle and the reference's golden vectors (run with -m gpu on an MI355X).

Every call goes through the C ABI (ctypes -> libdbg_hip.so).  Integer/byte/index work: bit-exact,
orders included: dict insertion order, Counter.most_common tie order of successor lists, the append
order of already_pull_out and the contig order in both modes.
"""
import contextlib
import io
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, case_reads, golden_case_names, load_golden
from golden_util import canonical
from oracle import dbg_oracle as orc

pytestmark = pytest.mark.gpu

DNA = set("ACGT")


def is_dna(reads):
    return all(set(r) <= DNA for r in reads)


def run_product(reads, k, threshold, final):
    import debruijn as prod
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g, pull, branch, pulled, ect = prod.construct_graph(list(reads), k, threshold=threshold, final=final)
        contigs = prod.output_contigs(g, branch, pulled)
    res = canonical(g, pull, branch, pulled, ect, contigs)
    res["stdout"] = buf.getvalue()
    res["scores"] = list(contigs.scores)
    res["ect"] = ect
    return res


def run_oracle(reads, k, threshold, final):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g, pull, branch, pulled, ect = orc.construct_graph(list(reads), k, threshold=threshold, final=final)
        contigs = orc.output_contigs(g, branch, pulled)
    res = canonical(g, pull, branch, pulled, ect, contigs)
    res["stdout"] = buf.getvalue()
    res["scores"] = [orc.get_score(ect, c, k) for c in contigs]
    return res


def assert_same(got, want, final, tag):
    assert got["vertices"] == want["vertices"], f"{tag}: vertices (label, indegree, outdegree, dict order)"
    assert got["edges"] == want["edges"], f"{tag}: edges (keys, dict order, successor lists in most_common order)"
    assert got["edge_count_table"] == want["edge_count_table"], f"{tag}: edge counts (values and insertion order)"
    assert got["branch_kmer"] == want["branch_kmer"], f"{tag}: branch_kmer"
    assert got["already_pull_out"] == want["already_pull_out"], f"{tag}: already_pull_out (append order)"
    assert got["pull_out_read"] == want["pull_out_read"], f"{tag}: pull_out_read"
    assert got["contigs"] == want["contigs"], f"{tag}: contigs"
    assert got["scores"] == want["scores"], f"{tag}: getScore"
    assert got["stdout"] == want["stdout"], f"{tag}: stdout lines"


@pytest.mark.parametrize("name", golden_case_names())
def test_golden_case(name):
    case = load_golden(name)
    reads = case_reads(case)
    inp = case["inputs"]
    got = run_product(reads, inp["k"], inp["threshold"], inp["final"])
    want = run_oracle(reads, inp["k"], inp["threshold"], inp["final"])
    assert_same(got, want, inp["final"], name)
    s = case["summary"]  # numbers produced by the reference itself
    assert len(got["vertices"]) == s["n_vertices"]
    assert len(got["edge_count_table"]) == s["n_edge_names"]
    assert sum(c for _, c in got["edge_count_table"]) == s["sum_edge_counts"]
    assert len(got["already_pull_out"]) == s["n_pulled"]
    assert len(got["pull_out_read"]) == s["n_pull_reads"]
    assert len(got["contigs"]) == s["n_contigs"]
    if "result" in case:  # full reference output available
        ref = dict(case["result"])
        ref["scores"] = want["scores"]
        assert_same(got, ref, inp["final"], name + " (reference)")


@pytest.mark.parametrize("name", [n for n in golden_case_names() if n.startswith("dna_")])
def test_golden_case_with_64_bit_stamps(name, monkeypatch):
    """What reads of 2 GiB and more run (64-bit stamps in the records, the LDS tables and the node arrays: k_sk_count<uint64_t>,
    k_wsk_count<uint64_t> -- the two-word engine took the global-table fallback there before round 3), forced at the size of the
    reference's vectors: the whole path must give the same dictionaries, orders and contigs."""
    monkeypatch.setenv("DBG_STAMP64", "1")
    case = load_golden(name)
    reads = case_reads(case)
    inp = case["inputs"]
    got = run_product(reads, inp["k"], inp["threshold"], inp["final"])
    want = run_oracle(reads, inp["k"], inp["threshold"], inp["final"])
    assert_same(got, want, inp["final"], name)


@pytest.mark.parametrize("family,n_min", [("fuzz_small", 400), ("fuzz_peptide", 240), ("fuzz_wide", 160), ("fuzz_peptide_wide", 120)])
def test_fuzz_family_against_reference_vectors(family, n_min):
    """fuzz_small: sub-alphabets of ACGT (2-bit path); fuzz_peptide: amino acids (generic 5-bit path);
    fuzz_wide: 32 <= k <= 63 (two-word k-mers) with repeats longer than k, tips and tandem-repeat cycles;
    fuzz_peptide_wide: peptides / N-containing / mixed-case reads at k = 12..40 (tables keyed by reference)."""
    with open(os.path.join(GOLDEN, family + ".json")) as fh:
        cases = json.load(fh)
    n = 0
    for i, case in enumerate(cases):
        inp = case["inputs"]
        got = run_product(inp["reads"], inp["k"], inp["threshold"], inp["final"])
        ref = dict(case["result"])
        ect = dict(map(tuple, ref["edge_count_table"]))
        ref["scores"] = [orc.get_score(ect, c, inp["k"]) for c in ref["contigs"]]
        assert_same(got, ref, inp["final"], f"{family} {i} {inp}")
        n += 1
    assert n >= n_min


@pytest.mark.parametrize("name", [n for n in golden_case_names()
                                  if n.startswith(("peptide_k3", "hand_tips_order_t2", "hand_cycle_t2", "dna_small", "dna_med_k31_e1_t2",
                                                   "dna_med_k63_e1"))])
def test_lazy_views_equal_the_dicts(name, monkeypatch):
    """Above LAZY_MIN_NODES construct_graph returns Mapping views over the exported arrays (dicts are
    infeasible at 10^8 nodes); materialised they equal the reference's dicts, order included."""
    import debruijn as prod
    monkeypatch.setattr(prod, "LAZY_MIN_NODES", 0)
    case = load_golden(name)
    reads = case_reads(case)
    inp = case["inputs"]
    with contextlib.redirect_stdout(io.StringIO()):
        (V, E), pull, branch, pulled, ect = prod.construct_graph(list(reads), inp["k"], threshold=inp["threshold"],
                                                                final=inp["final"])
        contigs = prod.output_contigs((V, E), branch, pulled)
    assert not isinstance(V, dict) and not isinstance(E, dict) and not isinstance(ect, dict)
    got = canonical((V, E), pull, branch, pulled, ect, contigs)
    got["stdout"], got["scores"] = None, list(contigs.scores)
    want = run_oracle(reads, inp["k"], inp["threshold"], inp["final"])
    want["stdout"] = None
    assert_same(got, want, inp["final"], name + " (lazy)")
    # the Mapping protocol
    wV = dict((v, (i, o)) for v, i, o in want["vertices"])
    assert len(V) == len(wV) and len(E) == len(want["edges"]) and len(ect) == len(want["edge_count_table"])
    assert E == dict((v, s) for v, s in want["edges"]) and ect == dict(map(tuple, want["edge_count_table"]))
    assert list(ect.items()) == list(map(tuple, want["edge_count_table"]))
    k = inp["k"]
    for bogus in ("", "?" * k, "A" * (k + 1), None, 7):
        assert bogus not in V and bogus not in E and bogus not in ect
        with pytest.raises(KeyError):
            V[bogus]
    for v in list(wV)[:50]:
        assert v in V and (V[v].label, V[v].indegree, V[v].outdegree) == (v, *wV[v])
    for v in pulled:
        assert v in V and v not in E


def test_other_alphabets_take_the_generic_path():
    """N, lower case, digits ...: distinct characters like in the reference (str slices), never dropped."""
    for reads, k in ((["ACGTN", "ACGTA", "NNACG"], 3), (["acgtacgt", "ACGTacgt"], 3), (["0120120", "1201"], 2)):
        got = run_product(reads, k, 2, False)
        want = run_oracle(reads, k, 2, False)
        assert_same(got, want, False, str(reads))


def test_alphabet_limits_fail_loudly():
    import _dbg
    import debruijn as prod
    with pytest.raises(ValueError):      # more than 32 distinct characters
        prod.construct_graph(["".join(chr(48 + i) for i in range(40))], 3)
    with pytest.raises(ValueError):      # k-mers longer than 63 characters
        prod.construct_graph(["ACGT" * 40], 64)
    g = _dbg.Graph()
    b = np.frombuffer(("".join(chr(48 + i) for i in range(40)) * 2).encode(), dtype=np.uint8)
    g.set_reads(b, np.array([0, b.size], dtype=np.uint64))
    with pytest.raises(_dbg.AlphabetError):
        g.build(12)
    # peptides beyond one packed word (k >= 12) take the by-reference tables: same results as the oracle
    reads = ["EVQLVESGGGLVQPGGSLRLSCAAS", "GGGLVQPGGSLRLSCAASGFTFS", "EVQLVESGGGLVQPGGSLRL"]
    for k in (11, 12, 16, 20):
        assert_same(run_product(reads, k, 2, False), run_oracle(reads, k, 2, False), False, f"peptides k={k}")


def test_empty_and_degenerate_inputs():
    import debruijn as prod
    for reads in ([], [""], ["A"], ["ACG"], ["", "", "AC"]):
        with contextlib.redirect_stdout(io.StringIO()):
            g, pull, branch, pulled, ect = prod.construct_graph(reads, 3)
            contigs = prod.output_contigs(g, branch, pulled)
        assert len(g[0]) == 0 and len(g[1]) == 0 and not pull and not branch and not pulled and not ect
        assert contigs == []


def test_csr_matches_node_table():
    import _dbg
    import synth
    reads = synth.reads_ascii(5, 20000, 3000, 100, 0.01)
    g = _dbg.Graph()
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, 100, dtype=np.uint64))
    g.build(21)
    keys, stamps, counts, flags = g.export_nodes()
    succ = g.export_succ()
    rp, col, cnt = g.export_csr()
    sz = g.sizes()
    assert sz["n_kmer_instances"] == 3000 * 80 and sz["n_edge_instances"] == 3000 * 79
    assert int(counts.sum()) == sz["n_edge_instances"]
    deg = (counts != 0).sum(axis=1)
    assert np.array_equal(np.diff(rp.astype(np.int64)), deg)
    assert int(rp[-1]) == sz["n_edges"] == int(deg.sum())
    assert np.array_equal(col, succ[counts != 0]) and np.array_equal(cnt, counts[counts != 0])
    # successor ids point at the shifted k-mer
    mask = np.uint64((1 << 42) - 1)
    for code in range(4):
        has = counts[:, code] != 0
        want = ((keys[has] << np.uint64(2)) | np.uint64(code)) & mask
        assert np.array_equal(keys[succ[has, code]], want)
    assert len(np.unique(keys)) == keys.size and len(np.unique(stamps)) == stamps.size


@pytest.mark.parametrize("name", [n for n in golden_case_names() if "nonfinal" in n or n.startswith("dna_")])
def test_walk_by_pointer_jumping(name, monkeypatch):
    """Non-final walk through the pointer-jumping path (the one used at scale), forced on at test sizes."""
    case = load_golden(name)
    reads = case_reads(case)
    inp = case["inputs"]
    if inp["final"]:
        pytest.skip("non-final only")
    monkeypatch.setenv("DBG_WALK_JUMP_MIN", "0")
    got = run_product(reads, inp["k"], inp["threshold"], False)
    want = run_oracle(reads, inp["k"], inp["threshold"], False)
    assert got["contigs"] == want["contigs"]
    assert got["scores"] == want["scores"]


def test_walk_index_without_text():
    """max_chars too small: the walk still returns the contig index (lengths, scores, start stamps)."""
    import _dbg
    import synth
    reads = synth.reads_ascii(8, 20000, 2500, 100, 0.01)
    g = _dbg.Graph()
    g.set_option("walk_jump_min_nodes", 0)
    g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, 100, dtype=np.uint64))
    g.build(21)
    g.prune(2)
    g.remove_tips()
    g.walk(False)
    off, chars, score, stamp, seq = g.export_contigs()
    g.walk(False, max_chars=1000)
    assert g.sizes()["contigs_materialised"] == 0
    off2, score2, stamp2, seq2 = g.export_contig_index()
    a, b = np.argsort(stamp), np.argsort(stamp2)  # the start list is collected in no particular order
    assert np.array_equal(stamp[a], stamp2[b]) and np.array_equal(score[a], score2[b])
    assert np.array_equal(np.diff(off.astype(np.int64))[a], np.diff(off2.astype(np.int64))[b])
    with pytest.raises(_dbg.DbgError):
        g.export_contigs()
    # ... and every text on demand (dbg_export_contig_text), equal to the materialised walk's
    text = chars.tobytes()
    want = {(int(stamp[i]), int(seq[i])): text[int(off[i]):int(off[i + 1])] for i in range(stamp.size)}
    for i in range(stamp2.size):
        assert g.export_contig_text(i, int(off2[i + 1] - off2[i])) == want[(int(stamp2[i]), int(seq2[i]))]
    with pytest.raises(_dbg.DbgError):
        g.export_contig_text(stamp2.size, 10)       # out of range
    if stamp2.size:
        with pytest.raises(_dbg.DbgError):
            g.export_contig_text(0, 0)              # buffer smaller than the contig
    g.prune(2)                                      # the graph changed: the old index is gone
    with pytest.raises(_dbg.DbgError):
        g.export_contig_text(0, 1 << 20)


@pytest.mark.parametrize("name", ["dna_med_k21_e1_t2", "dna_med_k63_e1_t2", "dna_small_k9_e1_t1", "peptide_k4_nonfinal",
                                  "hand_cycle_rho_t1_nonfinal", "hand_short_reads_t1_nonfinal"])
def test_contigs_fetched_on_demand_in_the_drop_in(name, monkeypatch):
    """Contig text larger than the device budget: output_contigs returns a lazy sequence with the same contents."""
    import debruijn as prod
    case = load_golden(name)
    reads = case_reads(case)
    inp = case["inputs"]
    if inp["final"]:
        pytest.skip("non-final only")
    monkeypatch.setenv("DBG_WALK_JUMP_MIN", "0")
    want = run_product(reads, inp["k"], inp["threshold"], False)
    monkeypatch.setattr(prod, "MAX_CONTIG_CHARS", 1)
    g, pull, branch, already, _ = prod.construct_graph(list(reads), inp["k"], threshold=inp["threshold"], final=False)
    lazy = prod.output_contigs(g, branch, already)
    assert isinstance(lazy, prod.LazyContigs) or len(want["contigs"]) == 0
    assert list(lazy) == want["contigs"] and list(lazy.scores) == want["scores"]
    if len(lazy):
        assert lazy[-1] == want["contigs"][-1] and lazy[0:2] == want["contigs"][0:2]
        assert lazy.lengths == [len(c) for c in want["contigs"]]


def test_unused_helpers_against_reference_vectors():
    """get_kmers (k-mer enumeration on the device) / get_graph_from_kmers of the drop-in module."""
    import debruijn as prod
    with open(os.path.join(GOLDEN, "aux_kmers.json")) as fh:
        cases = json.load(fh)
    for c in cases:
        work = list(c["sequences"])
        kmers = prod.get_kmers(work, c["k"])
        assert kmers == c["kmers"] and work == c["sequences_after"], c["sequences"]
        V, E = prod.get_graph_from_kmers(list(kmers), c["k"])
        assert [[v, V[v].indegree, V[v].outdegree] for v in V] == c["vertices"]
        assert [[v, list(E[v])] for v in E] == c["edges"]
    V, E = prod.get_graph_from_kmers(["AAA", "AAC", "AAA"], 3)   # a repeated k-mer starts over, like the reference
    Vo, Eo = orc.get_graph_from_kmers(["AAA", "AAC", "AAA"], 3)
    assert [(v, V[v].indegree, V[v].outdegree, E[v]) for v in V] == [(v, Vo[v].indegree, Vo[v].outdegree, Eo[v]) for v in Vo]
