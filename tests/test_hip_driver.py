"""Multi-k driver (II_assembleFromReads.py:56-75) and FASTA surface on the GPU path."""
import contextlib
import io
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, load_golden
from oracle import dbg_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["driver_dna_k5_8", "driver_dna_k12_15", "driver_peptide_k3_5", "driver_dna_k30_34", "driver_peptide_k10_14"])
def test_driver_matches_reference_vectors(name):
    import II_assembleFromReads as drv
    case = load_golden(name)
    inp = case["inputs"]
    with contextlib.redirect_stdout(io.StringIO()):
        final = drv.assemble(list(inp["reads"]), inp["k_lowerlimit"], inp["k_upperlimit"], inp["threshold"])
    assert final == case["result"]["final_contigs"]


def test_driver_nonfinal_rounds_exact():
    """Every non-final k reproduces the reference's sorted contigs and pull-out reads exactly."""
    import debruijn as prod
    case = load_golden("driver_dna_k12_15")
    inp = case["inputs"]
    seqs = list(inp["reads"])
    for k in range(inp["k_lowerlimit"], inp["k_upperlimit"]):
        with contextlib.redirect_stdout(io.StringIO()):
            g, pull, branch, pulled, ect = prod.construct_graph(seqs, k, threshold=inp["threshold"])
            contigs = prod.output_contigs(g, branch, pulled)
        scores = prod.get_score_device(contigs)
        assert scores == [orc.get_score(ect, c, k) for c in contigs]
        order = sorted(range(len(contigs)), key=lambda i: scores[i], reverse=True)
        seqs = [contigs[i] for i in order]
        want = case["result"]["trace"][str(k)]
        assert seqs == want["contigs"]
        assert pull == want["pull_out_read"]
        seqs.extend(pull)


def test_cli_fasta_in_contigs_out(tmp_path):
    """-froot CLI: setting.json + input_reads.fasta in, append-mode {froot}/{froot}.fasta out."""
    import synth
    reads = synth.reads_list(41, 3000, 400, 80, 0.005)
    froot = "froot_case"
    synth.write_froot(str(tmp_path / froot), reads, 15, 17, threshold=2)
    env = dict(os.environ, PYTHONPATH=PKG)
    for _ in range(2):  # the reference opens the output in 'a+' mode: a second run appends
        subprocess.check_call([sys.executable, os.path.join(PKG, "II_assembleFromReads.py"), "-froot", froot],
                              cwd=str(tmp_path), env=env, stdout=subprocess.DEVNULL)
    out = (tmp_path / froot / f"{froot}.fasta").read_text().splitlines()
    with contextlib.redirect_stdout(io.StringIO()):
        want, _ = orc.assemble(reads, 15, 17, 2)
    heads, seqs = out[0::2], out[1::2]
    assert len(seqs) == 2 * len(want)
    assert seqs[:len(want)] == want and seqs[:len(want)] == seqs[len(want):]
    assert heads[0] == ">SEQUENCE_0_17mer" and heads[len(want) - 1] == f">SEQUENCE_{len(want) - 1}_17mer"


def test_read_reads_semantics(tmp_path):
    import debruijn as prod
    p = tmp_path / "x.fasta"
    p.write_text(">a\nACGT\nTTGA  \n>b\n\n>c\nGG\n")
    assert prod.read_reads(str(p)) == orc.read_reads(str(p)) == ["ACGT", "TTGA", "", "GG"]


FASTA_CASES = {
    "plain": ">r0\nACGTACGTTG\n>r1\nTTGACCA\n",
    "no_final_newline": ">r0\nACGTACGTTG\n>r1\nTTGACCA",
    "crlf": ">r0\r\nACGTACGTTG\r\n>r1\r\nTTGACCA\r\n",
    "lone_cr": ">r0\rACGTACGTTG\r>r1\rTTGACCA\r",
    "blank_and_multiline": ">r0\nACGT\nACGTTG\n\n>r1\n\nTTGACCA\n\n",
    "trailing_space": ">r0 some text\nACGTACGTTG  \t\n>r1\nTTGACCA \n",
    "gt_inside": "ACG>TT\n>hdr\n >notheader\nAC\n",
    "empty": "",
    "only_headers": ">a\n>b\n",
    "double_cr": "ACGT\r\r\nTTGA\n",
}


@pytest.mark.parametrize("name", sorted(FASTA_CASES))
def test_device_fasta_ingest_matches_read_reads(name, tmp_path):
    """dbg_set_reads_fasta == read_reads (debruijn.py:22-32) on awkward files, through the C ABI."""
    import debruijn as prod
    p = tmp_path / (name + ".fasta")
    p.write_bytes(FASTA_CASES[name].encode())
    want = orc.read_reads(str(p))            # Python text mode: the reference's semantics
    assert prod.read_reads(str(p)) == want
    dev = prod.read_reads_device(str(p))
    assert len(dev) == len(want)
    assert list(dev) == want
    if want:
        assert dev[-1] == want[-1] and dev[0:2] == want[0:2]


def test_construct_graph_from_device_reads(tmp_path):
    import debruijn as prod
    import synth
    reads = synth.reads_list(51, 4000, 500, 90, 0.01)
    p = tmp_path / "input_reads.fasta"
    synth.write_fasta(str(p), reads)
    with contextlib.redirect_stdout(io.StringIO()):
        a = prod.construct_graph(reads, 21, threshold=2)
        ca = prod.output_contigs(a[0], a[2], a[3])
        dev = prod.read_reads_device(str(p))
        b = prod.construct_graph(dev, 21, threshold=2)
        cb = prod.output_contigs(b[0], b[2], b[3])
    assert list(a[0][0]) == list(b[0][0]) and a[0][1] == b[0][1]
    assert a[1] == b[1] and list(a[2]) == list(b[2]) and list(a[3]) == list(b[3]) and a[4] == b[4]
    assert list(ca) == list(cb)


def test_device_reads_hold_one_character_per_byte(tmp_path):
    """Bytes >= 0x80 in a FASTA file: the device reads decode as latin-1 (one byte, one character) like _pack_reads, the
    node labels and the contig text, so reads[i], pull_out_read and a graph built from list(reads) stay consistent."""
    import debruijn as prod
    p = tmp_path / "hi.fasta"
    p.write_bytes(b">h\nAC\xe9GTAC\xe9GA\n>i\nC\xe9GTT\n")
    dev = prod.read_reads_device(str(p))
    assert list(dev) == ["AC\xe9GTAC\xe9GA", "C\xe9GTT"]
    with contextlib.redirect_stdout(io.StringIO()):
        a = prod.construct_graph(dev, 3, threshold=1)
        ca = list(prod.output_contigs(a[0], a[2], a[3]))
        b = prod.construct_graph(list(dev), 3, threshold=1)
        cb = list(prod.output_contigs(b[0], b[2], b[3]))
    assert list(a[0][0]) == list(b[0][0]) and "C\xe9G" in a[0][0] and a[1] == b[1] and ca == cb


def test_output_contigs_refuses_a_graph_that_was_replaced(tmp_path):
    """construct_graph(DeviceReads) builds into the reads' own handle: a second call replaces the first graph on the
    device, and output_contigs on the first result must raise instead of walking the new graph."""
    import debruijn as prod
    import synth
    reads = synth.reads_list(52, 3000, 300, 80, 0.01)
    p = tmp_path / "input_reads.fasta"
    synth.write_fasta(str(p), reads)
    dev = prod.read_reads_device(str(p))
    with contextlib.redirect_stdout(io.StringIO()):
        first = prod.construct_graph(dev, 15, threshold=2)
        second = prod.construct_graph(dev, 17, threshold=2)
        with pytest.raises(ValueError, match="replaced"):
            prod.output_contigs(first[0], first[2], first[3])
        assert len(prod.output_contigs(second[0], second[2], second[3])) > 0


def test_sorted_fasta_from_the_device_equals_the_host_formatting():
    """dbg_export_sorted_fasta: the driver's sort (score descending, stable) and its record format
    (II_assembleFromReads.py:64-69), done on the device, against the same done on the host from the contig list."""
    import debruijn as prod
    import synth
    for seed, k, final in ((61, 15, True), (62, 21, False), (63, 9, True)):
        reads = synth.reads_list(seed, 3000, 400, 80, 0.01)
        with contextlib.redirect_stdout(io.StringIO()):
            g, pull, branch, pulled, ect = prod.construct_graph(reads, k, threshold=2, final=final)
            contigs = prod.output_contigs(g, branch, pulled)
        scores = prod.get_score_device(contigs)
        order = sorted(range(len(contigs)), key=lambda i: scores[i], reverse=True)
        want = "".join('>SEQUENCE_{}_{}mer\n{}\n'.format(i, k, contigs[j]) for i, j in enumerate(order))
        assert len(contigs) > 10 and contigs.sorted_fasta() == want


def test_lazy_views_load_on_first_use_and_refuse_a_replaced_graph(tmp_path, monkeypatch):
    """At scale vertices / edges / edge_count_table are views whose arrays leave the device when first asked for (the
    reference's driver never asks).  A view read later equals the dict built eagerly; a view whose graph a later
    construct_graph on the same resident reads replaced raises instead of showing the new graph."""
    import debruijn as prod
    import synth
    reads = synth.reads_list(71, 4000, 400, 80, 0.01)
    with contextlib.redirect_stdout(io.StringIO()):
        (V, E), pull, branch, pulled, ect = prod.construct_graph(reads, 15, threshold=2)   # dicts (small graph)
        monkeypatch.setattr(prod, "LAZY_MIN_NODES", 0)
        (Vl, El), pull_l, branch_l, pulled_l, ect_l = prod.construct_graph(reads, 15, threshold=2)
    assert Vl._s._arrays is None and len(Vl) == len(V)            # nothing exported yet, the size is known
    assert list(branch_l) == list(branch) and list(pulled_l) == list(pulled) and pull_l == pull
    contigs = prod.output_contigs((Vl, El), branch_l, pulled_l)
    assert Vl._s._arrays is None and len(contigs) > 0             # the walk does not need them either
    assert dict(El) == E and list(Vl) == list(V) and dict(ect_l.items()) == ect
    assert Vl._s._arrays is not None
    p = tmp_path / "input_reads.fasta"
    synth.write_fasta(str(p), reads)
    dev = prod.read_reads_device(str(p))
    with contextlib.redirect_stdout(io.StringIO()):
        first = prod.construct_graph(dev, 15, threshold=2)
        prod.construct_graph(dev, 17, threshold=2)
    with pytest.raises(RuntimeError, match="replaced"):
        len(first[0][1])      # edges needs the arrays of a graph that is gone
    assert len(first[0][0]) == len(V)   # the vertex count was known at construction


def test_export_marked_equals_the_host_selection():
    """dbg_export_marked (pulled nodes in pull order, branch nodes in dict order, selected and sorted on the device)
    against the same selection from the full exports; graphs keyed by reference refuse keys instead of faulting."""
    import _dbg
    import synth
    reads = synth.reads_ascii(81, 6000, 1500, 100, 0.02)
    for k in (21, 40):
        g = _dbg.Graph()
        g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, 100, dtype=np.uint64))
        g.build(k); g.refine_edge_order(); g.prune(2); g.remove_tips(); g.mark_pull_reads()
        keys, stamps, counts, flags = g.export_nodes()
        hi = g.export_keys_hi()
        ranks = g.export_pull_ranks()
        rows = np.nonzero(flags & _dbg.F_PULLED)[0]
        rows = rows[np.argsort(ranks[rows], kind="stable")]
        got = g.export_marked(_dbg.F_PULLED)
        assert rows.size > 0 and np.array_equal(got[0], rows) and np.array_equal(got[1], keys[rows])
        assert np.array_equal(got[2], hi[rows] if k > 32 else np.zeros(rows.size, np.uint64))
        rows = np.nonzero(flags & _dbg.F_BRANCH)[0]
        rows = rows[np.argsort(stamps[rows], kind="stable")]
        got = g.export_marked(_dbg.F_BRANCH)
        assert rows.size > 0 and np.array_equal(got[0], rows) and np.array_equal(got[1], keys[rows])
        g.close()
    rng = np.random.default_rng(9)
    aa = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    protein = rng.choice(aa, size=300)
    pep = []
    for _ in range(120):
        s0 = int(rng.integers(0, 300 - 60))
        r = protein[s0:s0 + 60].copy()
        m = rng.random(60) < 0.03
        r[m] = rng.choice(aa, size=int(m.sum()))
        pep.append(r)
    blob = np.concatenate(pep)
    g = _dbg.Graph()
    g.set_reads(blob, np.arange(0, blob.size + 1, 60, dtype=np.uint64))
    g.build(14); g.refine_edge_order(); g.prune(2); g.remove_tips()   # k = 14 peptides: tables keyed by reference
    assert g.sizes()["n_branch"] > 0
    with pytest.raises(_dbg.DbgError, match="no packed keys"):
        g.export_marked(_dbg.F_BRANCH)
    rows, kk, hh = g.export_marked(_dbg.F_BRANCH, keys=False)
    assert rows.size == g.sizes()["n_branch"] and kk is None


def test_device_reads_take_gathers_on_the_device(tmp_path):
    """DeviceReads.take == indexing one by one, for ragged reads, without the host copy of all reads."""
    import debruijn as prod
    rng = np.random.default_rng(17)
    reads = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 90)))) for _ in range(400)]
    p = tmp_path / "r.fasta"
    with open(p, "w") as f:
        for i, r in enumerate(reads):
            f.write(f">r{i}\n{r}\n")
    dev = prod.read_reads_device(str(p))
    idx = np.sort(rng.choice(400, size=37, replace=False))
    got = dev.take(idx)
    assert dev._host is None and got == [reads[i] for i in idx]
    assert dev.take([]) == [] and dev.take(np.arange(400)) == reads   # large selections go through the host copy
    with contextlib.redirect_stdout(io.StringIO()):
        a = prod.construct_graph(prod.read_reads_device(str(p)), 9, threshold=2)
        b = prod.construct_graph(reads, 9, threshold=2)
    assert a[1] == b[1] and list(a[2]) == list(b[2]) and list(a[3]) == list(b[3])
