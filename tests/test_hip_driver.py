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


@pytest.mark.parametrize("name", ["driver_dna_k5_8", "driver_dna_k12_15"])
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
