"""The CPU oracle (oracle/dbg_oracle.py) against vectors produced by the real reference.

These run on CPU (-m "not gpu").  They are what makes the oracle a trusted checker
for the HIP parity tests.
"""
import io
import contextlib
import json
import os

import pytest

from conftest import GOLDEN, case_reads, golden_case_names, load_golden
from golden_util import FIELDS, canonical, part_digests
from oracle import dbg_oracle as orc


def run_oracle(reads, k, threshold, final):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g, pull, branch, pulled, ect = orc.construct_graph(list(reads), k, threshold=threshold, final=final)
        contigs = orc.output_contigs(g, branch, pulled)
    res = canonical(g, pull, branch, pulled, ect, contigs)
    res["stdout"] = buf.getvalue()
    return res


@pytest.mark.parametrize("name", golden_case_names())
def test_oracle_matches_reference_vectors(name):
    case = load_golden(name)
    inp = case["inputs"]
    res = run_oracle(case_reads(case), inp["k"], inp["threshold"], inp["final"])
    if "result" in case:
        for f in FIELDS + ("stdout",):
            assert res[f] == case["result"][f], f"{name}: field {f} differs from the reference"
    got = part_digests(res)
    for key, want in case["digests"].items():
        assert got[key] == want, f"{name}: digest {key} differs from the reference"


@pytest.mark.parametrize("family,n_min", [("fuzz_small", 400), ("fuzz_peptide", 240), ("fuzz_wide", 160), ("fuzz_peptide_wide", 120)])
def test_oracle_fuzz_family(family, n_min):
    with open(os.path.join(GOLDEN, family + ".json")) as fh:
        cases = json.load(fh)
    assert len(cases) >= n_min
    for i, case in enumerate(cases):
        inp = case["inputs"]
        res = run_oracle(inp["reads"], inp["k"], inp["threshold"], inp["final"])
        for f in FIELDS + ("stdout",):
            assert res[f] == case["result"][f], f"fuzz case {i}: field {f} differs ({inp})"


@pytest.mark.parametrize("name", ["driver_dna_k5_8", "driver_dna_k12_15", "driver_peptide_k3_5", "driver_dna_k30_34", "driver_peptide_k10_14"])
def test_oracle_multi_k_driver(name):
    case = load_golden(name)
    inp = case["inputs"]
    final, trace = orc.assemble(inp["reads"], inp["k_lowerlimit"], inp["k_upperlimit"], inp["threshold"])
    assert final == case["result"]["final_contigs"]
    assert {str(k): v for k, v in trace.items()} == case["result"]["trace"]


def test_oracle_unused_helpers_against_reference_vectors():
    """get_kmers / get_graph_from_kmers (debruijn.py:35-95; the pipeline never calls them)."""
    import json
    import os
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, "aux_kmers.json")) as fh:
        cases = json.load(fh)
    assert len(cases) >= 300
    for c in cases:
        work = list(c["sequences"])
        kmers = orc.get_kmers(work, c["k"])
        assert kmers == c["kmers"] and work == c["sequences_after"], c["sequences"]
        V, E = orc.get_graph_from_kmers(list(kmers), c["k"])
        assert [[v, V[v].indegree, V[v].outdegree] for v in V] == c["vertices"]
        assert [[v, list(E[v])] for v in E] == c["edges"]
