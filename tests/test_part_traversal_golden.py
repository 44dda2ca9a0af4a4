"""The traversal of a graph in parts (part_traversal.py) against the vectors the REFERENCE produced (tests/golden): every
non-final DNA case -- hand-made edge cases (cycles, tips whose result depends on the order, reads of length k, duplicates,
homopolymers), synthetic reads at k = 5...63 and the randomised families -- built in 1 and in 4 parts and traversed there:
branch_kmer, already_pull_out (append order), pull_out_read and the contigs (text, order, getScore) equal the reference's."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, case_reads, golden_case_names, load_golden
from oracle import dbg_oracle as orc

pytestmark = pytest.mark.gpu

DNA = set("ACGT")


def run_in_parts(reads, k, threshold, n_passes):
    import _dbg
    import part_traversal
    blob = np.frombuffer("".join(reads).encode("latin-1"), dtype=np.uint8)
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in reads], out=offs[1:])
    g = _dbg.Graph()
    g.set_reads(blob, offs)
    g.build_multipass(k, n_passes)
    t, flags, branch, pulled = part_traversal.construct_graph(g, k, threshold)
    ctg = part_traversal.output_contigs(t)
    texts = ctg.texts(range(len(ctg)))
    out = {"branch_kmer": branch, "already_pull_out": pulled, "pull_out_read": [r for r, f in zip(reads, flags) if f],
           "contigs": texts, "scores": ctg.scores.tolist()}
    g.close()
    return out


def check(reads, inp, ref, tag):
    if inp["final"] or not all(set(r) <= DNA for r in reads) or inp["threshold"] < 1:
        return False
    ect = dict(map(tuple, ref["edge_count_table"]))
    want_scores = [orc.get_score(ect, c, inp["k"]) for c in ref["contigs"]]
    for n_passes in (1, 4):
        got = run_in_parts(list(reads), inp["k"], inp["threshold"], n_passes)
        for key in ("branch_kmer", "already_pull_out", "pull_out_read", "contigs"):
            assert got[key] == ref[key], f"{tag} P={n_passes}: {key}"
        assert got["scores"] == want_scores, f"{tag} P={n_passes}: getScore"
    return True


def test_golden_cases_in_parts():
    n = 0
    for name in golden_case_names():
        case = load_golden(name)
        if "result" not in case:
            continue
        n += check(case_reads(case), case["inputs"], case["result"], name)
    assert n >= 20


@pytest.mark.parametrize("family,n_min", [("fuzz_small", 150), ("fuzz_wide", 40)])
def test_fuzz_families_in_parts(family, n_min):
    with open(os.path.join(GOLDEN, family + ".json")) as fh:
        cases = json.load(fh)
    n = 0
    for i, case in enumerate(cases):
        n += check(case["inputs"]["reads"], case["inputs"], case["result"], f"{family} {i} {case['inputs']['k']}")
    assert n >= n_min
