"""The C restatement (oracle/dbg_oracle.c) against the pinned Python oracle, CPU only."""
import numpy as np
import pytest

from conftest import case_reads, golden_case_names, load_golden
from oracle import dbg_oracle as orc
from oracle import orc_c

CODE_CHAR = "ACTG"


def decode(key, k, hi=0):
    v = (int(hi) << 64) | int(key)
    return "".join(CODE_CHAR[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def pack(reads):
    blob = "".join(reads).encode("ascii")
    off = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    return np.frombuffer(blob, dtype=np.uint8), off


@pytest.mark.parametrize("name", [n for n in golden_case_names() if not n.startswith("peptide")])
def test_c_oracle_matches_python_oracle(name):
    case = load_golden(name)
    reads = case_reads(case)
    k = case["inputs"]["k"]
    V, E = orc.graph_from_reads(reads, k)
    ect = orc.edge_count_table(E)
    b, off = pack(reads)
    res = orc_c.build(b, off, k)
    assert res["n_nodes"] == len(V) == case["summary"]["n_vertices"]
    labels = [decode(x, k, hi) for x, hi in zip(res["keys"], res["keys_hi"])]
    assert labels == list(V.keys())  # dict order
    assert [int(s) & 1 for s in res["stamps"]] == [V[v].indegree for v in V]
    got = {}
    for lab, c in zip(labels, res["counts"]):
        for code in range(4):
            if c[code]:
                got[lab + CODE_CHAR[code]] = int(c[code])
    assert got == dict(ect)
    assert res["n_edge_instances"] == sum(ect.values())
