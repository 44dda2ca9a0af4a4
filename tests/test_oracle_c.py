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


def test_multithreaded_build_equals_the_single_threaded_one():
    """bench.py's cpu_baseline (orc_build_mt_partitioned, and orc_build_mt beside it: hash slices over threads) must build the same graph as orc_build."""
    import numpy as np
    import synth
    reads = synth.reads_ascii(5, 30000, 6000, 100, 0.01)
    off = np.arange(0, reads.size + 1, 100, dtype=np.uint64)
    for k in (5, 21, 31):
        a = orc_c.build(reads.reshape(-1), off, k)
        for threads in (1, 3, 8):
            for once in (False, True):  # every thread scans everything / k-mers partitioned once (bench.py uses the latter)
                b = orc_c.build_mt(reads.reshape(-1), off, k, threads, partition_once=once)
                assert b["n_nodes"] == a["n_nodes"] and b["n_kmer_instances"] == a["n_kmer_instances"]
                assert b["n_edge_instances"] == a["n_edge_instances"] and b["n_edges"] == int((a["counts"] != 0).sum())
                assert b["digest"] == orc_c.digest(a["keys"], a["stamps"], a["counts"])
