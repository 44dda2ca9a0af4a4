"""Canonical form + digests shared by the oracle tests and the HIP parity tests.

Must stay byte-compatible with ``canonical``/``part_digests`` in oracle/make_golden.py.
"""
import hashlib
import json


def digest(obj):
    return hashlib.sha256(json.dumps(obj, sort_keys=True, separators=(",", ":")).encode()).hexdigest()


def canonical(g, pull, branch, pulled, ect, contigs):
    V, E = g
    return {
        "vertices": [[v, V[v].indegree, V[v].outdegree] for v in V],
        "edges": [[v, list(E[v])] for v in E],
        "pull_out_read": list(pull),
        "branch_kmer": list(branch),
        "already_pull_out": list(pulled),
        "edge_count_table": [[n, c] for n, c in ect.items()],
        "contigs": list(contigs),
    }


FIELDS = ("vertices", "edges", "pull_out_read", "branch_kmer", "already_pull_out",
          "edge_count_table", "contigs")


def part_digests(res):
    d = {}
    for key in FIELDS:
        d[key] = digest(res[key])
        d[key + "_sorted"] = digest(sorted(res[key], key=lambda x: json.dumps(x)))
    d["edges_as_sets"] = digest(sorted([[v, sorted(s)] for v, s in res["edges"]]))
    return d
