"""Read-support scores: the oracle's restatement against vectors produced by the reference's
findSupportReadScore (IV_sortOutputs.py:10-15; oracle/make_golden.py support), CPU only."""
import json
import os

from oracle import dbg_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    with open(os.path.join(HERE, "golden", "support_scores.json")) as fh:
        cases = json.load(fh)
    for c in cases:
        c["table"] = {r: (float.fromhex(s) if isinstance(s, str) else s) for r, s in zip(c["reads"], c["scores"])}
        c["expect"] = [float.fromhex(w) if isinstance(w, str) else w for w in c["want"]]
    return cases


def test_restatement_equals_the_reference_bit_for_bit():
    cases = load_cases()
    assert len(cases) == 60
    for c in cases:
        got = [orc.find_support_read_score(x, c["table"]) for x in c["contigs"]]
        assert got == c["expect"] and [type(g) for g in got] == [type(w) for w in c["expect"]]
