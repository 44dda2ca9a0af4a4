"""Read-support scores on the device against vectors produced by the reference's findSupportReadScore
(IV_sortOutputs.py:10-15), through the C ABI; floating-point sums compare bit for bit."""
import numpy as np
import pytest

from test_oracle_support import load_cases

pytestmark = pytest.mark.gpu


def test_device_scores_equal_the_reference_bit_for_bit():
    import IV_sortOutputs as iv
    for c in load_cases():
        got = iv.support_read_scores(c["contigs"], c["table"])
        assert got == c["expect"] and [type(g) for g in got] == [type(w) for w in c["expect"]]
        one = iv.findSupportReadScore(c["contigs"][0], c["table"])
        assert one == c["expect"][0] and type(one) is type(c["expect"][0])


def test_sort_is_the_reference_sort():
    import IV_sortOutputs as iv
    c = max(load_cases(), key=lambda x: len(x["contigs"]))
    want = list(c["contigs"])
    table = dict(zip(c["contigs"], c["expect"]))
    want.sort(key=lambda x: table[x], reverse=True)  # IV_sortOutputs.py:55 with the reference's scores
    got, scores = iv.sort_contigs(c["contigs"], c["table"])
    assert got == want and scores == [table[x] for x in want]


def test_many_reads_sharing_a_prefix_and_many_hits():
    """All reads start with the same bytes (one chain in the index) and occur at almost every position of the contig: the
    hit list outgrows its first size (one slot per contig character) and is taken again at the exact size; a read found
    at many positions still counts once."""
    import IV_sortOutputs as iv
    from oracle import dbg_oracle as orc
    reads = {"A" * (7 + i): float(i) + 0.5 for i in range(64)}
    reads["A" * 7 + "C"] = 3
    contigs = ["A" * 2000, "A" * 40, "A" * 30 + "C", "ACGT", ""]
    want = [orc.find_support_read_score(x, reads) for x in contigs]
    assert want[0] > 0 and want[1] != want[0]
    assert iv.support_read_scores(contigs, reads) == want
