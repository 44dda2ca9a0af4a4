#!/usr/bin/env python3
"""Drop-in for the read-support scoring of the reference's ``IV_sortOutputs.py`` on the MI355X path.

``findSupportReadScore(contig, score_table)`` keeps the reference's name, arguments and result
(IV_sortOutputs.py:10-15: the scores of the reads -- the dict's keys -- that occur in the contig as a substring, added
in dict order); ``support_read_scores(contigs, score_table)`` scores a whole list in one device call, which is what a
caller with more than a handful of contigs wants: the reference tests reads x contigs substrings on the host, the
device indexes the reads once and looks every contig position up (csrc/dbg_support.h).  No CPU path: both raise
without the library and a GPU.  The rest of the reference's script -- the loader of the proprietary PSM score tables and
its CLI -- is outside the hot path and not mirrored here.
"""
import numpy as np

import _dbg
from debruijn import _pack_reads

_handle = None
# The last packed table: (the dict itself, a digest of its items, the packed arrays).  The reference calls
# findSupportReadScore once per contig with one table, so the packing is worth keeping -- but only for THAT dict with
# THAT content: the entry holds a reference to the dict (an id() alone is reused by CPython once a temporary dict is
# freed) and a digest of its items (the same dict may have been updated in place).
_packed = None


def _graph():
    global _handle
    if _handle is None:
        _handle = _dbg.Graph()
    return _handle


def _table_digest(score_table):
    """Order- and type-sensitive digest of the items: the dict order is the order the scores are added in, and an int
    score and the equal float give results of different types (1 == 1.0 and they hash alike, hence the type name)."""
    return hash(tuple((read, score, type(score).__name__) for read, score in score_table.items()))


def _pack_table(score_table):
    global _packed
    digest = _table_digest(score_table)
    if _packed is not None and _packed[0] is score_table and _packed[1] == digest:
        return _packed[2]
    reads = list(score_table.keys())
    values = list(score_table.values())
    for v in values:
        if not isinstance(v, (int, float, np.integer, np.floating)) or (isinstance(v, (int, np.integer)) and abs(int(v)) >= 1 << 53):
            raise ValueError("scores must be floats or integers below 2^53 (they are added as IEEE doubles on the device)")
    chars, off = _pack_reads(reads)
    packed = (chars, off, np.asarray(values, dtype=np.float64),
              np.fromiter((isinstance(v, (float, np.floating)) for v in values), dtype=np.uint8, count=len(values)))
    _packed = (score_table, digest, packed)
    return packed


def support_read_scores(contigs, score_table):
    """findSupportReadScore of every contig of a list, one device call.  Same values AND types as the reference:
    a contig that holds no float-scored read gets the int the reference's ``0 + ints`` gives."""
    contigs = list(contigs)
    rchars, roff, scores, is_float = _pack_table(score_table)
    cchars, coff = _pack_reads(contigs)
    out, float_hits = _graph().support_read_scores(rchars, roff, scores, is_float, cchars, coff)
    return [float(s) if f else int(s) for s, f in zip(out.tolist(), float_hits.tolist())]


def findSupportReadScore(contig, score_table):
    """IV_sortOutputs.py:10-15."""
    return support_read_scores([contig], score_table)[0]


def sort_contigs(contigs, score_table):
    """IV_sortOutputs.py:55: contigs by support score, descending, stable.  Returns (contigs, scores)."""
    scores = support_read_scores(contigs, score_table)
    order = sorted(range(len(contigs)), key=lambda i: scores[i], reverse=True)
    return [contigs[i] for i in order], [scores[i] for i in order]
