#!/usr/bin/env python3
"""Drop-in for the read-support scoring of the reference's ``IV_sortOutputs.py`` on the MI355X path.

``findSupportReadScore(contig, score_table)`` keeps the reference's name, arguments and result
(IV_sortOutputs.py:10-15: the scores of the reads -- the dict's keys -- that occur in the contig as a substring, added
in dict order); ``support_read_scores(contigs, score_table)`` scores a whole list in one device call, which is what a
caller with more than a handful of contigs wants: the reference tests reads x contigs substrings on the host, the
device indexes the reads once and looks every contig position up (csrc/dbg_support.h).  No CPU path: both raise
without the library and a GPU.  The CLI below is the reference's (``-froot``, ``setting.json``, the PSM score tables
under ``source``, ``{froot}/{froot}_sorted.fasta`` in append mode); the tables are proprietary inputs of the reference's
pipeline, so only the scoring and sorting it feeds are covered by tests.
"""
import argparse
import json
import os

import numpy as np

import _dbg
from debruijn import _pack_reads, read_reads

_handle = None
_packed = {}  # id(score_table) -> (len, packed arrays): the reference calls the function once per contig with one table


def _graph():
    global _handle
    if _handle is None:
        _handle = _dbg.Graph()
    return _handle


def _pack_table(score_table):
    key = id(score_table)
    hit = _packed.get(key)
    if hit is not None and hit[0] == len(score_table):
        return hit[1]
    reads = list(score_table.keys())
    values = list(score_table.values())
    for v in values:
        if not isinstance(v, (int, float, np.integer, np.floating)) or (isinstance(v, (int, np.integer)) and abs(int(v)) >= 1 << 53):
            raise ValueError("scores must be floats or integers below 2^53 (they are added as IEEE doubles on the device)")
    chars, off = _pack_reads(reads)
    packed = (chars, off, np.asarray(values, dtype=np.float64),
              np.fromiter((isinstance(v, (float, np.floating)) for v in values), dtype=np.uint8, count=len(values)))
    _packed.clear()
    _packed[key] = (len(score_table), packed)
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


def get_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('-froot', type=str)
    return parser.parse_args()


def load_score_table(file_path, score_cut):
    """IV_sortOutputs.py:34-49: DENOVO sequence -> summed Score of the PSM rows that pass the cuts."""
    import pandas as pd
    sequences_scores = dict()
    for root, _dirs, files in os.walk(file_path):
        root = root + '/'
        for file in files:
            data = pd.read_csv(root + file, delimiter='\t')
            temp = data[data['Score'] >= score_cut]
            temp = temp[-50 < temp['PPM Difference']]
            temp = temp[temp['PPM Difference'] < 50]
            temp.reset_index(inplace=True)
            for i in range(len(temp)):
                seq = temp['DENOVO'][i]
                sequences_scores[seq] = temp['Score'][i] + sequences_scores[seq] if seq in sequences_scores else temp['Score'][i]
    return sequences_scores


def sort_contigs(contigs, score_table):
    """IV_sortOutputs.py:55: contigs by support score, descending, stable.  Returns (contigs, scores)."""
    scores = support_read_scores(contigs, score_table)
    order = sorted(range(len(contigs)), key=lambda i: scores[i], reverse=True)
    return [contigs[i] for i in order], [scores[i] for i in order]


if __name__ == '__main__':
    args = get_args()
    froot = args.froot
    with open(f'{froot}/setting.json') as f:
        setting = json.load(f)
    print(setting)
    table = load_score_table(setting['source'], setting['score_cut'])
    contigs, scores = sort_contigs(read_reads(f'{froot}/{froot}.fasta'), table)
    k = setting['k_upperlimit']
    with open(f'{froot}/{froot}_sorted.fasta', mode='a+') as out_file:  # append mode, as in the reference
        for i in range(len(contigs)):
            out_file.writelines('>SEQUENCE_{}_{}mer_{}\n{}\n'.format(i, k, round(scores[i], 2), contigs[i]))
