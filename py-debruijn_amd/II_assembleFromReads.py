#!/usr/bin/env python3
"""Drop-in for the reference's ``II_assembleFromReads.py`` on the MI355X path.

Same CLI (``-froot``), same ``{froot}/setting.json`` keys (score_cut, k_lowerlimit,
k_upperlimit, threshold, source -- II_assembleFromReads.py:32-38), same stdout lines and
the same append-mode FASTA output ``{froot}/{froot}.fasta`` with records
``>SEQUENCE_{i}_{k}mer`` (II_assembleFromReads.py:65-69).  Input is the FASTA surface the
reference keeps commented out at II_assembleFromReads.py:53:
``sequences = read_reads(f'{froot}/input_reads.fasta')`` (the live TSV path needs the
proprietary PSM tables and is outside the hot path).

Graph construction, pruning, tip removal, pull-out reads, the contig walk and the contig
scores all run on the GPU through ``debruijn`` (this directory) -> ``include/dbg.h``.
"""
import argparse
import json

import debruijn as db
from debruijn import read_reads, read_reads_device


def getScore(edge_count_table, contig, k):
    """II_assembleFromReads.py:14-18: the counts of all (k+1)-mers of a contig, added up.  Host version of the score the
    walk kernels attach to every contig (``ContigList.scores`` / ``get_score_device``); kept for callers that import it."""
    return sum(edge_count_table[contig[i:i + k + 1]] for i in range(len(contig) - k))


def get_args():
    parser = argparse.ArgumentParser()
    parser.add_argument('-froot', type=str)
    return parser.parse_args()


def assemble(sequences, k_lowerlimit, k_upperlimit, threshold, out_path=None):
    """II_assembleFromReads.py:56-75.  Returns the final, score-sorted contig list."""
    for k in range(k_lowerlimit, k_upperlimit + 1):
        final = not (k <= k_upperlimit - 1)
        g, pull_out_read, branch_kmer, already_pull_out, edge_count_table = db.construct_graph(
            sequences, k, threshold=threshold, final=final)
        contigs = db.output_contigs(g, branch_kmer, already_pull_out)
        if isinstance(contigs, db.LazyContigs):
            # more text than fits the host: the index is sorted (device scores == getScore), no contig text moves
            contigs.sort(reverse=True)
            sequences = contigs
            max_len = max(contigs.lengths, default=0)
        else:
            scores = db.get_score_device(contigs)  # == getScore(edge_count_table, x, k) for every contig
            order = sorted(range(len(contigs)), key=lambda i: scores[i], reverse=True)  # stable, like list.sort
            sequences = [contigs[i] for i in order]
            max_len = max((len(x) for x in sequences), default=0)
        if k == k_upperlimit:
            if out_path is not None:
                with open(out_path, mode='a+') as out_file:  # append mode, as in the reference
                    if hasattr(contigs, "sorted_fasta"):
                        out_file.write(contigs.sorted_fasta())  # the same records, sorted and formatted on the device
                    else:  # record by record: a LazyContigs fetches each text from the device as it is written
                        for i in range(len(sequences)):
                            out_file.write('>SEQUENCE_{}_{}mer\n{}\n'.format(i, k, sequences[i]))
            break
        print('max length: ', max_len)
        print('number of output for k={}: '.format(k), len(sequences))
        if k <= k_upperlimit - 1:
            sequences.extend(pull_out_read)
            print('number of pull out read: ', len(pull_out_read))
    return sequences


if __name__ == '__main__':
    args = get_args()
    froot = args.froot
    with open(f'{froot}/setting.json') as f:
        setting = json.load(f)
    k_lowerlimit = setting['k_lowerlimit']
    k_upperlimit = setting['k_upperlimit']
    threshold = setting['threshold']
    # II_assembleFromReads.py:53 with the FASTA parsed on the GPU (read_reads semantics, no Python string per read)
    sequences = read_reads_device(f'{froot}/input_reads.fasta')
    print(len(sequences))
    assemble(sequences, k_lowerlimit, k_upperlimit, threshold, out_path=f'{froot}/{froot}.fasta')
