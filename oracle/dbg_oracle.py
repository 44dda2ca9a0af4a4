"""CPU oracle for the py-debruijn hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, iterative restatement of the algorithm in the
reference's ``debruijn.py`` / ``II_assembleFromReads.py``.  It exists so that the
HIP path can be checked on machines where the reference itself is absent (the
GPU box).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product (``py-debruijn_amd/``) never does.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the real reference in
the build container and writes ``tests/golden/*.json``; ``tests/test_oracle_golden.py``
checks every function below against those vectors.

Each function cites the reference lines it restates (paths are relative to the
reference checkout).  The restatement is alphabet-agnostic (reads are ``str``),
like the reference.
"""
from __future__ import annotations

from collections import OrderedDict

TIP_DEPTH = 5  # debruijn.py:246 (literal 5 passed as ``depth``)


class Node:
    """debruijn.py:8-14 -- label / indegree / outdegree."""

    __slots__ = ("label", "indegree", "outdegree")

    def __init__(self, label, indegree=0, outdegree=0):
        self.label = label
        self.indegree = indegree
        self.outdegree = outdegree

    def __repr__(self):
        return f"Node({self.label!r}, in={self.indegree}, out={self.outdegree})"


def read_reads(fname):
    """debruijn.py:22-32 -- every line not starting with '>' is one read (rstrip'ed)."""
    out = []
    with open(fname, "r") as fh:
        for line in fh.readlines():
            if line[0] != ">":
                out.append(line.rstrip())
    return out


# --------------------------------------------------------------------------- a3
def graph_from_reads(reads, k):
    """debruijn.py:98-147 in closed form.

    * vertices: distinct k-mers of reads with len > k, dict-ordered by first
      occurrence (read index, then position).
    * edges[v]: list (multiset, occurrence order) of successors.
    * outdegree: number of DISTINCT successors (debruijn.py:130-131,135).
    * indegree: 0 iff the first occurrence of the k-mer is position 0 of its read,
      else 1 (debruijn.py:134,141-142; the test at :138 can never fire because
      v2 was appended to edges[v1] two lines earlier).
    """
    vertices = OrderedDict()
    edges = OrderedDict()
    for read in reads:
        n = len(read)
        if n <= k:  # debruijn.py:126 -- ``i + k < len(read)`` never true
            continue
        for pos in range(n - k + 1):
            kmer = read[pos:pos + k]
            node = vertices.get(kmer)
            if node is None:
                node = Node(kmer, 0 if pos == 0 else 1, 0)
                vertices[kmer] = node
                edges[kmer] = []
            if pos < n - k:
                nxt = read[pos + 1:pos + k + 1]
                lst = edges[kmer]
                if nxt not in lst:
                    node.outdegree += 1
                lst.append(nxt)
    return vertices, edges


def _first_seen_counts(lst):
    """Counter(lst) with keys in first-seen order (CPython dict semantics)."""
    c = OrderedDict()
    for x in lst:
        c[x] = c.get(x, 0) + 1
    return c


# --------------------------------------------------------------------------- a4
def edge_count_table(edges):
    """debruijn.py:213-222 -- (k+1)-mer -> number of instances."""
    table = OrderedDict()
    for v, lst in edges.items():
        for s, c in _first_seen_counts(lst).items():
            name = v + s[-1]
            table[name] = table.get(name, 0) + c
    return table


# --------------------------------------------------------------------------- a5
def prune_edges(edges, threshold):
    """debruijn.py:150-166.

    Successor s of v survives iff count(v->s) >= max_count / threshold (true
    division, debruijn.py:163); the arg-max always survives (:159).  Result lists
    are de-duplicated, ordered by count descending, ties by first-seen
    (``Counter.most_common`` is a stable sort).
    """
    for v in edges:
        counts = _first_seen_counts(edges[v])
        if len(counts) == 0:
            continue
        ranked = sorted(counts.items(), key=lambda kv: -kv[1])  # stable
        if len(ranked) == 1:
            edges[v] = [ranked[0][0]]
            continue
        top = ranked[0][1]
        keep = [ranked[0][0]]
        for s, c in ranked[1:]:
            if c >= top / threshold:
                keep.append(s)
        edges[v] = keep
    return edges


# --------------------------------------------------------------------------- a7
def _tip_paths(root, edges, vertices, pulled):
    """debruijn.py:169-186 as an explicit-stack DFS.

    Returns the de-duplicated list of paths root..t (<= TIP_DEPTH nodes) whose
    last node has PRE-pruning outdegree 0.  ``pulled`` nodes are not entered.
    The ``indegree > 1`` arm (:178-182) is unreachable (indegree is 0 or 1).
    """
    out = []
    path = [root]
    # stack of iterators over successors; depth budget = TIP_DEPTH - len(path)
    if vertices[root].outdegree == 0:
        return [[root]]
    stack = [iter(edges[root])]
    while stack:
        advanced = False
        for s in stack[-1]:
            if s in pulled:
                continue
            if len(path) >= TIP_DEPTH:  # child would be visited with depth == 0
                continue
            path.append(s)
            if vertices[s].outdegree == 0:
                if path not in out:
                    out.append(list(path))
                path.pop()
                continue
            stack.append(iter(edges[s]))
            advanced = True
            break
        if not advanced:
            stack.pop()
            path.pop()
    return out


# -------------------------------------------------------------------------- a10
def construct_graph(reads, k, threshold=3, final=False, verbose=True):
    """debruijn.py:206-285.

    Returns ((vertices, edges), pull_out_read, branch_kmer, already_pull_out,
    edge_count_table) exactly as the reference does.
    """
    vertices, edges = graph_from_reads(reads, k)
    ect = edge_count_table(edges)
    if verbose:
        print('number of {}mer: '.format(k), len(vertices))  # debruijn.py:224

    edges = prune_edges(edges, threshold)

    branch_kmer = [v for v in edges if len(edges[v]) > 1]  # :230-235
    if verbose:
        print('branch number: ', len(branch_kmer))  # :236
    branch_set = set(branch_kmer)

    already_pull_out = []  # :238
    pulled = set()
    for b in branch_kmer:  # :241-254
        if b in pulled:  # dead: branch nodes are never pulled (:251)
            continue
        for path in _tip_paths(b, edges, vertices, pulled):
            for item in path:
                if item not in pulled and item not in branch_set:
                    already_pull_out.append(item)
                    pulled.add(item)
                    edges.pop(item)

    # debruijn.py:259-272 (pruningErrorContigFromHead) is a no-op: it only emits
    # when indegree > 1, which never happens.  Nothing to restate.

    pull_out_read = []  # :274-278
    if not final:
        for read in reads:
            n = len(read)
            hit = False
            for pos in range(n - k + 1):
                if read[pos:pos + k] in branch_set:
                    hit = True
                    break
            if hit:
                pull_out_read.append(read)
    else:  # :281-283
        branch_kmer = []

    return (vertices, edges), pull_out_read, branch_kmer, already_pull_out, ect


# ---------------------------------------------------------------------- a11/a12
def output_contigs(g, branch_kmer, already_pull_out, verbose=True):
    """debruijn.py:326-347 with DFS (:288-316) as an explicit-stack walk.

    starts = vertices with indegree 0 in dict order.  Per start, every simple
    path that ends (a) just before a pulled node, (b) at a branch node, or (c) at
    a node without surviving successors is emitted once; a revisit of a node on
    the current path emits nothing.
    """
    V, E = g
    branch = set(branch_kmer)
    pulled = set(already_pull_out)
    starts = [v for v in V if V[v].indegree == 0]
    if verbose:
        print('Number of kmers have no income edges: ', len(starts))  # :336
    contigs = []
    for s in starts:
        contigs.extend(_walk_from(s, E, branch, pulled))
    return contigs


def _spell(path):
    return path[0] + "".join(p[-1] for p in path[1:])


def _walk_from(start, E, branch, pulled):
    emitted = []  # list of paths (lists) -- ``vec not in output`` at :297,:305
    result = []
    if start in pulled:  # :292-295, len(vec)==1
        return result
    path = []
    on_path = set()
    stack = []

    def enter(node):
        """Returns True if the node was pushed (needs child iteration)."""
        if node in on_path:  # :289
            return False
        if node in pulled:  # :292-303 -- emit path BEFORE it
            if path and path not in emitted:
                emitted.append(list(path))
                result.append(_spell(path))
            return False
        path.append(node)
        if node in branch or len(E[node]) == 0:  # :304-313
            if path not in emitted:
                emitted.append(list(path))
                result.append(_spell(path))
            path.pop()
            return False
        on_path.add(node)
        stack.append(iter(E[node]))
        return True

    enter(start)
    while stack:
        advanced = False
        for nxt in stack[-1]:
            if enter(nxt):
                advanced = True
                break
        if not advanced:
            stack.pop()
            on_path.discard(path.pop())
    return result


# -------------------------------------------------------------------------- a13
def get_score(ect, contig, k):
    """II_assembleFromReads.py:14-18."""
    return sum(ect[contig[i:i + k + 1]] for i in range(len(contig) - k))


# -------------------------------------------------------------------------- a14
def assemble(sequences, k_lower, k_upper, threshold, verbose=False):
    """II_assembleFromReads.py:56-75 without the file I/O.

    Returns (final_contigs_sorted, per_k_trace) where per_k_trace[k] holds the
    sorted contigs and pull-out reads of every non-final k.
    """
    sequences = list(sequences)
    trace = {}
    for k in range(k_lower, k_upper + 1):
        final = not (k <= k_upper - 1)
        g, pull, branch, pulled, ect = construct_graph(sequences, k, threshold=threshold,
                                                       final=final, verbose=verbose)
        sequences = output_contigs(g, branch, pulled, verbose=verbose)
        sequences.sort(key=lambda x: get_score(ect, x, k), reverse=True)
        if k == k_upper:
            return sequences, trace
        trace[k] = {"contigs": list(sequences), "pull_out_read": list(pull)}
        sequences.extend(pull)
    return sequences, trace


# ---- the two helpers the reference's pipeline never calls (debruijn.py:35-75, :78-95) ----------------------------
def get_kmers(sequences, k):
    """debruijn.py:35-75, literally: short sequences are removed, glued onto sequences that overlap them by three
    characters, then the distinct k-mers of what is left, in first-occurrence order.  Mutates ``sequences``."""
    short = []
    for s in sequences:
        if len(s) < k:
            short.append(s)
    for s in short:
        sequences.remove(s)
    it = 0
    while it < len(short):          # a for-loop over a list that shrinks underneath it
        piece = short[it]
        it += 1
        glued = False
        for j in range(len(sequences)):
            seq = sequences[j]
            if seq[len(seq) - 3:] == piece[:3]:
                sequences[j] = seq + piece[3:]
                glued = True
            if piece[len(piece) - 3:] == seq[:3]:
                sequences[j] = piece + seq[3:]
                glued = True
        if glued:
            short.remove(piece)
    seen = {}
    for s in sequences:
        for i in range(len(s)):
            km = s[i:i + k]
            if len(km) == k:
                seen[km] = seen.get(km, 0) + 1
    return list(seen)


def get_graph_from_kmers(kmers, k):
    """debruijn.py:78-95, literally: all-pairs (k-1)-overlap scan."""
    edges, vertices = {}, {}
    for km in kmers:
        vertices[km] = Node(km)
        edges[km] = []
        for other in edges:
            if km[1:] == other[:k - 1]:
                edges[km] = edges[km] + [other]
                vertices[km].outdegree += 1
                vertices[other].indegree += 1
            if km[:k - 1] == other[1:]:
                edges[other] = edges[other] + [km]
                vertices[other].outdegree += 1
                vertices[km].indegree += 1
    return vertices, edges


def find_support_read_score(contig, score_table):
    """IV_sortOutputs.py:10-15: the scores of the reads (dict keys, in dict order) that occur in ``contig`` as a
    substring, added up from 0 in that order (the order matters for floating-point scores; an empty read occurs in
    every contig)."""
    score = 0
    for read, value in score_table.items():
        if contig.find(read) >= 0:
            score += value
    return score
