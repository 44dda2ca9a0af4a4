"""Drop-in for the reference's ``debruijn.py`` backed by the gfx950 HIP library.

Same module surface as the reference (``read_reads``, ``construct_graph``,
``output_contigs``, ``Node``; used by II_assembleFromReads.py:11-12,58,61,63), same
argument meaning, return shapes and stdout lines; the work happens on the MI355X
through the C ABI of ``include/dbg.h`` (binding: ``_dbg.py``).  There is no CPU
path: without the built library and a GPU every call raises.

Differences a caller can observe, all documented in DESIGN.md:
  * limits: 1 <= k <= 63 (reads of upper-case A/C/G/T only: two 64-bit words per k-mer above 31; any other
    alphabet -- peptides, lower case, N, ... --: at most 32 distinct single-byte characters, packed at
    5 bits up to k = 11 and keyed by reference into the reads above); ValueError otherwise, never a
    silent drop or split;
  * ``output_contigs`` needs the objects returned by this module's ``construct_graph``;
  * graphs of ``LAZY_MIN_NODES`` (2e6, env ``DBG_LAZY_MIN_NODES``) or more nodes come back as read-only
    ``Mapping`` views over the exported arrays (label lookup, ``len``, ordered iteration; equal to the
    reference's dicts when materialised) instead of Python dicts.
Everything else -- values AND orders (dict insertion order, ``Counter.most_common`` tie order,
the append order of ``already_pull_out``, contig order) -- equals the reference.
"""
from __future__ import annotations

import os
from collections.abc import ItemsView, Mapping, Sequence

import numpy as np

import _dbg

__all__ = ["Node", "read_reads", "read_reads_device", "DeviceReads", "construct_graph", "output_contigs",
           "get_score_device", "get_kmers", "get_graph_from_kmers"]



class Node:
    """debruijn.py:8-14."""

    __slots__ = ("label", "indegree", "outdegree")

    def __init__(self, lab, indegree=0, outdegree=0):
        self.label = lab
        self.indegree = indegree
        self.outdegree = outdegree


def read_reads(fname):
    """debruijn.py:22-32: every line that does not start with '>' is one read (rstrip'ed).

    Linear time (the reference's ``reads = reads + [...]`` is quadratic); same result.
    """
    with open(fname, "r") as fh:
        return [line.rstrip() for line in fh.readlines() if line[0] != ">"]


class DeviceReads:
    """``read_reads`` on the GPU: a read-only sequence of the reads of a FASTA file.

    The file is parsed on the device (``dbg_set_reads_fasta``: newline scan, header lines dropped,
    ``rstrip``) and stays there; ``construct_graph`` accepts the object in place of the list and
    builds from the resident reads, so no Python string is created per read.  Items are
    materialised lazily (one device-to-host copy of the packed buffer on first access).
    The object owns one device handle: a later ``construct_graph`` on the same object replaces the
    graph of an earlier one (``output_contigs`` on the earlier result then raises instead of walking the new graph).
    """

    def __init__(self, fname):
        self._graph = _dbg.Graph()
        self._graph.set_reads_fasta(fname)
        self._n = self._graph.sizes()["n_reads"]
        self._host = None

    def _pull(self):
        if self._host is None:
            bases, offsets = self._graph.copy_reads()
            self._host = (bases.tobytes().decode("latin-1"), offsets)  # one byte, one character, as everywhere else
        return self._host

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        text, off = self._pull()
        return text[int(off[i]):int(off[i + 1])]

    def __iter__(self):
        text, off = self._pull()
        for i in range(self._n):
            yield text[int(off[i]):int(off[i + 1])]

    def take(self, indices):
        """``[self[i] for i in indices]`` without bringing every read to the host: the selected reads are gathered on
        the device (pull_out_read of construct_graph is 3 % of the reads at the BASELINE size)."""
        idx = np.asarray(indices, dtype=np.int64)
        idx = np.where(idx < 0, idx + self._n, idx)
        if idx.size and (int(idx.min()) < 0 or int(idx.max()) >= self._n):
            raise IndexError("read index out of range")
        if self._host is not None or idx.size == 0 or idx.size > self._n // 4:
            return [self[int(i)] for i in idx]
        chars, off = self._graph.take_reads(idx)  # dbg_take_reads: gathered on the device, one copy of the selection
        text = chars.tobytes().decode("latin-1")
        return [text[int(a):int(b)] for a, b in zip(off[:-1], off[1:])]


def read_reads_device(fname):
    """``read_reads`` (debruijn.py:22-32) without leaving the GPU; see DeviceReads."""
    return DeviceReads(fname)


class _Vertices(dict):
    """``vertices`` dict that also carries the device handle for output_contigs."""
    _graph = None
    _state = None


# Above this many nodes construct_graph returns read-only Mapping views over the exported arrays instead of
# Python dicts (dicts are infeasible at 10^8 nodes); equal to the dicts when materialised.
LAZY_MIN_NODES = int(os.environ.get("DBG_LAZY_MIN_NODES", "2000000"))


class _NodeStore:
    """Host arrays of one graph (in the library's table order) + the dict-order permutation + label <-> index
    conversion.  Indices of this class's methods are positions in dict (first-occurrence) order; ``order[i]`` is the
    row of node i in the arrays, so nothing of size n_nodes is gathered or sorted on the host up front."""

    def __init__(self, k, alphabet, bits, n, loader):
        self.k, self.alphabet, self.bits = k, alphabet, bits
        self.chars = alphabet.decode("latin-1")
        self.code_of = {ch: i for i, ch in enumerate(self.chars)}  # alphabet[code] = character
        self.n = int(n)
        # The arrays leave the device when something first asks for them: the reference's driver
        # (II_assembleFromReads.py:56-75) never reads vertices / edges / edge_count_table, and their export is most of
        # construct_graph's time at scale (37 bytes per node + the dict-order sort).
        self._loader = loader
        self._arrays = None
        self._sorted = None
        self._inverse = None

    def _get(self, name):
        if self._arrays is None:
            self._arrays = self._loader()
            self._loader = None
        return self._arrays[name]

    order = property(lambda self: self._get("order"))
    keys = property(lambda self: self._get("keys"))
    keys_hi = property(lambda self: self._get("keys_hi"))
    counts = property(lambda self: self._get("counts"))
    rank_mc = property(lambda self: self._get("rank_mc"))
    rank_fs = property(lambda self: self._get("rank_fs"))
    flags = property(lambda self: self._get("flags"))
    keep = property(lambda self: self._get("keep"))

    def rows(self, idx):
        return self.order[np.asarray(idx, dtype=np.int64)]

    def row_labels(self, rows):
        hi = None if self.keys_hi is None else self.keys_hi[rows]
        return _dbg.decode_keys(self.keys[rows], self.k, self.alphabet, self.bits, hi)

    def labels(self, idx):
        return self.row_labels(self.rows(idx))

    def indeg(self, i):
        return int(self.flags[self.order[i]] & _dbg.F_INDEG)

    def outdeg(self, i):
        return int(np.count_nonzero(self.counts[self.order[i]]))

    def pulled(self, i):
        return bool(self.flags[self.order[i]] & _dbg.F_PULLED)

    def alive(self):
        """Dict-order indices of the nodes that were not pulled out."""
        return np.nonzero((self.flags & _dbg.F_PULLED)[self.order] == 0)[0]

    def encode(self, lab):
        """(hi, lo) words of a k-character label, or None if it is not a k-mer over the alphabet."""
        if not isinstance(lab, str) or len(lab) != self.k:
            return None
        v = 0
        for ch in lab:
            c = self.code_of.get(ch)
            if c is None:
                return None
            v = (v << self.bits) | c
        return v >> 64, v & 0xFFFFFFFFFFFFFFFF

    def find(self, lab):
        """Index (dict order) of a label, or -1."""
        e = self.encode(lab)
        if e is None:
            return -1
        if self._sorted is None:  # one sort, on the first lookup
            if self.keys_hi is None:
                perm = np.argsort(self.keys, kind="stable")
            else:
                perm = np.lexsort((self.keys, self.keys_hi))
            self._sorted = (perm, self.keys[perm], None if self.keys_hi is None else self.keys_hi[perm])
            self._inverse = np.empty(self.n, dtype=np.int64)
            self._inverse[self.order] = np.arange(self.n, dtype=np.int64)
        perm, lo_s, hi_s = self._sorted
        hi, lo = e
        if hi_s is None:
            if hi:
                return -1
            a = int(np.searchsorted(lo_s, np.uint64(lo), side="left"))
            return int(self._inverse[perm[a]]) if a < self.n and int(lo_s[a]) == lo else -1
        a = int(np.searchsorted(hi_s, np.uint64(hi), side="left"))
        b = int(np.searchsorted(hi_s, np.uint64(hi), side="right"))
        c = a + int(np.searchsorted(lo_s[a:b], np.uint64(lo), side="left"))
        return int(self._inverse[perm[c]]) if c < b and int(lo_s[c]) == lo else -1

    def successors(self, i, pruned):
        """Successor labels of node i: Counter.most_common order; only the kept ones when `pruned`."""
        r = self.order[i]
        lab = self.row_labels([r])[0]
        c = self.counts[r]
        nd = int(np.count_nonzero(c))
        ranked = [int(code) for code in np.atleast_1d(self.rank_mc[r]) if code != 0xFF and c[code]][:nd]
        kp = int(self.keep[r])
        return [lab[1:] + self.chars[code] for code in ranked if not pruned or (kp >> code) & 1]

    def edge_names(self, i):
        r = self.order[i]
        lab = self.row_labels([r])[0]
        c = self.counts[r]
        return [(lab + self.chars[code], int(c[code])) for code in np.atleast_1d(self.rank_fs[r]) if code != 0xFF and c[code]]

    def edge_count(self, i, code):
        return int(self.counts[self.order[i], code]) if code < self.counts.shape[1] else 0


class _LazyVertices(Mapping):
    """``vertices``: label -> Node, dict order; nothing is materialised until it is asked for."""

    def __init__(self, store):
        self._s = store
        self._graph = None
        self._state = None

    def __len__(self):
        return self._s.n

    def __iter__(self):
        for lo in range(0, self._s.n, 1 << 16):
            yield from self._s.labels(np.arange(lo, min(lo + (1 << 16), self._s.n)))

    def __contains__(self, lab):
        return self._s.find(lab) >= 0

    def __getitem__(self, lab):
        i = self._s.find(lab)
        if i < 0:
            raise KeyError(lab)
        return Node(lab, self._s.indeg(i), self._s.outdeg(i))


class _LazyEdges(Mapping):
    """``edges``: label -> surviving successor labels, for every node that was not pulled out."""

    def __init__(self, store):
        self._s = store
        self._alive_idx = None

    @property
    def _alive(self):
        if self._alive_idx is None:
            self._alive_idx = self._s.alive()
        return self._alive_idx

    def __len__(self):
        return int(self._alive.size)

    def __iter__(self):
        for lo in range(0, self._alive.size, 1 << 16):
            yield from self._s.labels(self._alive[lo:lo + (1 << 16)])

    def __contains__(self, lab):
        i = self._s.find(lab)
        return i >= 0 and not self._s.pulled(i)

    def __getitem__(self, lab):
        i = self._s.find(lab)
        if i < 0 or self._s.pulled(i):
            raise KeyError(lab)
        return self._s.successors(i, True)


class _LazyEdgeCounts(Mapping):
    """``edge_count_table``: (k+1)-mer -> multiplicity, in the reference's insertion order."""

    def __init__(self, store):
        self._s = store
        self._n = None

    def __len__(self):
        if self._n is None:
            self._n = int(np.count_nonzero(self._s.counts))
        return self._n

    def __iter__(self):
        for i in range(self._s.n):
            for name, _ in self._s.edge_names(i):
                yield name

    def items(self):
        return _EdgeItems(self)

    def __contains__(self, name):
        try:
            self[name]
            return True
        except KeyError:
            return False

    def __getitem__(self, name):
        if not isinstance(name, str) or len(name) != self._s.k + 1:
            raise KeyError(name)
        i = self._s.find(name[:-1])
        code = self._s.code_of.get(name[-1])
        n = self._s.edge_count(i, code) if i >= 0 and code is not None else 0
        if not n:
            raise KeyError(name)
        return n


class _EdgeItems(ItemsView):
    def __iter__(self):
        s = self._mapping._s
        for i in range(s.n):
            yield from s.edge_names(i)


class _Tracked(list):
    """list that remembers which construct_graph call produced it."""
    _token = None


class ContigList(list):
    """list[str] of contigs plus the device-computed getScore of each (``.scores``).

    ``sorted_fasta()`` returns the text the reference's driver writes for this list -- contigs sorted by score,
    descending and stable, as ``>SEQUENCE_{i}_{k}mer`` records (II_assembleFromReads.py:64-69) -- sorted and formatted
    on the device (dbg_export_sorted_fasta); valid until the graph handle builds or walks again."""
    scores = None
    _graph = None
    _generation = None
    _walk = None

    def sorted_fasta(self):
        if self._graph is None or self._generation != self._graph.generation or self._walk != self._graph.walks:
            raise ValueError("the device contigs of this list are gone (the handle built another graph or walked again)")
        return self._graph.export_sorted_fasta()[0].decode("latin-1")


MAX_CONTIG_CHARS = 0  # dbg_walk's max_chars; 0 = the library default (1 GiB of contig text kept on the device)


class LazyContigs(Sequence):
    """Contigs of a walk whose text was larger than MAX_CONTIG_CHARS: same order, ``.scores`` and ``.lengths`` as
    ContigList, every text fetched from the device when it is indexed (dbg_export_contig_text).

    ``sort`` and ``extend`` exist so that the reference's own driver (II_assembleFromReads.py:64,74:
    ``sequences.sort(key=lambda x: getScore(edge_count_table, x, k), reverse=True)`` and
    ``sequences.extend(pull_out_read)``) runs unchanged on this object.  Both only permute / append: no text moves.
    ``sort`` orders by the device-computed getScore of every contig -- the one key the pipeline sorts by; evaluating an
    arbitrary ``key`` would mean fetching every text (10^12 characters at the BASELINE size).  A ``key`` that is not
    that score is detected on a sample of the shortest contigs and refused."""

    def __init__(self, graph, order, off, score):
        self._graph = graph
        self._order = np.asarray(order)
        self._off = off
        self._score = np.asarray(score)
        self._tail = []   # plain strings appended by extend()
        self.scores = self._score[self._order].tolist()
        self.lengths = (off[1:] - off[:-1])[self._order].tolist()

    def __len__(self):
        return len(self._order) + len(self._tail)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if i >= len(self._order):
            return self._tail[i - len(self._order)]
        c = int(self._order[i])
        return self._graph.export_contig_text(c, int(self._off[c + 1] - self._off[c])).decode("latin-1")

    def sort(self, key=None, reverse=False):
        """In-place, stable (``list.sort`` semantics) by the device getScore; see the class docstring."""
        if self._tail:
            raise NotImplementedError("sort after extend: the appended strings carry no device score")
        n = len(self._order)
        if key is not None and n:
            lens = (self._off[1:] - self._off[:-1])[self._order]
            for i in np.argsort(lens, kind="stable")[:4].tolist():  # the shortest ones are cheap to fetch
                if key(self[i]) != self.scores[i]:
                    raise NotImplementedError("LazyContigs.sort orders by the contig score (getScore); this key is a "
                                              "different function and would need every contig's text")
        sc = self._score[self._order].astype(np.int64)
        # stable in both directions: equal scores keep their current relative order, as list.sort(reverse=True) does
        perm = np.argsort(-sc if reverse else sc, kind="stable")
        self._order = self._order[perm]
        self.scores = [self.scores[i] for i in perm.tolist()]
        self.lengths = [self.lengths[i] for i in perm.tolist()]

    def extend(self, more):
        self._tail.extend(more)


def _pack_reads(reads):
    try:
        blob = "".join(reads).encode("latin-1")  # one byte per character
    except UnicodeEncodeError as e:
        raise ValueError("reads must be made of single-byte characters for the device path") from e
    arr = np.frombuffer(blob, dtype=np.uint8)
    lens = np.fromiter((len(r) for r in reads), dtype=np.uint64, count=len(reads))
    offsets = np.zeros(len(reads) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    return arr, offsets


def construct_graph(reads, k, threshold=3, final=False):
    """debruijn.py:206-285 on the GPU.

    Returns ((vertices, edges), pull_out_read, branch_kmer, already_pull_out, edge_count_table).
    """
    if not isinstance(k, (int, np.integer)) or not (1 <= int(k) <= 63):
        raise ValueError("the device path supports 1 <= k <= 63")
    k = int(k)
    if isinstance(reads, DeviceReads):
        g = reads._graph  # reads are resident (alphabet is checked by the kernels: AlphabetError is a ValueError)
    else:
        bases, offsets = _pack_reads(reads)
        g = _dbg.Graph()
        g.set_reads(bases, offsets)
    g.build(k)
    g.refine_edge_order()  # Counter order of the successors (first-seen ties), debruijn.py:159-165, :215-216
    sz = g.sizes()
    print('number of {}mer: '.format(k), sz["n_nodes"])  # debruijn.py:224

    if threshold == 0:
        # debruijn.py:163 divides by threshold as soon as a vertex has two distinct successors
        _, _, counts, _ = g.export_nodes(keys=False, stamps=False, flags=False)
        if ((counts != 0).sum(axis=1) > 1).any():
            raise ZeroDivisionError("division by zero")
        threshold = 1
    g.prune(threshold)
    n_branch = g.sizes()["n_branch"]
    print('branch number: ', n_branch)  # debruijn.py:236
    g.remove_tips()
    if not final:
        g.mark_pull_reads()

    alphabet, bits = g.alphabet()                 # code -> character
    chars = alphabet.decode("latin-1")
    n_nodes = sz["n_nodes"]
    two_words = bits == 2 and bits * k > 64

    def load_arrays():
        keys, stamps, counts, flags = g.export_nodes()
        rank_mc, rank_fs = g.export_orders()      # (n, D): successor codes by rank
        return {"keys": keys, "stamps": stamps, "counts": counts, "flags": flags, "rank_mc": rank_mc, "rank_fs": rank_fs,
                "order": g.export_dict_order().astype(np.int64),  # dict order == first-occurrence order (sorted on the device)
                "keys_hi": g.export_keys_hi() if two_words else None, "keep": g.export_keepmask()}

    byref = bits == 5 and bits * (k + 1) > 64  # generic alphabet, k >= 12: a node's k-mer is the text at its first occurrence
    lazy = n_nodes >= LAZY_MIN_NODES and not byref
    if lazy:  # the arrays stay on the device until a view is asked for something; then in table order, through `order`
        generation = g.generation

        def load_later():
            if g.generation != generation:
                raise RuntimeError("these views belong to a graph that a later build on the same reads replaced: read them "
                                   "(or dict() them) before the next construct_graph")
            return load_arrays()

        store = _NodeStore(k, alphabet, bits, n_nodes, load_later)
        vertices, edges, ect = _LazyVertices(store), _LazyEdges(store), _LazyEdgeCounts(store)
    else:
        a = load_arrays()
        keys, stamps, counts, flags, rank_mc, rank_fs = a["keys"], a["stamps"], a["counts"], a["flags"], a["rank_mc"], a["rank_fs"]
        order, keys_hi, keep = a["order"], a["keys_hi"], a["keep"]
        n_ranks = rank_mc.shape[1]
        if byref:
            text = reads._pull()[0] if isinstance(reads, DeviceReads) else bases.tobytes().decode("latin-1")
            labels = [text[p:p + k] for p in (stamps[order] >> np.uint64(1)).tolist()]
        else:
            labels = _dbg.decode_keys(keys[order], k, alphabet, bits, None if keys_hi is None else keys_hi[order])
        vertices, edges, ect = _Vertices(), {}, {}
        inverse = np.empty(len(order), dtype=np.int64)
        inverse[order] = np.arange(len(order))
        row_labels = lambda rows: [labels[i] for i in inverse[rows]]
        counts_o = counts[order]
        rank_mc, rank_fs = rank_mc[order], rank_fs[order]
        flags_o = flags[order]
        outdeg = (counts_o != 0).sum(axis=1)
        indeg = flags_o & _dbg.F_INDEG
        pulled_o = (flags_o & _dbg.F_PULLED) != 0
        keep_o = keep[order]
    if not lazy:
        # plain Python lists: indexing numpy scalars per node costs more than everything else in this loop
        outdeg_l, indeg_l, pulled_l, keep_l = outdeg.tolist(), indeg.tolist(), pulled_o.tolist(), keep_o.tolist()
        code1_l, cnt1_l = counts_o.argmax(axis=1).tolist(), counts_o.max(axis=1).tolist()
        for i, lab in enumerate(labels):
            nd = outdeg_l[i]
            vertices[lab] = Node(lab, indeg_l[i], nd)
            if nd == 1:    # the common case: one successor, no ranking to consult
                code = code1_l[i]
                ch = chars[code]
                ect[lab + ch] = cnt1_l[i]
                if not pulled_l[i]:
                    edges[lab] = [lab[1:] + ch] if (keep_l[i] >> code) & 1 else []
                continue
            if nd == 0:
                if not pulled_l[i]:
                    edges[lab] = []
                continue
            c = counts_o[i]
            tail = lab[1:]
            # a rank holds a code only where the node has that many successors (DNA ranks all four codes)
            ranked = [int(code) for code in rank_mc[i, :n_ranks] if code != 0xFF and c[code]][:nd]  # Counter.most_common order
            for code in rank_fs[i, :n_ranks]:                                                        # Counter key order
                if code != 0xFF and c[code]:
                    ect[lab + chars[code]] = int(c[code])
            if not pulled_l[i]:
                kp = keep_l[i]
                edges[lab] = [tail + chars[code] for code in ranked if (kp >> code) & 1]

    # the two label lists: the (few) marked rows, selected, put into the reference's order and spelled on the device side
    def marked_labels(flag):
        rows, mk, mhi = g.export_marked(flag, keys=not byref)
        if byref:
            return row_labels(rows.astype(np.int64))
        return _dbg.decode_keys(mk, k, alphabet, bits, mhi if two_words else None)

    already_pull_out = _Tracked(marked_labels(_dbg.F_PULLED))

    if final:  # debruijn.py:281-283
        pull_out_read = []
        branch_kmer = _Tracked()
    else:
        rf = g.export_pull_reads()
        pulled_idx = np.nonzero(rf)[0]
        pull_out_read = reads.take(pulled_idx) if isinstance(reads, DeviceReads) else [reads[i] for i in pulled_idx]
        branch_kmer = _Tracked(marked_labels(_dbg.F_BRANCH))

    token = object()
    vertices._graph = g
    vertices._state = {"token": token, "final": bool(final), "k": k, "n_branch": int(n_branch), "generation": g.generation}
    branch_kmer._token = token
    already_pull_out._token = token
    return (vertices, edges), pull_out_read, branch_kmer, already_pull_out, ect


def output_contigs(g, branch_kmer, already_pull_out):
    """debruijn.py:326-347 (DFS of :288-316) on the GPU.

    ``g``, ``branch_kmer`` and ``already_pull_out`` must come from one construct_graph call
    of this module (the graph lives on the device); ``branch_kmer == []`` selects the
    reference's final-mode walk.
    """
    V = g[0]
    graph = getattr(V, "_graph", None)
    state = getattr(V, "_state", None)
    if graph is None or state is None:
        raise TypeError("output_contigs needs the (vertices, edges) returned by this module's construct_graph")
    if state.get("generation") != graph.generation:
        raise ValueError("the device graph of this construct_graph result was replaced by a later build on the same handle "
                         "(construct_graph on the same DeviceReads): call output_contigs before building again")
    tok = state["token"]
    if getattr(already_pull_out, "_token", None) is not tok or (
            getattr(branch_kmer, "_token", None) is not tok and len(branch_kmer) != 0):
        raise ValueError("branch_kmer / already_pull_out must be the lists returned with this graph "
                         "(the device walk uses the graph state they describe)")
    final_mode = len(branch_kmer) == 0  # identical to the chain walk when the graph has no branch node
    sz = graph.sizes()
    print('Number of kmers have no income edges: ', sz["n_starts"])  # debruijn.py:336
    graph.walk(final_mode, MAX_CONTIG_CHARS)
    if not graph.sizes()["contigs_materialised"]:  # index only: the texts stay on the device until asked for
        off, score, stamp, seq = graph.export_contig_index()
        return LazyContigs(graph, np.lexsort((seq, stamp)), off, score)
    off, chars, score, stamp, seq = graph.export_contigs()
    order = np.lexsort((seq, stamp))  # starts in dict order, emission order inside a start
    text = chars.tobytes().decode("latin-1")
    out = ContigList(text[int(off[i]):int(off[i + 1])] for i in order)
    out.scores = [int(score[i]) for i in order]
    out._graph, out._generation, out._walk = graph, graph.generation, graph.walks
    return out


def get_score_device(contigs):
    """getScore (II_assembleFromReads.py:14-18) of each contig as computed by the walk kernel."""
    return list(contigs.scores)


# ------------------------------------------------------------------------------------------------------------------
# The two helpers of the reference module that its pipeline never calls (debruijn.py:210 mentions get_kmers in a
# comment only).  Kept so that the module surface is complete; k-mer enumeration runs on the device, the glue
# step and the overlap adjacency are host bookkeeping over short lists.
# ------------------------------------------------------------------------------------------------------------------
def _glue_short_sequences(sequences, k):
    """debruijn.py:39-56: sequences shorter than k leave the list; each is then glued onto every remaining sequence
    that overlaps it by exactly three characters (at either end).  Mutates ``sequences`` like the reference, and walks
    the short list the way a Python ``for`` does while elements are removed from it (the element after a removed
    one is skipped)."""
    short = [s for s in sequences if len(s) < k]
    for s in short:
        sequences.remove(s)
    i = 0
    while i < len(short):
        piece = short[i]
        glued = False
        for j, seq in enumerate(sequences):
            if seq[len(seq) - 3:] == piece[:3]:
                sequences[j] = seq + piece[3:]
                glued = True
            if piece[len(piece) - 3:] == seq[:3]:      # tested against the sequence as it was before this step
                sequences[j] = piece + seq[3:]
                glued = True
        if glued:
            short.remove(piece)
        i += 1


def get_kmers(sequences, k):
    """debruijn.py:35-75: distinct k-mers of the sequences (after the glue step above, which mutates the argument), in
    first-occurrence order.  Sequences longer than k go through the device build (its node table IS that list);
    sequences of exactly k characters hold one k-mer each and are merged in by position on the host."""
    _glue_short_sequences(sequences, k)
    first = {}  # k-mer -> position of its first occurrence in the concatenation of the sequences
    if any(len(s) > k for s in sequences):
        bases, offsets = _pack_reads(sequences)
        g = _dbg.Graph()
        try:
            g.set_reads(bases, offsets)
            g.build(k)
            keys, stamps, _, _ = g.export_nodes(counts=False, flags=False)
            alphabet, bits = g.alphabet()
            pos = (stamps >> np.uint64(1)).tolist()
            if bits == 5 and bits * (k + 1) > 64:      # keyed by reference: the text at the stamp is the k-mer
                text = bases.tobytes().decode("latin-1")
                labels = [text[p:p + k] for p in pos]
            else:
                hi = g.export_keys_hi() if bits == 2 and bits * k > 64 else None
                labels = _dbg.decode_keys(keys, k, alphabet, bits, hi)
            first = dict(zip(labels, pos))
        finally:
            g.close()
    off = 0
    for s in sequences:
        if len(s) == k and first.get(s, off + 1) > off:
            first[s] = off
        off += len(s)
    return sorted(first, key=first.get)


def get_graph_from_kmers(kmers, k):
    """debruijn.py:78-95: (k-1)-overlap adjacency over a list of k-mers with true in/out degrees -- same dict and list
    orders as the reference's all-pairs scan (quadratic there; prefix / suffix indexes here)."""
    vertices, edges = {}, {}
    seq_no = {}                     # key -> its position in `edges` (insertion order)
    by_prefix, by_suffix = {}, {}   # (k-1)-character prefix / suffix -> keys in insertion order
    for kmer in kmers:
        if kmer not in seq_no:
            seq_no[kmer] = len(seq_no)
            by_prefix.setdefault(kmer[:k - 1], []).append(kmer)
            by_suffix.setdefault(kmer[1:], []).append(kmer)
        node = vertices[kmer] = Node(kmer)   # a repeated k-mer starts over, as in the reference
        edges[kmer] = []
        follows = by_prefix.get(kmer[1:], ())        # kmer -> other
        precedes = by_suffix.get(kmer[:k - 1], ())   # other -> kmer
        # the reference visits every key once, in insertion order, and tests "kmer -> key" before "key -> kmer"
        a = b = 0
        while a < len(follows) or b < len(precedes):
            if b >= len(precedes) or (a < len(follows) and seq_no[follows[a]] <= seq_no[precedes[b]]):
                other = follows[a]
                a += 1
                edges[kmer].append(other)
                node.outdegree += 1
                vertices[other].indegree += 1
            else:
                other = precedes[b]
                b += 1
                edges[other].append(kmer)
                vertices[other].outdegree += 1
                node.indegree += 1
    return vertices, edges
