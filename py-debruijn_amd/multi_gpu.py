"""Hash-prefix sharded graph build across the GPUs of one node.

One process per GPU (``torch.distributed``; backend ``nccl`` == RCCL over xGMI).  The path has
real exchange steps, so there are collectives -- but only all-to-alls, which use every xGMI
link of a rank at once (a ring collective would be bound by one link):

  1. every rank cuts ITS reads into super-k-mer records and groups them by owner shard
     (owner = top bits of the minimizer bucket hash, so a k-mer and its 4 successor counters
     live on exactly one rank)                                        -- dbg_shard_extract
  2. all-to-all of the records (3 arrays: 2 x uint64 + uint32 stamp)  -- RCCL
  3. every rank builds the node table of its buckets; successors owned by another shard come
     back as (successor k-mer) queries grouped by owner               -- dbg_shard_build
  4. all-to-all of the queries, owners look the k-mers up             -- dbg_shard_answer
  5. all-to-all of the answers (uint32 node ids) back                 -- dbg_shard_apply

After step 5 every rank holds its shard: node ids are (owner << 29) | local id, stamps are global.
Two-word k-mers (k > 31) go through the same five calls: a record travels by value (four words of bases, the meta
word, the rank-local stamp: dbg_wsk.h), owner = top bits of the minimizer-bucket hash as for k <= 31, and a
successor query is a (lo, hi) pair of words.
Traversal (prune / tips / pull-out reads / contig walk) crosses ranks; ``gather_graph`` moves the shards
and the reads to one rank, whose handle then behaves like a single-GPU build (gather first).
The graph object may be an ``_dbg.Graph`` or anything with the same four shard_* methods
(the CPU test uses a numpy stand-in), and the process group may be gloo (tensors are staged
through the host) or nccl.
"""
from __future__ import annotations

import torch


def _is_gloo(dist):
    return dist.get_backend() == "gloo"


def exchange_counts(dist, counts, device):
    """counts[d] = items this rank sends to d  ->  items this rank receives from every rank."""
    w = dist.get_world_size()
    send = torch.tensor(counts, dtype=torch.int64, device="cpu" if _is_gloo(dist) else device)
    recv = torch.empty(w, dtype=torch.int64, device=send.device)
    dist.all_to_all_single(recv, send)
    return [int(x) for x in recv.tolist()]


# Largest (source, destination) message of one all_to_all_single call; larger transfers go in rounds.
# Evidence (tools/nccl_large_msg.py, one rank over nccl = RCCL 2.26.6, profiles/r02_nccl_large_msg.log): a message a
# rank sends to ITSELF arrives exactly up to 1 GiB; from 1.5 GiB on only its first half is written (1.5 -> 0.75,
# 2 -> 1, 2.5 -> 1.25, 3 -> 1.5, 4 -> 2 GiB; the rest of the destination keeps its old contents), identically for
# 1-byte and 8-byte elements and with or without explicit split sizes -- so the loss is a byte-count matter inside
# RCCL's send-to-self path, not torch's split-size arithmetic.  Consequences here: (1) no message of a call exceeds
# 1 GiB, the largest size seen exact -- for the self message that is the measured limit, for messages to other ranks
# (whether the xGMI path shares the defect cannot be tested on one GPU) it is the same conservative bound; larger
# transfers go in rounds; (2) with one rank the exchange is a plain device copy; (3) every message is checked against a
# digest of its sender (ExchangeCheck), so a damaged one raises instead of building a wrong graph.
MAX_MESSAGE_BYTES = 1 << 30


def _alltoallv_once(dist, tensor, send_counts, recv_counts):
    n_out = sum(recv_counts)
    send_counts, recv_counts = list(send_counts), list(recv_counts)
    if _is_gloo(dist):
        out = torch.empty(n_out, dtype=tensor.dtype)
        dist.all_to_all_single(out, tensor.cpu().contiguous(), recv_counts, send_counts)
        return out.to(tensor.device)
    if len(send_counts) == 1 and dist.get_backend() == "nccl":
        return tensor.contiguous().clone()  # one rank: the exchange is a device copy, RCCL's send-to-self is not involved
    # every message is at most MAX_MESSAGE_BYTES here (alltoallv cuts larger ones into rounds), the size range in which
    # the send-to-self path was seen exact too
    out = torch.empty(n_out, dtype=tensor.dtype, device=tensor.device)
    dist.all_to_all_single(out, tensor.contiguous(), recv_counts, send_counts)
    return out


def alltoallv(dist, tensor, send_counts, recv_counts, biggest=None):
    """Variable all-to-all of a 1-D tensor laid out contiguously in destination order
    (all_to_all_single with split sizes; staged through the host for gloo).

    A large (source, destination) message is not safe with every backend (observed over nccl: a 7 GB
    self-copy arrived truncated, a 2 GiB one damaged), so pairs above MAX_MESSAGE_BYTES are moved in rounds of that
    size.  Every rank must run the same number of rounds: ``biggest`` = the largest message (elements) between ANY pair
    of ranks, when the caller knows it (the ranks of a sharded build know every pair's count from their one
    all-gather); None: agreed on with an all-reduce and a host sync here.
    """
    limit = max(1, MAX_MESSAGE_BYTES // tensor.element_size())
    if biggest is None:
        t = torch.tensor([max(list(send_counts) + list(recv_counts) + [0])], dtype=torch.int64,
                         device="cpu" if _is_gloo(dist) else tensor.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        biggest = int(t.item())
    rounds = max(1, -(-int(biggest) // limit))
    if rounds == 1:
        return _alltoallv_once(dist, tensor, send_counts, recv_counts)
    w = len(send_counts)
    s_off = [sum(send_counts[:d]) for d in range(w)]
    r_off = [sum(recv_counts[:d]) for d in range(w)]
    out = torch.empty(sum(recv_counts), dtype=tensor.dtype, device=tensor.device)
    for r in range(rounds):
        sc = [max(0, min(limit, c - r * limit)) for c in send_counts]
        rc = [max(0, min(limit, c - r * limit)) for c in recv_counts]
        part = torch.cat([tensor[s_off[d] + r * limit: s_off[d] + r * limit + sc[d]] for d in range(w)])
        got = _alltoallv_once(dist, part, sc, rc)
        pos = 0
        for d in range(w):
            out[r_off[d] + r * limit: r_off[d] + r * limit + rc[d]] = got[pos:pos + rc[d]]
            pos += rc[d]
    return out


def stamp_bases(dist, n_bytes, device):
    """Byte offset of every rank's reads in the rank-major concatenation of all reads."""
    w = dist.get_world_size()
    mine = torch.tensor([n_bytes], dtype=torch.int64, device="cpu" if _is_gloo(dist) else device)
    allb = [torch.empty(1, dtype=torch.int64, device=mine.device) for _ in range(w)]
    dist.all_gather(allb, mine)
    bases, acc = [], 0
    for t in allb:
        bases.append(acc)
        acc += int(t.item())
    return bases


def message_digests(t, counts):
    """Wrapping 64-bit sum of the elements of every message of a 1-D integer tensor laid out as consecutive messages
    of counts[d] elements -> int64 tensor [len(counts)] on t's device (no host sync)."""
    out, pos = [], 0
    for c in counts:
        out.append(t[pos:pos + c].sum(dtype=torch.int64).reshape(1))
        pos += c
    return torch.cat(out) if out else torch.zeros(0, dtype=torch.int64, device=t.device)


class ExchangeCheck:
    """Integrity check of the all-to-alls of one sharded step.  An all-to-all has no checksum of its own: a damaged or
    truncated message would come back as a wrong graph with rc 0 (seen once: messages above 2 GiB, see
    MAX_MESSAGE_BYTES).  Every sender digests each message it sends and every receiver each message that arrived
    (message_digests, on the device, no host sync); ``verify`` moves ALL of them -- every array exchanged since the last
    verify -- in ONE all-gather, after which every rank holds every pair's two digests and reaches the same verdict: one
    collective and one host sync per group of exchanges, and no rank leaves the protocol alone."""

    def __init__(self, dist):
        self.dist, self.pending, self.posted = dist, [], []

    def alltoallv(self, tensor, send_counts, recv_counts, what, biggest=None):
        out = alltoallv(self.dist, tensor, send_counts, recv_counts, biggest)
        self.pending.append((what, message_digests(tensor, send_counts), message_digests(out, recv_counts)))
        return out

    def post(self, tensor, send_counts, out, recv_counts, what, biggest):
        """alltoallv into ``out`` (a 1-D view with room for sum(recv_counts) elements) that does not wait: over nccl the
        exchange runs on the communicator's stream while the caller goes on (the next part of the records is cut meanwhile);
        ``wait`` joins all posted exchanges and digests what arrived.  ``tensor`` must stay untouched until then."""
        dist = self.dist
        send_counts, recv_counts = list(send_counts), list(recv_counts)
        work = None
        if len(send_counts) == 1:
            out.copy_(tensor[:send_counts[0]])    # one rank: a device copy
        elif not _is_gloo(dist) and biggest * tensor.element_size() <= MAX_MESSAGE_BYTES:
            work = dist.all_to_all_single(out, tensor.contiguous(), recv_counts, send_counts, async_op=True)
        else:                                     # gloo (staged through the host) or messages that go in rounds
            out.copy_(alltoallv(dist, tensor, send_counts, recv_counts, biggest))
        self.posted.append((what, work, tensor, out, recv_counts, self._digest(tensor, send_counts)))

    def wait(self):
        for what, work, _tensor, out, recv_counts, sent in self.posted:
            if work is not None:
                work.wait()
            self.pending.append((what, sent, self._digest(out, recv_counts)))
        self.posted = []

    _digest = staticmethod(message_digests)

    def verify(self):
        """Collective: every rank calls it at the same point, after the same sequence of alltoallv calls.  Raises on EVERY
        rank, naming (receiver, senders, array) of each damaged message."""
        dist = self.dist
        w = dist.get_world_size()
        if not self.pending:
            return
        names = [what for what, _, _ in self.pending]
        mine = torch.stack([torch.stack([snd, rcv]) for _, snd, rcv in self.pending])  # [arrays, 2, w]
        self.pending = []
        if w > 1:
            send = mine.cpu() if _is_gloo(dist) else mine
            got = [torch.empty_like(send) for _ in range(w)]
            dist.all_gather(got, send)
            table = torch.stack(got).cpu()                                               # [rank, arrays, 2, w]
        else:
            table = mine.cpu().unsqueeze(0)
        sent, arrived = table[:, :, 0, :], table[:, :, 1, :]     # sent[s, a, r]: s's digest of its message to r; arrived[r, a, s]
        bad = sent.permute(2, 1, 0) != arrived                   # [r, a, s]
        if bool(bad.any()):
            found = [(r, names[a], [x for x in range(w) if bool(bad[r, a, x])]) for r in range(w) for a in range(len(names))
                     if bool(bad[r, a].any())]
            text = "; ".join(f"rank {recv} received damaged '{what}' messages from ranks {senders}" for recv, what, senders in found)
            raise RuntimeError(f"sharded build (seen from rank {dist.get_rank()}): {text} "
                               f"(digest of the received bytes != digest the sender computed)")


class _NoCheck:
    def __init__(self, dist):
        self.dist, self.posted = dist, []

    def post(self, tensor, send_counts, out, recv_counts, what, biggest):
        ExchangeCheck.post(self, tensor, send_counts, out, recv_counts, what, biggest)

    def wait(self):
        for _what, work, _tensor, _out, _rc, _sent in self.posted:
            if work is not None:
                work.wait()
        self.posted = []

    @staticmethod
    def _digest(t, counts):
        return None

    def alltoallv(self, tensor, send_counts, recv_counts, what, biggest=None):
        return alltoallv(self.dist, tensor, send_counts, recv_counts, biggest)

    def verify(self):
        pass


def sharded_build(g, k, dist, check=True):
    """Runs steps 1-5 on graph handle ``g`` (reads already set on every rank).  Returns ``g``.
    check: verify every exchanged message against a digest computed by its sender (ExchangeCheck); raises on damage."""
    w, me = dist.get_world_size(), dist.get_rank()
    xc = ExchangeCheck(dist) if check else _NoCheck(dist)
    send_counts, (w0, w1, st) = g.shard_extract(k, w)
    device = w0.device
    # words per record in the first array: 1, or 4 where the records of two-word k-mers travel by value (their bases)
    words = g.shard_record_layout()[0] if hasattr(g, "shard_record_layout") else 1
    # one small all-gather carries everything the ranks need to know about each other: the bytes of reads every rank
    # holds (stamp bases), the record layout (all ranks must agree) and how its records split over the 512 level-1
    # buckets -- the receive counts are sums of those, and the owner can start its build at the second multisplit level
    presplit = hasattr(g, "shard_bucket_counts") and (k <= 31 or words == 4)
    st_bytes = st.element_size()
    meta = [g.sizes()["n_bytes"], words, st_bytes] + (g.shard_bucket_counts() if presplit else list(send_counts))
    metas = _all_gather_ints(dist, meta, device)
    if any(m_r[1] != words for m_r in metas):
        raise RuntimeError("sharded build: the ranks disagree about the record layout (different engines or input sizes)")
    # rank-local stamps are 32-bit while a rank's reads stay below 2 GiB: a receiver takes ONE width, the widest any
    # sender uses (a rank with narrow stamps zero-extends them: they are unsigned positions)
    if max(m_r[2] for m_r in metas) > st_bytes:
        st = widen_stamps(st)
    bases, acc = [], 0
    for m_r in metas:
        bases.append(acc)
        acc += m_r[0]
    sender_buckets = None
    if presplit:
        bps = 512 // w
        sender_buckets = [m_r[3 + me * bps: 3 + (me + 1) * bps] for m_r in metas]
        recv_counts = [sum(row) for row in sender_buckets]
    else:
        recv_counts = [m_r[3 + me] for m_r in metas]
    # every rank knows every pair's record count from the all-gather above: the number of rounds of the exchange is a
    # local computation (no all-reduce, no host sync between the three arrays)
    if presplit:
        big = max(sum(m_r[3 + d * bps: 3 + (d + 1) * bps]) for m_r in metas for d in range(w))
    else:
        big = max(m_r[3 + d] for m_r in metas for d in range(w))
    r_w0 = xc.alltoallv(w0, [words * c for c in send_counts], [words * c for c in recv_counts], "records w0", words * big)
    r_w1 = xc.alltoallv(w1, send_counts, recv_counts, "records w1", big)
    r_st = xc.alltoallv(st, send_counts, recv_counts, "records st", big)
    xc.verify()  # before anything is built from them
    if not _is_gloo(dist) and device.type == "cuda":
        torch.cuda.synchronize(device)  # the library works on its own stream
    if presplit:
        q_starts, q_counts, q_keys = g.shard_build(k, w, me, r_w0, r_w1, r_st, recv_counts, bases, sender_buckets)
    else:
        q_starts, q_counts, q_keys = g.shard_build(k, w, me, r_w0, r_w1, r_st, recv_counts, bases)

    # successors owned by other shards: keys out, node ids back.  The library lists the queries
    # grouped by owner with its own group in between: pack the remote groups for the wire.
    # one all-gather of the query counts: every rank learns what it is asked, whether anybody asks at all (every rank must
    # take the same branch: the exchange below is collective) and the largest message of the two exchanges
    q_table = _all_gather_ints(dist, list(q_counts), device)   # q_table[s][d]: rank s asks rank d about that many k-mers
    q_recv = [row[me] for row in q_table]
    if sum(sum(row) for row in q_table):
        qw = g.query_words() if hasattr(g, "query_words") else 1  # two-word k-mers: a query is a (lo, hi) pair
        q_big = max(max(row) for row in q_table)
        groups = [q_keys[qw * s:qw * (s + c)] for s, c in zip(q_starts, q_counts)]
        packed = torch.cat(groups) if groups else q_keys[:0]
        keys_in = xc.alltoallv(packed, [qw * c for c in q_counts], [qw * c for c in q_recv], "successor queries", qw * q_big)
        if not _is_gloo(dist) and device.type == "cuda":
            torch.cuda.synchronize(device)
        answers_out = g.shard_answer(keys_in)
        back = xc.alltoallv(answers_out, q_recv, q_counts, "successor answers", q_big)
        xc.verify()
        answers = torch.empty(q_keys.numel() // qw, dtype=torch.int32, device=device)
        off = 0
        for s, c in zip(q_starts, q_counts):
            answers[s:s + c] = back[off:off + c]
            off += c
        if not _is_gloo(dist) and device.type == "cuda":
            torch.cuda.synchronize(device)
    else:  # one rank (or the global-table engine of two-word k-mers): nothing to ask
        answers = torch.empty(0, dtype=torch.int32, device=device)
    g.shard_apply(answers)
    return g


def _exchange_records(g, k, dist, xc):
    """Steps 1-3 of a sharded build for super-k-mer records split by the 512 level-1 groups: extract, one all-gather
    of what the ranks must know about each other, all-to-all of the records.  -> (received w0, w1, st, recv_counts,
    stamp bases, sender_buckets)."""
    w, me = dist.get_world_size(), dist.get_rank()
    send_counts, (w0, w1, st) = g.shard_extract(k, w)
    device = w0.device
    words = g.shard_record_layout()[0]
    st_bytes = st.element_size()
    metas = _all_gather_ints(dist, [g.sizes()["n_bytes"], words, st_bytes] + g.shard_bucket_counts(), device)
    if any(m_r[1] != words for m_r in metas):
        raise RuntimeError("sharded build: the ranks disagree about the record layout (different engines or input sizes)")
    if max(m_r[2] for m_r in metas) > st_bytes:
        st = widen_stamps(st)
    bases, acc = [], 0
    for m_r in metas:
        bases.append(acc)
        acc += m_r[0]
    bps = 512 // w
    sender_buckets = [m_r[3 + me * bps: 3 + (me + 1) * bps] for m_r in metas]
    recv_counts = [sum(row) for row in sender_buckets]
    big = max(sum(m_r[3 + d * bps: 3 + (d + 1) * bps]) for m_r in metas for d in range(w))  # largest message of any pair
    r_w0 = xc.alltoallv(w0, [words * c for c in send_counts], [words * c for c in recv_counts], "records w0", words * big)
    r_w1 = xc.alltoallv(w1, send_counts, recv_counts, "records w1", big)
    r_st = xc.alltoallv(st, send_counts, recv_counts, "records st", big)
    xc.verify()
    if not _is_gloo(dist) and device.type == "cuda":
        torch.cuda.synchronize(device)
    return r_w0, r_w1, r_st, recv_counts, bases, sender_buckets


# receive buffers of an exchange in parts: sized from the first part x the number of parts x this factor (tests shrink it to
# walk the growth path)
PART_SLACK = 1.15


def _exchange_records_in_parts(g, k, dist, xc, chunks):
    """_exchange_records with the rank's records cut and sent in ``chunks`` parts (dbg_shard_extract_part): part c
    is on the wire -- three posted all-to-alls on the communicator's stream -- while part c + 1 is extracted and split on
    the library's stream.  The receiver sees chunks x world senders: sender c * world + r = part c of rank r, all parts of a
    rank with that rank's stamp base.  The receive buffers are sized from the first part (parts are equal slices of the
    reads) with 15 % to spare and grown by a copy if a later part turns out larger."""
    w, me = dist.get_world_size(), dist.get_rank()
    bps = 512 // w
    bufs, cap, cursor = None, 0, 0
    recv_all, rows_all, bases = [], [], None
    for c in range(chunks):
        send_counts, (w0, w1, st) = g.shard_extract_part(k, w, c, chunks)
        device = w0.device
        words = g.shard_record_layout()[0]
        st_bytes = st.element_size()
        metas = _all_gather_ints(dist, [g.sizes()["n_bytes"], words, st_bytes] + g.shard_bucket_counts(), device)
        if any(m_r[1] != words for m_r in metas):
            raise RuntimeError("sharded build: the ranks disagree about the record layout (different engines or input sizes)")
        if max(m_r[2] for m_r in metas) > st_bytes:
            st = widen_stamps(st)
        if bases is None:
            bases, acc = [], 0
            for m_r in metas:
                bases.append(acc)
                acc += m_r[0]
        rows = [m_r[3 + me * bps: 3 + (me + 1) * bps] for m_r in metas]
        recv_counts = [sum(row) for row in rows]
        big = max(sum(m_r[3 + d * bps: 3 + (d + 1) * bps]) for m_r in metas for d in range(w))
        n_in = sum(recv_counts)
        if bufs is None or cursor + n_in > cap:
            new_cap = cursor + max(n_in, int(n_in * (chunks - c) * PART_SLACK) + (4096 if PART_SLACK >= 1 else 0))  # this part fits in any case
            xc.wait()  # (growing: the posted exchanges write into the old buffers)
            new = [torch.empty(words * new_cap, dtype=torch.int64, device=device), torch.empty(new_cap, dtype=torch.int64, device=device),
                   torch.empty(new_cap, dtype=st.dtype, device=device)]
            if bufs is not None:
                for old_b, new_b, mult in zip(bufs, new, (words, 1, 1)):
                    new_b[:mult * cursor] = old_b[:mult * cursor]
            bufs, cap = new, new_cap
        xc.post(w0, [words * x for x in send_counts], bufs[0][words * cursor: words * (cursor + n_in)], [words * x for x in recv_counts],
                f"records w0, part {c}", words * big)
        xc.post(w1, send_counts, bufs[1][cursor: cursor + n_in], recv_counts, f"records w1, part {c}", big)
        xc.post(st, send_counts, bufs[2][cursor: cursor + n_in], recv_counts, f"records st, part {c}", big)
        cursor += n_in
        recv_all += recv_counts
        rows_all += rows
    xc.wait()
    xc.verify()
    if not _is_gloo(dist) and device.type == "cuda":
        torch.cuda.synchronize(device)
    return bufs[0][:words * cursor], bufs[1][:cursor], bufs[2][:cursor], recv_all, bases * chunks, rows_all


def sharded_build_multipass(g, k, dist, n_passes, check=True, chunks=1):
    """Ranks x passes (BASELINE.json configs[3]: shards that outgrow one 32-bit id space): every rank builds its shard
    as ``n_passes`` parts (dbg_shard_build_multipass); part p of rank r is virtual shard r * n_passes + p.  Successors
    owned by another rank are resolved pass by pass -- the same p on every rank at a time: keys out, part-local node ids
    back (two all-to-alls per pass).  Afterwards ``g.export_part(p)`` / ``g.part_tensors(p)`` hold the rank's parts with
    ``col_part`` = the virtual shard of every successor.
    chunks > 1: the records are cut and sent in that many parts, the exchange of one part under the extraction of the
    next (_exchange_records_in_parts); the graph is the same."""
    w, me, P = dist.get_world_size(), dist.get_rank(), int(n_passes)
    xc = ExchangeCheck(dist) if check else _NoCheck(dist)
    if chunks > 1 and hasattr(g, "shard_extract_part"):
        r_w0, r_w1, r_st, recv_counts, bases, sender_buckets = _exchange_records_in_parts(g, k, dist, xc, int(chunks))
    else:
        r_w0, r_w1, r_st, recv_counts, bases, sender_buckets = _exchange_records(g, k, dist, xc)
    device = r_w0.device
    g.shard_build_multipass(k, w, me, P, r_w0, r_w1, r_st, recv_counts, bases, sender_buckets)
    for p in range(P):
        q_starts, q_counts, q_keys = g.part_queries(p)
        qw = g.query_words() if hasattr(g, "query_words") else 1  # two-word k-mers: a query is a (lo, hi) pair
        # what I ask every (rank, part) about: one all-gather, after which every rank knows the split of every message over
        # the owner's parts and the largest message of this pass's two exchanges (no all-reduce inside them)
        q_table = _all_gather_ints(dist, list(q_counts), device)      # q_table[s][d * P + q]: rank s asks part q of rank d
        theirs = [q_table[s][me * P + q] for s in range(w) for q in range(P)]   # theirs[s * P + q]: rank s asks my part q
        send = [sum(q_counts[d * P:(d + 1) * P]) for d in range(w)]
        recv = [sum(theirs[s * P:(s + 1) * P]) for s in range(w)]
        q_big = max(sum(row[d * P:(d + 1) * P]) for row in q_table for d in range(w))
        groups = [q_keys[qw * q_starts[v]:qw * (q_starts[v] + q_counts[v])] for v in range(w * P) if q_counts[v]]
        packed = torch.cat(groups) if groups else q_keys[:0]
        keys_in = xc.alltoallv(packed, [qw * c for c in send], [qw * c for c in recv], "successor queries", qw * q_big)
        if not _is_gloo(dist) and device.type == "cuda":
            torch.cuda.synchronize(device)
        out, pos = [], 0
        for s in range(w):
            for q in range(P):
                n = theirs[s * P + q]
                if n:
                    out.append(g.part_answer(q, keys_in[qw * pos:qw * (pos + n)]))
                pos += n
        answers_out = torch.cat(out) if out else torch.empty(0, dtype=torch.int32, device=device)
        back = xc.alltoallv(answers_out, recv, send, "successor answers", q_big)
        xc.verify()
        if not _is_gloo(dist) and device.type == "cuda":
            torch.cuda.synchronize(device)
        pos = 0
        for v in range(w * P):
            n = q_counts[v]
            if n:
                g.part_apply(p, v, back[pos:pos + n].contiguous())
            pos += n
    g.multipass_finish()
    return g


def widen_stamps(st):
    """32-bit rank-local stamps (unsigned, carried in an int32 tensor) -> the same values as int64."""
    return st.to(torch.int64) & 0xFFFFFFFF


def _all_gather_ints(dist, vals, device):
    w = dist.get_world_size()
    mine = torch.tensor(vals, dtype=torch.int64, device="cpu" if _is_gloo(dist) else device)
    out = [torch.empty_like(mine) for _ in range(w)]
    dist.all_gather(out, mine)
    return [[int(x) for x in t.tolist()] for t in out]


def gather_graph(g, k, dist, dst=0, make_graph=None):
    """After ``sharded_build``: every rank sends its shard (node arrays) and its reads to rank ``dst``.

    Returns, on ``dst``, a fresh handle holding the whole graph over the rank-major concatenation of the
    reads -- successor ids rewritten to positions in the concatenated node arrays (dbg_import_graph) -- ready
    for refine_edge_order / prune / remove_tips / mark_pull_reads / walk; ``None`` on the other ranks.
    The gather is an all-to-all in which only ``dst`` receives (variable sizes, one call per array).
    """
    w, me = dist.get_world_size(), dist.get_rank()
    nodes = g.node_tensors()
    bases, offsets = g.reads_tensors()
    device = bases.device
    n, nb, nr = nodes["keys"].numel(), bases.numel(), offsets.numel() - 1
    sizes = _all_gather_ints(dist, [n, nb, nr], device)
    byte_base = [sum(s[1] for s in sizes[:r]) for r in range(w)]

    def to_dst(x, which, width=1):
        send = [x.numel() if r == dst else 0 for r in range(w)]
        recv = [sizes[r][which] * width if me == dst else 0 for r in range(w)]
        return alltoallv(dist, x, send, recv)

    keys = to_dst(nodes["keys"], 0)
    keys_hi = to_dst(nodes["keys_hi"], 0) if "keys_hi" in nodes else None  # two-word k-mers (k > 31)
    stamps = to_dst(nodes["stamps"], 0)
    counts = to_dst(nodes["counts"], 0, 4)
    succ = to_dst(nodes["succ"], 0, 4)
    all_bases = to_dst(bases, 1)
    ends = to_dst(offsets[1:] + byte_base[me], 2)  # read ends in the concatenation
    if me != dst:
        return None
    all_offsets = torch.cat([torch.zeros(1, dtype=torch.int64, device=ends.device), ends])
    if not _is_gloo(dist) and device.type == "cuda":
        torch.cuda.synchronize(device)
    if make_graph is None:
        import _dbg
        merged = _dbg.Graph(device=g.sizes_device())
    else:
        merged = make_graph()
    merged.set_reads_tensors(all_bases, all_offsets)
    if keys_hi is None:
        merged.import_graph(k, [s[0] for s in sizes], keys, stamps, counts, succ)
    else:
        merged.import_graph(k, [s[0] for s in sizes], keys, stamps, counts, succ, keys_hi)
    return merged
