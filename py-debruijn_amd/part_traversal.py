"""Traversal of a graph that lives in PARTS -- a multi-pass build on one GPU (BASELINE.json configs[3]), the ranks of a
sharded build (configs[2], [4]), or ranks x passes -- without any GPU ever holding the whole graph:

    pruningEdges + branch list          debruijn.py:150-166, :230-236     per part (dbg_part_prune)
    tip removal                         debruijn.py:169-186, :241-254     the <= 5-step neighbourhoods of the branch nodes are
                                                                          collected from all parts; every rank runs the library's
                                                                          reservation kernels on that small graph
    pull-out reads + Counter order      debruijn.py:274-278, :159-165     every rank streams ITS reads against the branch k-mers of
                                                                          the whole graph (dbg_scan_reads_for_keys)
    contig walk (non-final)             debruijn.py:288-347               chains leave a part at almost every change of minimizer:
                                                                          one short segment per entry node inside each part
                                                                          (dbg_part_segments), then the skeleton of segments --
                                                                          a few per cent of the nodes -- is ranked by pointer jumping

A node is (virtual shard v, local id); v = rank * n_passes + part, global id = v << 32 | local.  The heavy per-node work
is in the library (include/dbg.h "traversal of a graph in parts"); this module moves the small id lists and rows between
parts and ranks (``torch.distributed`` all-to-all, or nothing at all on one process) and sorts / joins them with torch
and numpy.  The result equals the reference's on the whole read set: ``branch_kmer`` (dict order), ``already_pull_out``
(append order), the pull-out reads of every rank, and the contig index (start stamp, length, score) in dict order.
threshold >= 1 (below 1 the kept successor itself depends on the Counter order).
"""
from __future__ import annotations

import math

import numpy as np
import torch

import _dbg
import multi_gpu

F_INDEG, F_KEEP_MASK, F_KEEP_SHIFT, F_BRANCH, F_PULLED, PF_MARK = 0x01, 0x1E, 1, 0x20, 0x40, 0x80
NO_NODE = 0xFFFFFFFF
TIP_REACH = 5  # debruijn.py:246: the DFS enters nodes up to 4 steps from the branch node and looks at the flags of the fifth
K_EMIT, K_PULLED, K_REMOTE, K_CYCLE, K_NEXT_PULLED = 0, 1, 2, 3, 4


def _u32(t):
    """int32 tensor holding uint32 values -> int64"""
    return t.to(torch.int64) & 0xFFFFFFFF


def rank_skeleton(e, k, keep=False):
    """The skeleton of segments -> the contig index.  ``e``: dict of equally long int64 tensors, one row per entry node:
    gid (global id), kind (K_*), next (global id of the entry a REMOTE segment hands over to), hops, score, exit (count of the
    leaving edge), stamp, start (1: indegree 0).  Entering a pulled node ends the path at the previous one (debruijn.py:
    292-302); a chain that meets a cycle inside a part (K_CYCLE) or across parts (never resolves) emits nothing (:289-290), nor
    does a start that is itself pulled.  -> (dict(stamp, length, score) numpy in dict order of the starts, skeleton or None)."""
    n = e["gid"].numel()
    if n == 0:
        z = np.empty(0, dtype=np.int64)
        sk = {"gid": z, "next": z, "go_on": z.astype(bool), "hops": z, "emit": z} if keep else None
        return {"stamp": z.astype(np.uint64), "length": z, "score": z}, sk
    o = torch.argsort(e["gid"])
    e = {c: t[o] for c, t in e.items()}
    remote = e["kind"] == K_REMOTE
    want = torch.where(remote, e["next"], e["gid"])
    j = torch.searchsorted(e["gid"], want).clamp_(max=n - 1)
    assert bool((e["gid"][j] == want).all()), "a chain continues at a node that is no entry"
    go_on = remote & (e["kind"][j] != K_PULLED)     # entering a pulled node ends the path at the previous one
    sk = {"gid": e["gid"].cpu().numpy(), "next": j.cpu().numpy(), "go_on": go_on.cpu().numpy(), "hops": e["hops"].cpu().numpy()} if keep else None
    hops = e["hops"] + go_on.to(torch.int64)
    score = e["score"] + torch.where(go_on, e["exit"], torch.zeros_like(e["exit"]))
    dead = e["kind"] == K_CYCLE
    done = ~go_on
    jump = torch.where(go_on, j, torch.arange(n, device=e["gid"].device))
    # pointer jumping over the entries that are not resolved yet (the list shrinks fast: most chains are a few segments);
    # a round reads the old values of its targets before anything is written
    act = torch.nonzero(~done).reshape(-1)
    for _ in range(max(1, math.ceil(math.log2(n + 1))) + 1):
        if act.numel() == 0:
            break
        tj = jump[act]
        h2, s2, d2, dn, j2 = hops[act] + hops[tj], score[act] + score[tj], dead[act] | dead[tj], done[tj], jump[tj]
        hops[act], score[act], dead[act], done[act], jump[act] = h2, s2, d2, dn, j2
        act = act[~dn]
    emit = (e["start"] == 1) & (e["kind"] != K_PULLED) & done & ~dead   # not done after log2(n) doublings: a cycle across parts
    em = torch.nonzero(emit).reshape(-1)
    em = em[torch.argsort(e["stamp"][em])]           # dict order of the starts (stamps are distinct), sorted on the device
    if keep:
        sk["emit"] = em.cpu().numpy()                # skeleton index of contig i's start
    return {"stamp": e["stamp"][em].cpu().numpy().astype(np.uint64), "length": (hops[em] + k).cpu().numpy(),
            "score": score[em].cpu().numpy()}, sk


class _Net:
    """How rows reach the rank that owns them: torch.distributed, or a single process (dist None)."""

    def __init__(self, dist, device):
        self.dist, self.device = dist, device
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0

    def route(self, dest_rank, *cols):
        """Rows (parallel 1-D tensors) to the rank dest_rank[i]; returns the received columns."""
        if self.world == 1:
            return cols
        order = torch.argsort(dest_rank, stable=True)
        send = torch.bincount(dest_rank, minlength=self.world).tolist()
        recv = multi_gpu.exchange_counts(self.dist, send, self.device)
        return tuple(multi_gpu.alltoallv(self.dist, c[order].contiguous(), send, recv) for c in cols)

    def gather_all(self, *cols, widths=None):
        """Every rank's rows on every rank (rank order).  widths[i]: elements per row of column i (1 unless given)."""
        if self.world == 1:
            return cols
        widths = widths or [1] * len(cols)
        n = cols[0].numel() // widths[0]
        sizes = [m[0] for m in multi_gpu._all_gather_ints(self.dist, [n], self.device)]
        return tuple(multi_gpu.alltoallv(self.dist, c.reshape(-1).repeat(self.world), [n * w] * self.world, [s * w for s in sizes])
                     for c, w in zip(cols, widths))

    def all_ints(self, vals):
        if self.world == 1:
            return [list(vals)]
        return multi_gpu._all_gather_ints(self.dist, list(vals), self.device)

    def min_u64(self, arr):
        """element-wise minimum over ranks of a numpy uint64 array (values below 2^63 or the all-ones sentinel)"""
        if self.world == 1:
            return arr
        t = torch.from_numpy(arr.view(np.int64).copy())
        t = torch.where(t < 0, torch.full_like(t, (1 << 63) - 1), t)  # the sentinel as the largest int64
        if self.dist.get_backend() == "nccl":
            t = t.to(self.device)
        parts = self.gather_all(t.reshape(-1))[0].reshape(self.world, -1)
        m = parts.min(dim=0).values.cpu().numpy().astype(np.uint64)
        m[m == np.uint64((1 << 63) - 1)] = np.iinfo(np.uint64).max
        return m.reshape(arr.shape)


class PartTraversal:
    """The traversal state of one process: its handle ``g`` with ``n_passes`` parts (part p = virtual shard rank * P + p)."""

    def __init__(self, g, k, dist=None):
        self.g, self.k = g, int(k)
        self.P = g.part_count()
        if self.P < 1:
            raise _dbg.DbgError(_dbg.DBG_E_ARG, "part_traversal: the handle holds no graph in parts (build_multipass / sharded_build_multipass first)")
        self.device = torch.device("cuda", g.sizes_device())
        self.net = _Net(dist, self.device)
        self.rank, self.world = self.net.rank, self.net.world
        self.nv = self.world * self.P
        self.n_nodes = [g.part_sizes(p)["n_nodes"] for p in range(self.P)]

    # ---- helpers
    def _v(self, p):
        return self.rank * self.P + p

    def _to_parts(self, gids):
        """global ids (int64, any rank's) -> list over my parts of int32 local-id tensors, after routing to the owners"""
        v = gids >> 32
        (mine,) = self.net.route(torch.div(v, self.P, rounding_mode="floor"), gids)
        v = (mine >> 32) - self.rank * self.P
        loc = (mine & 0xFFFFFFFF).to(torch.int32)  # wraps to the int32 carrying the uint32
        return [loc[v == p].contiguous() for p in range(self.P)]

    def _rows(self, p, ids):
        r = self.g.part_gather(p, ids)
        n = ids.numel()
        gid = (self._v(p) << 32) | _u32(ids)
        so = r["succ_owner"].to(torch.int64).reshape(n, 4)
        sl = _u32(r["succ_local"]).reshape(n, 4)
        succ_gid = torch.where(so == 0xFF, torch.full_like(sl, -1), (so << 32) | sl)
        return {"gid": gid, "keys": r["keys"], "keys_hi": r["keys_hi"], "stamps": r["stamps"],
                "counts": _u32(r["counts"]).reshape(n, 4), "succ": succ_gid, "pflags": r["pflags"].to(torch.int64)}

    @staticmethod
    def _cat(rows):
        keys = rows[0].keys()
        return {c: torch.cat([r[c] for r in rows]) for c in keys}

    # ---- a5 + a6
    def prune(self, threshold):
        """pruningEdges + branch detection on every part; returns the branch nodes of the WHOLE graph in dict order:
        dict(gid, keys, keys_hi, stamps, counts [n, 4], succ [n, 4] global ids, pflags) as numpy arrays."""
        if not threshold >= 1:
            raise ValueError("traversal in parts takes threshold >= 1")
        self.threshold = float(threshold)
        rows = []
        for p in range(self.P):
            self.g.part_prune(p, threshold)
            ids = self.g.part_select(p, F_BRANCH, F_BRANCH) if self.n_nodes[p] else torch.empty(0, dtype=torch.int32, device=self.device)
            rows.append(self._rows(p, ids))
        mine = self._cat(rows)
        cols = list(mine.keys())
        got = dict(zip(cols, self.net.gather_all(*[mine[c].reshape(-1) for c in cols], widths=[4 if c in ("counts", "succ") else 1 for c in cols])))
        n = got["gid"].numel()
        got["counts"], got["succ"] = got["counts"].reshape(n, 4), got["succ"].reshape(n, 4)
        o = torch.argsort(got["stamps"])   # dict order = ascending first-occurrence stamp (distinct, below 2^63); sorted on the device
        self.branch = {c: t[o].cpu().numpy() for c, t in got.items()}
        return self.branch

    # ---- a9 + the Counter order of the branch nodes' successors
    def pull_out_reads(self):
        """-> uint8 flags over THIS rank's reads: 1 where the read holds a branch k-mer of the whole graph (debruijn.py:274-278).
        Also fills ``branch_order``: per branch node the successor codes by (count descending, first appearance) --
        Counter.most_common order, debruijn.py:159-165 -- needed by the tip removal."""
        b = self.branch
        keys = b["keys"].astype(np.uint64)
        hi = b["keys_hi"].astype(np.uint64)
        flags, seen = self.g.scan_reads_for_keys(self.k, keys, hi if self.k > 32 else None, first_seen=True)
        sizes = self.net.all_ints([self.g.sizes()["n_bytes"]])
        base = sum(s[0] for s in sizes[:self.rank])
        if seen.size:
            present = seen != np.iinfo(np.uint64).max
            seen = np.where(present, seen + np.uint64(base), seen)
            seen = self.net.min_u64(seen)
        self.first_seen = seen
        self.read_flags = flags
        return flags

    def _order_bytes(self, counts, first_seen):
        """rank bytes (dbg_export_orders layout: code at rank r in bits 2r+1:2r) by (count desc, first appearance asc, code)."""
        n = counts.shape[0]
        order = np.zeros(n, dtype=np.uint8)
        if not n:
            return order
        c = counts.astype(np.int64)
        fs = first_seen.astype(np.float64)  # only compared among present successors; 2^64 - 1 sorts last
        idx = np.lexsort((np.tile(np.arange(4), (n, 1)), fs, -c), axis=1)  # per row: codes sorted by (-count, first seen, code)
        for r in range(4):
            order |= (idx[:, r].astype(np.uint8) << np.uint8(2 * r))
        return order

    # ---- a7
    def remove_tips(self):
        """Tip removal on the neighbourhoods of the branch nodes; marks the pulled nodes in their parts and returns
        ``already_pull_out`` as dict(gid, keys, keys_hi) in append order (numpy)."""
        g, P = self.g, self.P
        for p in range(P):
            g.part_clear(p, PF_MARK)
        # breadth-first over kept edges from the branch nodes: TIP_REACH rounds, the last round's nodes are only looked at
        frontier = [g.part_select(p, F_BRANCH, F_BRANCH) if self.n_nodes[p] else torch.empty(0, dtype=torch.int32, device=self.device)
                    for p in range(P)]
        for p in range(P):
            g.part_mark(p, frontier[p], PF_MARK)
        rows, dist_of = [], []
        for depth in range(TIP_REACH + 1):
            out_gids = []
            for p in range(P):
                r = self._rows(p, frontier[p])
                rows.append(r)
                dist_of.append(torch.full((frontier[p].numel(),), depth, dtype=torch.int64, device=self.device))
                if depth < TIP_REACH:
                    keep = (r["pflags"][:, None] >> (F_KEEP_SHIFT + torch.arange(4, device=self.device)[None, :])) & 1
                    out_gids.append(r["succ"][(keep == 1) & (r["succ"] >= 0)])
            if depth == TIP_REACH:
                break
            nxt = self._to_parts(torch.cat(out_gids) if out_gids else torch.empty(0, dtype=torch.int64, device=self.device))
            frontier = []
            for p in range(P):
                ids = torch.unique(nxt[p])
                new = g.part_mark(p, ids, PF_MARK, newly=True)
                frontier.append(ids[new == 1].contiguous())
        mine = self._cat(rows)
        mine["dist"] = torch.cat(dist_of)
        cols = list(mine.keys())
        got = dict(zip(cols, self.net.gather_all(*[mine[c].reshape(-1) for c in cols], widths=[4 if c in ("counts", "succ") else 1 for c in cols])))
        n = got["gid"].numel()
        got["counts"], got["succ"] = got["counts"].reshape(n, 4), got["succ"].reshape(n, 4)
        o = torch.argsort(got["gid"])                      # sorted by global id on the device: the join below is a binary search
        got = {c: t[o] for c, t in got.items()}
        pos_t = torch.searchsorted(got["gid"], torch.where(got["succ"] >= 0, got["succ"], torch.zeros_like(got["succ"]))).clamp_(max=max(n - 1, 0))
        pulled_gid = np.empty(0, dtype=np.int64)
        if n:
            # the mini-graph stays on the device: masked counts, successors as positions in the sorted list, rank bytes
            dev = self.device
            ar4 = torch.arange(4, device=dev)
            keep = ((got["pflags"][:, None] >> (F_KEEP_SHIFT + ar4[None, :])) & 1) == 1
            full = (got["dist"] < TIP_REACH)[:, None]
            counts = torch.where(keep & full, got["counts"], torch.zeros_like(got["counts"]))  # pruned edges and the outermost ring: no successors
            found = (got["gid"][pos_t] == got["succ"]) & (counts != 0)
            dup = (got["gid"][1:] == got["gid"][:-1]).any() if n > 1 else torch.zeros((), dtype=torch.bool, device=dev)
            ok = torch.stack([~dup, (found == (counts != 0)).all()]).cpu().tolist()   # one host sync for both checks
            assert ok[0], "a node was collected twice"
            assert ok[1], "a kept successor inside the neighbourhood was not collected"
            succ = torch.where(found, pos_t, torch.full_like(pos_t, -1))
            # Counter order: exact for the branch nodes (the only nodes whose successor order the DFS can see);
            # any permutation serves a node with at most one kept successor
            order = torch.full((n,), 0xE4, dtype=torch.uint8, device=dev)
            b = self.branch
            if b["gid"].size:
                bpos = torch.searchsorted(got["gid"], torch.from_numpy(b["gid"].astype(np.int64)).to(dev))
                order[bpos] = torch.from_numpy(self._order_bytes(counts[bpos].cpu().numpy(), self.first_seen)).to(dev)
            helper = _dbg.Graph(device=g.sizes_device())
            try:
                helper.set_reads(np.frombuffer(b"A", dtype=np.uint8), np.array([0, 1], dtype=np.uint64))  # import wants a read set
                tc = counts.to(torch.int32).reshape(-1).contiguous()     # (the uint32 counts in int32 / int64 clothes)
                tsu = succ.to(torch.int32).reshape(-1).contiguous()      # -1 = NO_NODE
                helper.import_graph(self.k, [n], got["keys"].contiguous(), got["stamps"].contiguous(), tc, tsu,
                                    got["keys_hi"].contiguous() if self.k > 31 else None)
                helper.set_orders(order.cpu().numpy())
                helper.prune(self.threshold)
                helper.remove_tips()
                _, _, _, hflags = helper.export_nodes(keys=False, stamps=False, counts=False)
                hr = helper.export_pull_ranks()
                self.tip_rounds = helper.sizes()["tip_rounds"]
            finally:
                helper.close()
            sel = np.nonzero(hflags & F_PULLED)[0]
            sel = sel[np.argsort(hr[sel], kind="stable")]
            st = torch.from_numpy(sel.astype(np.int64)).to(dev)
            pulled_gid = got["gid"][st].cpu().numpy()
            self.pulled = {"gid": pulled_gid, "keys": got["keys"][st].cpu().numpy().astype(np.uint64),
                           "keys_hi": got["keys_hi"][st].cpu().numpy().astype(np.uint64)}
        else:
            self.pulled = {"gid": pulled_gid, "keys": np.empty(0, np.uint64), "keys_hi": np.empty(0, np.uint64)}
        # every rank computed the same list: it marks its own nodes
        gt = torch.from_numpy(pulled_gid.astype(np.int64)).to(self.device)
        v = gt >> 32
        for p in range(P):
            ids = (gt[v == self._v(p)] & 0xFFFFFFFF).to(torch.int32).contiguous()
            g.part_mark(p, ids, F_PULLED)
            g.part_clear(p, PF_MARK)
        return self.pulled

    # ---- a11 + a12 + a13, non-final mode
    def walk_index(self, keep_skeleton=False):
        """output_contigs (debruijn.py:326-347) with the branch list of construct_graph: the contig index in dict order of the
        starts -- dict(stamp uint64, length int64, score int64) numpy arrays.  A start whose chain runs into a cycle emits
        nothing (debruijn.py:289-290), nor does a start that was pulled.  keep_skeleton: keep the segment skeleton on the
        host (26 bytes per entry) so that ``contig_texts`` can spell contigs afterwards."""
        g, P, dev = self.g, self.P, self.device
        # entries: the starts (indegree 0) and every node a chain of another part continues at
        sends = []
        for p in range(P):
            g.part_clear(p, PF_MARK)
            if self.n_nodes[p]:
                g.part_mark(p, g.part_select(p, F_INDEG, 0), PF_MARK)
                counts, targets = g.part_cross_targets(p)
                owner = torch.repeat_interleave(torch.arange(self.nv, device=dev), torch.tensor(counts, device=dev))
                sends.append((owner << 32) | _u32(targets))
        got = self._to_parts(torch.cat(sends) if sends else torch.empty(0, dtype=torch.int64, device=dev))
        segs = []
        for p in range(P):
            if not self.n_nodes[p]:
                continue
            g.part_mark(p, torch.unique(got[p]), PF_MARK)
            ent = g.part_select(p, PF_MARK, PF_MARK)
            s = g.part_segments(p, ent)
            r = g.part_gather(p, ent, what=("stamps", "pflags"))
            gid = (self._v(p) << 32) | _u32(ent)
            nxt = (s["next_owner"].to(torch.int64) << 32) | _u32(s["next_local"])
            segs.append({"gid": gid, "kind": s["kind"].to(torch.int64), "next": nxt, "hops": _u32(s["hops"]), "score": s["score"],
                         "exit": _u32(s["last"]), "stamp": r["stamps"], "start": ((r["pflags"] & F_INDEG) == 0).to(torch.int64)})
            g.part_clear(p, PF_MARK)
        if segs:
            mine = self._cat(segs)
        else:
            mine = {c: torch.empty(0, dtype=torch.int64, device=dev) for c in ("gid", "kind", "next", "hops", "score", "exit", "stamp", "start")}
        cols = list(mine.keys())
        e = dict(zip(cols, self.net.gather_all(*[mine[c] for c in cols])))
        index, skeleton = rank_skeleton(e, self.k, keep_skeleton)
        if keep_skeleton:
            self.skeleton = skeleton
        return index

    def contig_texts(self, which):
        """The text of the contigs ``which`` (positions in the index ``walk_index(keep_skeleton=True)`` returned) -- every rank
        calls it with the same list and gets the same strings.  A contig is its start k-mer followed by the last base of every
        node its chain appends (debruijn.py:296-299); every part spells the segments it holds (dbg_part_segment_text)."""
        sk, g, dev = self.skeleton, self.g, self.device
        which = [int(c) for c in which]
        if not which:
            return []
        seg, owner_of_seg, first = [], [], []
        for c in which:
            i = int(sk["emit"][c])
            first.append(len(seg))
            seg.append(i)
            while sk["go_on"][i]:
                i = int(sk["next"][i])
                seg.append(i)
        first.append(len(seg))
        seg = np.asarray(seg, dtype=np.int64)
        gid = sk["gid"][seg] if seg.size else np.empty(0, dtype=np.int64)
        v = gid >> 32
        plen = 1 + sk["hops"][seg] if seg.size else np.empty(0, dtype=np.int64)
        # my pieces, in segment order; then everybody's
        pos_l, bytes_l = [], []
        for p in range(self.P):
            mine = np.nonzero(v == self._v(p))[0]
            if not mine.size:
                continue
            ent = torch.from_numpy((gid[mine] & 0xFFFFFFFF).astype(np.uint32).view(np.int32).copy()).to(dev)
            off = torch.zeros(mine.size + 1, dtype=torch.int64, device=dev)
            off[1:] = torch.cumsum(torch.from_numpy(plen[mine]).to(dev), 0)
            bytes_l.append(g.part_segment_text(p, ent, off))
            pos_l.append(torch.from_numpy(mine).to(dev))
        pos = torch.cat(pos_l) if pos_l else torch.empty(0, dtype=torch.int64, device=dev)
        byts = torch.cat(bytes_l) if bytes_l else torch.empty(0, dtype=torch.uint8, device=dev)
        # the start k-mers
        starts = np.asarray([int(sk["emit"][c]) for c in which], dtype=np.int64)
        sg = sk["gid"][starts] if starts.size else np.empty(0, dtype=np.int64)
        kl, kh, kg = [], [], []
        for p in range(self.P):
            mine = np.nonzero((sg >> 32) == self._v(p))[0]
            if mine.size:
                ids = torch.from_numpy((sg[mine] & 0xFFFFFFFF).astype(np.uint32).view(np.int32).copy()).to(dev)
                r = g.part_gather(p, ids, what=("keys", "keys_hi"))
                kl.append(r["keys"]); kh.append(r["keys_hi"]); kg.append(torch.from_numpy(sg[mine]).to(dev))
        z = torch.empty(0, dtype=torch.int64, device=dev)
        kl, kh, kg = (torch.cat(x) if x else z for x in (kl, kh, kg))
        if self.world > 1:
            n_mine = pos.numel()
            sizes = [m[0] for m in self.net.all_ints([n_mine])]
            (pos,) = self.net.gather_all(pos)
            # the bytes: variable-length rows, gathered as one flat column of known per-rank sizes
            nb = [m[0] for m in self.net.all_ints([byts.numel()])]
            byts = multi_gpu.alltoallv(self.net.dist, byts.repeat(self.world), [byts.numel()] * self.world, nb)
            kl, kh, kg = self.net.gather_all(kl, kh, kg)
            del sizes
        pos, byts = pos.cpu().numpy(), byts.cpu().numpy()
        # pieces arrive grouped by rank and part, each group in ascending segment position: place them by position
        lens = plen[pos]
        src_off = np.concatenate([[0], np.cumsum(lens)])
        dst_off = np.concatenate([[0], np.cumsum(plen)])
        flat = np.empty(int(dst_off[-1]), dtype=np.uint8)
        for q, a in zip(pos.tolist(), range(pos.size)):
            flat[dst_off[q]:dst_off[q + 1]] = byts[src_off[a]:src_off[a + 1]]
        kmers = dict(zip(kg.cpu().numpy().tolist(), _dbg.decode_keys(kl.cpu().numpy().astype(np.uint64), self.k,
                                                                        keys_hi=kh.cpu().numpy().astype(np.uint64))))
        out = []
        for ci, c in enumerate(which):
            a, b = first[ci], first[ci + 1]
            body = flat[dst_off[a] + 1:dst_off[b]].tobytes().decode("ascii")   # the start segment's first character is the start's own last base
            # drop the entry character of no later segment: entering a node appends its last base
            out.append(kmers[int(sg[ci])] + body)
        return out


class PartContigs:
    """The contigs of a graph in parts, in the reference's order (dict order of the starts): ``lengths``, ``scores`` (getScore,
    II_assembleFromReads.py:14-18), ``start_stamps``; ``texts(indices)`` spells any of them (collective: every rank asks for
    the same ones).  At scale the contigs overlap massively -- 2e12 characters for 10 M reads -- so the text is on demand."""

    def __init__(self, traversal, index):
        self._t = traversal
        self.start_stamps, self.lengths, self.scores = index["stamp"], index["length"], index["score"]

    def __len__(self):
        return int(self.start_stamps.size)

    def texts(self, indices):
        return self._t.contig_texts(indices)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.texts(range(*i.indices(len(self))))
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self.texts([i])[0]


def construct_graph(g, k, threshold=3, dist=None):
    """construct_graph (debruijn.py:206-285) after a build in parts (``g.build_multipass`` or, with ``dist``,
    ``multi_gpu.sharded_build_multipass``), every rank calling it: -> (traversal, pull_out_read_flags, branch_kmer,
    already_pull_out) with the reference's lists as str (dict order / append order; the same on every rank) and the
    pull-out reads as uint8 flags over THIS rank's reads (the reference's list is the flagged reads of rank 0, 1, ... in order)."""
    t = PartTraversal(g, k, dist)
    b = t.prune(threshold)
    flags = t.pull_out_reads()
    p = t.remove_tips()
    branch_kmer = _dbg.decode_keys(b["keys"].astype(np.uint64), t.k, keys_hi=b["keys_hi"].astype(np.uint64))
    already_pull_out = _dbg.decode_keys(p["keys"], t.k, keys_hi=p["keys_hi"])
    return t, flags, branch_kmer, already_pull_out


def output_contigs(traversal):
    """output_contigs (debruijn.py:326-347) for the branch list ``construct_graph`` above found (the non-final walk)."""
    return PartContigs(traversal, traversal.walk_index(keep_skeleton=True))


def traverse(g, k, threshold, dist=None):
    """prune -> pull-out reads -> tips -> non-final walk over the graph in parts on ``g`` (every rank calls it).
    -> dict(branch, pulled, read_flags, contigs)."""
    t = PartTraversal(g, k, dist)
    branch = t.prune(threshold)
    flags = t.pull_out_reads()
    pulled = t.remove_tips()
    contigs = t.walk_index()
    return {"branch": branch, "pulled": pulled, "read_flags": flags, "contigs": contigs, "tip_rounds": getattr(t, "tip_rounds", 0)}
