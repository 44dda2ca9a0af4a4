"""One rank of a sharded build (launched by test_multi_gpu.py; test infrastructure).

mode "fake": the device library is replaced by NumpyShardGraph below, a tiny numpy model of the
four shard_* steps, so that multi_gpu.py's exchange logic runs on CPU under gloo.
mode "gpu": the real _dbg.Graph on cuda:0 (both ranks share the one GPU of the test box; gloo).
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import multi_gpu  # noqa: E402
import synth  # noqa: E402

M64 = (1 << 64) - 1


def mix(x):
    x &= M64
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & M64
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & M64
    x ^= x >> 31
    return x


def as_i64(vals):
    return torch.from_numpy(np.array(vals, dtype=np.uint64).view(np.int64).copy())


class NumpyShardGraph:
    """Model of dbg_shard_* with one record per k-mer occurrence (format is opaque to multi_gpu.py)."""

    def __init__(self, reads, k, stamp64=False):
        self.reads, self.k = reads, k
        self.mask = (1 << (2 * k)) - 1
        self.stamp64 = stamp64  # this rank hands out 64-bit rank-local stamps (a rank that holds 2 GiB of reads or more)

    def sizes(self):
        return {"n_bytes": sum(len(r) for r in self.reads)}

    def owner(self, key, n):
        bits = n.bit_length() - 1
        return (mix(key) >> (64 - bits)) if bits else 0

    def shard_extract(self, k, n):
        recs, off = [], 0
        for r in self.reads:
            codes = [(ord(c) >> 1) & 3 for c in r]
            if len(r) > k:
                for p in range(len(r) - k + 1):
                    key = 0
                    for c in codes[p:p + k]:
                        key = (key << 2) | c
                    succ = codes[p + k] if p + k < len(r) else 4
                    recs.append((self.owner(key, n), key, succ, ((off + p) << 1) | (p != 0)))
            off += len(r)
        recs.sort(key=lambda t: t[0])
        counts = [sum(1 for t in recs if t[0] == d) for d in range(n)]
        return counts, (as_i64([t[1] for t in recs]), as_i64([t[2] for t in recs]),
                        torch.tensor([t[3] for t in recs], dtype=torch.int64 if self.stamp64 else torch.int32))

    def shard_build(self, k, n, me, w0, w1, st, recv_counts, bases):
        keys = w0.numpy().view(np.uint64)
        succ = w1.numpy().view(np.uint64)
        stl = st.numpy().view(np.uint64 if st.dtype == torch.int64 else np.uint32)  # one width for all senders
        table, seg = {}, 0
        for r, c in enumerate(recv_counts):
            for i in range(seg, seg + c):
                ent = table.setdefault(int(keys[i]), [[0, 0, 0, 0], None])
                if int(succ[i]) < 4:
                    ent[0][int(succ[i])] += 1
                stamp = (bases[r] << 1) + int(stl[i])
                ent[1] = stamp if ent[1] is None else min(ent[1], stamp)
            seg += c
        self.node_keys = sorted(table)
        self.ids = {key: i for i, key in enumerate(self.node_keys)}
        self.counts = np.array([table[key][0] for key in self.node_keys], dtype=np.uint32).reshape(-1, 4)
        self.stamps = np.array([table[key][1] for key in self.node_keys], dtype=np.uint64)
        self.succ = np.full((len(self.node_keys), 4), 0xFFFFFFFF, dtype=np.uint32)
        self.me, self.n = me, n
        groups = [[] for _ in range(n)]
        for i, key in enumerate(self.node_keys):
            for code in range(4):
                if self.counts[i, code]:
                    sk = ((key << 2) | code) & self.mask
                    d = self.owner(sk, n)
                    if d == me:
                        self.succ[i, code] = (me << 29) | self.ids[sk]
                    else:
                        groups[d].append((sk, i * 4 + code))
        self.q_slots = [slot for gquery in groups for _, slot in gquery]
        self.q_owner = [d for d, gquery in enumerate(groups) for _ in gquery]
        q_counts = [len(gq) for gq in groups]
        q_starts = [sum(q_counts[:d]) for d in range(n)]
        return q_starts, q_counts, as_i64([sk for gq in groups for sk, _ in gq])

    def shard_answer(self, keys):
        return torch.tensor([self.ids[int(x)] for x in keys.numpy().view(np.uint64)], dtype=torch.int32)

    def shard_apply(self, answers):
        flat = self.succ.reshape(-1)
        for slot, d, a in zip(self.q_slots, self.q_owner, answers.tolist()):
            flat[slot] = (d << 29) | a

    def export(self):
        return np.array(self.node_keys, dtype=np.uint64), self.stamps, self.counts, self.succ

    # ---- gather protocol (multi_gpu.gather_graph)
    def sizes_device(self):
        return "cpu"

    def node_tensors(self):
        return {"keys": as_i64(self.node_keys), "stamps": torch.from_numpy(self.stamps.view(np.int64).copy()),
                "counts": torch.from_numpy(self.counts.astype(np.int32).reshape(-1).copy()),
                "succ": torch.from_numpy(self.succ.view(np.int32).reshape(-1).copy())}

    def reads_tensors(self):
        blob = np.frombuffer("".join(self.reads).encode(), dtype=np.uint8).copy()
        off = np.zeros(len(self.reads) + 1, dtype=np.int64)
        np.cumsum([len(r) for r in self.reads], out=off[1:])
        return torch.from_numpy(blob), torch.from_numpy(off)


class NumpyMultipassGraph(NumpyShardGraph):
    """Model of the ranks x passes protocol (dbg_shard_build_multipass, dbg_part_*): records grouped by the 512 level-1
    groups (top 9 bits of the key hash), part p of rank r = virtual shard r * n_passes + p."""

    def group(self, key):
        return mix(key) >> (64 - 9)

    def shard_extract(self, k, n):
        counts, (w0, w1, st) = super().shard_extract(k, n)
        keys = [int(x) for x in w0.numpy().view(np.uint64)]
        order = sorted(range(len(keys)), key=lambda i: self.group(keys[i]))  # stable: owner order is kept (owner = group prefix)
        self.l1 = [0] * 512
        for key in keys:
            self.l1[self.group(key)] += 1
        idx = torch.tensor(order, dtype=torch.int64)
        return counts, (w0[idx], w1[idx], st[idx])

    def shard_extract_part(self, k, n, part, n_parts):
        """Model of dbg_shard_extract_part: the records whose first base lies in slice ``part`` of the rank's position space."""
        _, (w0, w1, st) = self.shard_extract(k, n)
        total = max(1, self.sizes()["n_bytes"])
        stl = st.numpy().view(np.uint64 if st.dtype == torch.int64 else np.uint32)
        keep = [i for i in range(w0.numel()) if min(n_parts - 1, (int(stl[i]) >> 1) * n_parts // total) == part]
        keys = [int(x) for x in w0.numpy().view(np.uint64)[keep]] if keep else []
        self.l1 = [0] * 512
        counts = [0] * n
        for key in keys:
            self.l1[self.group(key)] += 1
            counts[self.owner(key, n)] += 1
        idx = torch.tensor(keep, dtype=torch.int64)
        return counts, (w0[idx], w1[idx], st[idx])

    def shard_record_layout(self):
        return 1, 8 if self.stamp64 else 4

    def shard_bucket_counts(self):
        return list(self.l1)

    def shard_build_multipass(self, k, n, me, n_passes, w0, w1, st, recv_counts, bases, sender_buckets):
        assert [sum(row) for row in sender_buckets] == list(recv_counts)
        super().shard_build(k, n, me, w0, w1, st, recv_counts, bases)  # the whole shard's table; the parts cut it by hash
        nv = n * n_passes
        bits = nv.bit_length() - 1
        virt = lambda key: (mix(key) >> (64 - bits)) if bits else 0
        self.nv, self.P, self.v_first = nv, n_passes, me * n_passes
        self.parts = []
        for p in range(n_passes):
            sel = [i for i, key in enumerate(self.node_keys) if virt(key) == self.v_first + p]
            keys = [self.node_keys[i] for i in sel]
            self.parts.append({"keys": keys, "ids": {key: j for j, key in enumerate(keys)},
                               "stamps": self.stamps[sel], "counts": self.counts[sel],
                               "succ_part": np.full((len(sel), 4), 255, dtype=np.uint8),
                               "succ_id": np.full((len(sel), 4), 0xFFFFFFFF, dtype=np.uint32), "queries": None})
        for p, d in enumerate(self.parts):
            groups = [[] for _ in range(nv)]
            for j, key in enumerate(d["keys"]):
                for code in range(4):
                    if d["counts"][j, code]:
                        sk = ((key << 2) | code) & self.mask
                        v = virt(sk)
                        if self.v_first <= v < self.v_first + n_passes:  # a part of this rank: resolved here
                            d["succ_part"][j, code] = v
                            d["succ_id"][j, code] = self.parts[v - self.v_first]["ids"][sk]
                        else:
                            groups[v].append((sk, j * 4 + code))
            d["queries"] = groups

    def part_queries(self, p):
        groups = self.parts[p]["queries"]
        counts = [len(gq) for gq in groups]
        starts = [sum(counts[:v]) for v in range(self.nv)]
        return starts, counts, as_i64([sk for gq in groups for sk, _ in gq])

    def part_answer(self, q, keys):
        ids = self.parts[q]["ids"]
        return torch.tensor([ids[int(x)] for x in keys.numpy().view(np.uint64)], dtype=torch.int32)

    def part_apply(self, p, owner, answers):
        d = self.parts[p]
        slots = [slot for _, slot in d["queries"][owner]]
        assert len(slots) == answers.numel()
        for slot, a in zip(slots, answers.tolist()):
            d["succ_part"].reshape(-1)[slot] = owner
            d["succ_id"].reshape(-1)[slot] = a
        d["queries"][owner] = []

    def multipass_finish(self):
        assert all(not gq for d in self.parts for gq in d["queries"]), "a successor owned by another rank stayed open"


class NumpyMergedGraph:
    """Model of dbg_import_graph: concatenated shard arrays, successor ids rewritten to global positions."""

    def set_reads_tensors(self, bases, offsets):
        self.bases, self.offsets = bases.numpy(), offsets.numpy()

    def import_graph(self, k, shard_nodes, keys, stamps, counts, succ, keys_hi=None):
        base = np.concatenate([[0], np.cumsum(shard_nodes)]).astype(np.uint64)
        s = succ.numpy().view(np.uint32).astype(np.uint64)
        none = s == 0xFFFFFFFF
        glob = base[(s >> np.uint64(29)).astype(np.int64) % len(base)] + (s & np.uint64((1 << 29) - 1))
        self.succ = np.where(none, np.uint64(0xFFFFFFFF), glob).astype(np.uint32).reshape(-1, 4)
        self.keys, self.stamps = keys.numpy().view(np.uint64), stamps.numpy().view(np.uint64)
        self.counts = counts.numpy().view(np.uint32).reshape(-1, 4)


def main():
    mode, out_dir, k, n_reads, read_len = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("SHARD_MAX_MSG"):  # force the multi-round path of alltoallv at test sizes
        multi_gpu.MAX_MESSAGE_BYTES = int(os.environ["SHARD_MAX_MSG"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = n_reads // world
    reads = synth.reads_ascii(77, max(4 * read_len, n_reads * read_len // 20), per, read_len, 0.01, first_read=rank * per)
    if mode == "fake_mp":  # ranks x passes over gloo: the parts of this rank, one npz entry per array and part
        n_passes = int(os.environ.get("SHARD_PASSES", "2"))
        g = NumpyMultipassGraph([row.tobytes().decode() for row in reads], k, stamp64=(rank == 0))
        multi_gpu.sharded_build_multipass(g, k, dist, n_passes, chunks=int(os.environ.get("SHARD_CHUNKS", "1")))
        out = {}
        for p, d in enumerate(g.parts):
            out[f"keys{p}"] = np.array(d["keys"], dtype=np.uint64)
            out[f"stamps{p}"], out[f"counts{p}"] = d["stamps"], d["counts"].reshape(-1, 4)
            out[f"succ_part{p}"], out[f"succ_id{p}"] = d["succ_part"], d["succ_id"]
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "gpu_parts":  # ranks x passes on cuda:0 over gloo, then the traversal in parts: every rank writes what it ends with
        import _dbg
        import part_traversal
        n_passes = int(os.environ.get("SHARD_PASSES", "2"))
        g = _dbg.Graph(device=0)
        g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64))
        multi_gpu.sharded_build_multipass(g, k, dist, n_passes, chunks=int(os.environ.get("SHARD_CHUNKS", "1")))
        t, flags, br, pu = part_traversal.construct_graph(g, k, 2, dist)
        ctg = part_traversal.output_contigs(t)
        texts = ctg.texts(range(len(ctg)))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), branch=np.array(br), pulled=np.array(pu), read_flags=flags,
                 contigs=np.array(texts), scores=ctg.scores, lengths=ctg.lengths)
        g.close()
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "fake":
        g = NumpyShardGraph([row.tobytes().decode() for row in reads], k)
        multi_gpu.sharded_build(g, k, dist)
        keys, stamps, counts, succ = g.export()
    else:
        import _dbg
        g = _dbg.Graph(device=0)
        g.set_reads(reads.reshape(-1), np.arange(0, reads.size + 1, read_len, dtype=np.uint64))
        multi_gpu.sharded_build(g, k, dist)
        keys, stamps, counts, _ = g.export_nodes()
        keys_hi = g.export_keys_hi()
        succ = g.export_succ()
        rp, col, cnt = g.export_csr()
        assert np.array_equal(col, succ[counts != 0]) and np.array_equal(cnt, counts[counts != 0])
        if world > 1:  # a shard's successor ids point into other ranks' tables: traversal must refuse, not follow them
            try:
                g.prune(2)
                raise AssertionError("prune on a shard must fail")
            except _dbg.DbgError as e:
                assert "gather" in str(e)
    if mode == "fake":
        keys_hi = np.zeros_like(keys)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), keys=keys, keys_hi=keys_hi, stamps=stamps, counts=counts, succ=succ)
    # ---- traversal after the sharded build: gather to rank 0, then the single-GPU path
    if mode == "fake":
        merged = multi_gpu.gather_graph(g, k, dist, dst=0, make_graph=NumpyMergedGraph)
        if rank == 0:
            np.savez(os.path.join(out_dir, "merged.npz"), keys=merged.keys, stamps=merged.stamps, counts=merged.counts,
                     succ=merged.succ, bases=merged.bases, offsets=merged.offsets)
    else:
        merged = multi_gpu.gather_graph(g, k, dist, dst=0)
        if rank == 0:
            def rest_of_path(h):
                h.refine_edge_order()
                h.prune(2)
                h.remove_tips()
                h.mark_pull_reads()
                h.walk(False)
                kk, st, cn, fl = h.export_nodes()
                o = np.argsort(st, kind="stable")
                mc, fs = h.export_orders()
                ranks = h.export_pull_ranks()
                off, chars, score, stamp, seq = h.export_contigs()
                co = np.lexsort((seq, stamp))
                text = chars.tobytes()
                return {"keys": kk[o], "keys_hi": h.export_keys_hi()[o], "stamps": st[o], "counts": cn[o], "flags": fl[o],
                        "order": mc[o], "fsorder": fs[o],
                        "pull_ranks": ranks[o], "pull_reads": h.export_pull_reads(),
                        "contigs": [text[int(off[i]):int(off[i + 1])] for i in co], "scores": score[co],
                        "sizes": {key: h.sizes()[key] for key in ("n_nodes", "n_edges", "n_branch", "n_pulled", "n_pull_reads",
                                                                  "n_starts", "n_contigs")}}
            got = rest_of_path(merged)
            allb, allo = merged.copy_reads()
            single = _dbg.Graph(device=0)      # the same reads through the single-GPU build
            single.set_reads(allb, allo)
            single.build(k)
            want = rest_of_path(single)
            assert got["sizes"] == want["sizes"], (got["sizes"], want["sizes"])
            for key in ("keys", "keys_hi", "stamps", "counts", "flags", "order", "fsorder", "pull_ranks", "pull_reads", "scores"):
                assert np.array_equal(got[key], want[key]), key
            assert got["contigs"] == want["contigs"] and len(want["contigs"]) > 0
            assert allb.size == n_reads * read_len
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
