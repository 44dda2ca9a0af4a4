"""An in-process stand-in for ``torch.distributed`` (test infrastructure): N ranks are N threads of one process.

The GPU box allows only a few processes on its card, so an 8-rank sharded build cannot be rehearsed there with
one process per rank.  multi_gpu.py only needs ``get_rank / get_world_size / get_backend / all_to_all_single /
all_reduce / all_gather / barrier / ReduceOp.MAX`` of its ``dist`` argument; this module provides them over shared
memory of one process, so that eight ``_dbg.Graph`` handles (one per thread, all on cuda:0) run the real
``multi_gpu.sharded_build`` -- the 3-owner-bit path -- against the C ABI.  ctypes releases the GIL during library
calls and every handle has its own stream, so the ranks really interleave.
"""
import threading
import types

import torch


class _World:
    def __init__(self, n):
        self.n = n
        self.barrier = threading.Barrier(n)
        self.slots = [None] * n


class InProcDist:
    ReduceOp = types.SimpleNamespace(MAX="max", SUM="sum")

    def __init__(self, world, rank):
        self._w, self._rank = world, rank

    def get_rank(self):
        return self._rank

    def get_world_size(self):
        return self._w.n

    def get_backend(self):
        return "inproc"

    def barrier(self):
        self._w.barrier.wait()

    def _sync(self, t):
        if t.is_cuda:
            torch.cuda.synchronize(t.device)

    class _Done:
        """What an asynchronous collective returns (torch.distributed's Work): here the exchange has already happened."""

        def wait(self):
            return True

    def all_to_all_single(self, out, inp, out_splits=None, in_splits=None, async_op=False):
        w, n = self._w, self._w.n
        if in_splits is None:
            in_splits = [inp.numel() // n] * n
        if out_splits is None:
            out_splits = [out.numel() // n] * n
        self._sync(inp)
        w.slots[self._rank] = (inp, list(in_splits))
        w.barrier.wait()
        pos = 0
        for src in range(n):
            t, splits = w.slots[src]
            off = sum(splits[:self._rank])
            c = splits[self._rank]
            assert c == out_splits[src], (src, self._rank, c, out_splits[src])
            out[pos:pos + c] = t[off:off + c].to(out.device)
            pos += c
        self._sync(out)
        w.barrier.wait()
        return self._Done() if async_op else None

    def all_reduce(self, t, op=None):
        w = self._w
        self._sync(t)
        w.slots[self._rank] = t.clone()
        w.barrier.wait()
        vals = torch.stack([x.to(t.device) for x in w.slots])
        res = vals.max(dim=0).values if op == "max" else vals.sum(dim=0)
        w.barrier.wait()
        t.copy_(res)

    def all_gather(self, outs, t):
        w = self._w
        self._sync(t)
        w.slots[self._rank] = t.clone()
        w.barrier.wait()
        for o, x in zip(outs, w.slots):
            o.copy_(x.to(o.device))
        w.barrier.wait()


def run_ranks(n, fn):
    """fn(dist, rank) on n threads; returns the list of results, re-raises the first failure."""
    world = _World(n)
    results, errors = [None] * n, []

    def body(r):
        try:
            results[r] = fn(InProcDist(world, r), r)
        except BaseException as e:  # noqa: BLE001 -- reported below
            errors.append((r, e))
            world.barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        errors.sort(key=lambda x: isinstance(x[1], threading.BrokenBarrierError))
        raise errors[0][1]
    return results
