"""ctypes loader for the oracle's C restatement (oracle/dbg_oracle.c) -- test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liborc.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        l = C.CDLL(_LIB)
        l.orc_build.restype = C.c_void_p
        l.orc_build.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        l.orc_free.argtypes = [C.c_void_p]
        for n in ("orc_n_nodes", "orc_n_kmer_instances", "orc_n_edge_instances"):
            getattr(l, n).restype = C.c_uint64
            getattr(l, n).argtypes = [C.c_void_p]
        l.orc_export2.restype = C.c_int
        l.orc_export2.argtypes = [C.c_void_p] * 5
        l.orc_build_mt.restype = C.c_int
        l.orc_build_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p]
        l.orc_build_mt_partitioned.restype = C.c_int
        l.orc_build_mt_partitioned.argtypes = l.orc_build_mt.argtypes
        _lib = l
    return _lib


def build(bases, offsets, k, export=True):
    """Returns dict(keys, keys_hi, stamps, counts, n_nodes, n_kmer_instances, n_edge_instances); dict order.

    A k-mer is the 2k-bit number keys_hi * 2**64 + keys (keys_hi is zero for k <= 32)."""
    l = lib()
    b = np.ascontiguousarray(np.frombuffer(bases, dtype=np.uint8) if not isinstance(bases, np.ndarray) else bases)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    h = l.orc_build(b.ctypes.data, o.ctypes.data, o.size - 1, k)
    if not h:
        raise ValueError("orc_build failed (k outside 1..63, non-ACGT byte, or out of memory)")
    try:
        n = l.orc_n_nodes(h)
        out = {"n_nodes": n, "n_kmer_instances": l.orc_n_kmer_instances(h),
               "n_edge_instances": l.orc_n_edge_instances(h)}
        if export:
            keys = np.empty(n, dtype=np.uint64)
            keys_hi = np.empty(n, dtype=np.uint64)
            stamps = np.empty(n, dtype=np.uint64)
            counts = np.empty((n, 4), dtype=np.uint32)
            assert l.orc_export2(h, keys.ctypes.data, keys_hi.ctypes.data, stamps.ctypes.data, counts.ctypes.data) == 0
            out.update(keys=keys, keys_hi=keys_hi, stamps=stamps, counts=counts)
        return out
    finally:
        l.orc_free(h)


M64 = (1 << 64) - 1


def _mix64(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(33); x *= np.uint64(0xff51afd7ed558ccd)
    x ^= x >> np.uint64(33); x *= np.uint64(0xc4ceb9fe1a85ec53)
    x ^= x >> np.uint64(33)
    return x


def digest(keys, stamps, counts):
    """The node digest orc_build_mt returns, from exported arrays (numpy; wraps modulo 2**64)."""
    with np.errstate(over="ignore"):
        c = counts.astype(np.uint64)
        w = c[:, 0] + np.uint64(3) * c[:, 1] + np.uint64(5) * c[:, 2] + np.uint64(7) * c[:, 3] + np.uint64(1)
        return int(_mix64(keys ^ _mix64(stamps) ^ _mix64(w)).sum(dtype=np.uint64))


def build_mt(bases, offsets, k, n_threads, partition_once=False):
    """Multi-threaded build (k <= 31): totals and the node digest only -- bench.py's cpu_baseline on all host cores.
    partition_once: orc_build_mt_partitioned (every window is rolled and hashed by ONE thread and handed to the owner of its
    hash slice; 16 bytes per k-mer instance between the phases) instead of orc_build_mt (every thread scans everything)."""
    l = lib()
    b = np.ascontiguousarray(np.frombuffer(bases, dtype=np.uint8) if not isinstance(bases, np.ndarray) else bases)
    o = np.ascontiguousarray(offsets, dtype=np.uint64)
    out = np.zeros(5, dtype=np.uint64)
    fn = l.orc_build_mt_partitioned if partition_once else l.orc_build_mt
    rc = fn(b.ctypes.data, o.ctypes.data, o.size - 1, int(k), int(n_threads), out.ctypes.data)
    if rc:
        raise ValueError(f"orc_build_mt failed ({rc})")
    return {"n_nodes": int(out[0]), "n_edges": int(out[1]), "n_kmer_instances": int(out[2]),
            "n_edge_instances": int(out[3]), "digest": int(out[4])}
