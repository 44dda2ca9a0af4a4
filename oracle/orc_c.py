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
