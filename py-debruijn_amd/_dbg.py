"""ctypes binding of include/dbg.h (the C ABI of the gfx950 de Bruijn hot path).

There is no CPU fallback: if ``libdbg_hip.so`` is missing or no GPU is visible the
calls fail loudly.  Build the library with ``py-debruijn_amd/csrc/build.sh`` (or
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DBG_LIB") or os.path.join(_HERE, "libdbg_hip.so")  # DBG_LIB: A/B builds (tools/)

DBG_OK, DBG_E_ARG, DBG_E_HIP, DBG_E_ALPHABET, DBG_E_CAPACITY, DBG_E_NOMEM = 0, -1, -2, -3, -4, -5
F_INDEG, F_KEEP_MASK, F_KEEP_SHIFT, F_BRANCH, F_PULLED = 0x01, 0x1E, 1, 0x20, 0x40
NO_NODE = 0xFFFFFFFF
ABI_VERSION = 5

# symbols declared in include/dbg.h; tests check that the library exports every one of them
SYMBOLS = (
    "dbg_create", "dbg_destroy", "dbg_last_error", "dbg_abi_version", "dbg_set_option", "dbg_set_reads", "dbg_set_reads_fasta",
    "dbg_set_reads_device",
    "dbg_synth_reads", "dbg_reads_checksum", "dbg_copy_reads", "dbg_reads_device", "dbg_build", "dbg_refine_edge_order", "dbg_export_orders",
    "dbg_get_alphabet", "dbg_export_keepmask", "dbg_export_dict_order",
    "dbg_prune", "dbg_remove_tips",
    "dbg_mark_pull_reads", "dbg_walk", "dbg_get_sizes", "dbg_get_stats", "dbg_export_nodes", "dbg_export_keys_hi",
    "dbg_export_succ",
    "dbg_export_csr", "dbg_export_pull_ranks", "dbg_export_pull_reads", "dbg_export_contigs",
    "dbg_export_contig_index", "dbg_export_contig_text", "dbg_device_views", "dbg_shard_extract", "dbg_shard_bucket_counts", "dbg_shard_record_layout", "dbg_shard_build", "dbg_shard_answer", "dbg_shard_apply",
    "dbg_import_graph", "dbg_device_keys_hi",
    "dbg_support_read_scores", "dbg_export_sorted_fasta", "dbg_build_multipass", "dbg_part_count", "dbg_part_sizes", "dbg_export_part", "dbg_part_device_views",
    "dbg_shard_build_multipass", "dbg_shard_build_multipass_from", "dbg_shard_extract_part", "dbg_part_queries", "dbg_part_answer", "dbg_part_apply", "dbg_multipass_finish",
    "dbg_export_marked", "dbg_part_keys_hi", "dbg_take_reads",
    "dbg_part_prune", "dbg_part_select", "dbg_part_gather", "dbg_part_mark", "dbg_part_clear", "dbg_part_cross_targets",
    "dbg_part_segments", "dbg_part_pflags", "dbg_scan_reads_for_keys", "dbg_set_orders", "dbg_part_segment_text",
)


class Sizes(C.Structure):
    _fields_ = [("k", C.c_int32), ("abi_version", C.c_int32)] + [
        (n, C.c_uint64) for n in (
            "n_reads", "n_bytes", "n_kmer_instances", "n_edge_instances", "table_capacity", "n_nodes", "n_edges",
            "n_branch", "n_pulled", "n_pull_reads", "n_starts", "n_contigs", "contig_chars", "tip_rounds",
            "contigs_materialised", "max_degree")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Stats(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "ms_startbits", "ms_table_init", "ms_count", "ms_compact", "ms_succ", "ms_csr", "ms_build_total", "ms_prune",
        "ms_tips", "ms_pull_reads", "ms_walk", "ms_h2d", "ms_extract", "ms_partition")] + [
        (n, C.c_uint64) for n in ("count_launches", "n_records", "n_buckets", "n_queries")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class DbgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"dbg error {code}: {msg}")
        self.code = code


class AlphabetError(DbgError, ValueError):
    pass


_lib = None


def load_library():
    """Loads libdbg_hip.so (once).  Raises OSError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found: build it with py-debruijn_amd/csrc/build.sh "
                      f"(there is no CPU fallback for the device path)")
    # torch (used for device tensors and torch.distributed, never for compute) bundles its own HIP runtime under the same
    # SONAME as the system one this library links to.  Whichever is loaded first serves both; torch fails to find a GPU
    # when it is the second ("No HIP GPUs are available"), so it goes first when it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    u64p, u32p, u8p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    vp = C.c_void_p
    sig = {
        "dbg_create": (C.c_int, [C.c_int, C.POINTER(H)]),
        "dbg_destroy": (None, [H]),
        "dbg_last_error": (C.c_char_p, [H]),
        "dbg_abi_version": (C.c_int, []),
        "dbg_set_option": (C.c_int, [H, C.c_char_p, C.c_int64]),
        "dbg_set_reads": (C.c_int, [H, vp, vp, C.c_uint64]),
        "dbg_set_reads_fasta": (C.c_int, [H, vp, C.c_uint64]),
        "dbg_set_reads_device": (C.c_int, [H, vp, C.c_uint64, vp, C.c_uint64]),
        "dbg_synth_reads": (C.c_int, [H, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32]),
        "dbg_reads_checksum": (C.c_int, [H, u64p]),
        "dbg_copy_reads": (C.c_int, [H, vp, vp]),
        "dbg_build": (C.c_int, [H, C.c_int, C.c_uint64]),
        "dbg_refine_edge_order": (C.c_int, [H]),
        "dbg_export_orders": (C.c_int, [H, vp, vp]),
        "dbg_get_alphabet": (C.c_int, [H, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "dbg_export_keepmask": (C.c_int, [H, vp]),
        "dbg_prune": (C.c_int, [H, C.c_double]),
        "dbg_remove_tips": (C.c_int, [H]),
        "dbg_mark_pull_reads": (C.c_int, [H]),
        "dbg_walk": (C.c_int, [H, C.c_int, C.c_uint64]),
        "dbg_get_sizes": (C.c_int, [H, C.POINTER(Sizes)]),
        "dbg_get_stats": (C.c_int, [H, C.POINTER(Stats)]),
        "dbg_export_nodes": (C.c_int, [H, vp, vp, vp, vp]),
        "dbg_export_keys_hi": (C.c_int, [H, vp]),
        "dbg_export_dict_order": (C.c_int, [H, vp]),
        "dbg_export_succ": (C.c_int, [H, vp]),
        "dbg_export_csr": (C.c_int, [H, vp, vp, vp]),
        "dbg_export_pull_ranks": (C.c_int, [H, vp]),
        "dbg_export_pull_reads": (C.c_int, [H, vp]),
        "dbg_export_contigs": (C.c_int, [H, vp, vp, vp, vp, vp]),
        "dbg_export_contig_index": (C.c_int, [H, vp, vp, vp, vp]),
        "dbg_export_contig_text": (C.c_int, [H, C.c_uint64, vp, C.c_uint64]),
        "dbg_device_views": (C.c_int, [H] + [C.POINTER(vp)] * 5),
        "dbg_shard_extract": (C.c_int, [H, C.c_int, C.c_int, u64p, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "dbg_shard_bucket_counts": (C.c_int, [H, u64p]),
        "dbg_shard_record_layout": (C.c_int, [H, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "dbg_shard_build": (C.c_int, [H, C.c_int, C.c_int, C.c_int, vp, vp, vp, u64p, u64p, u64p, u64p, C.POINTER(vp), u64p, C.c_int]),
        "dbg_shard_answer": (C.c_int, [H, vp, C.c_uint64, vp]),
        "dbg_build_multipass": (C.c_int, [H, C.c_int, C.c_int]),
        "dbg_export_sorted_fasta": (C.c_int, [H, vp, vp, C.c_uint64, u64p]),
        "dbg_support_read_scores": (C.c_int, [H, vp, vp, C.c_uint64, vp, vp, vp, vp, C.c_uint64, vp, vp]),
        "dbg_part_count": (C.c_int, [H, C.POINTER(C.c_int)]),
        "dbg_shard_build_multipass": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, u64p, u64p, u64p]),
        "dbg_shard_build_multipass_from": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, u64p, u64p, u64p]),
        "dbg_shard_extract_part": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_int, u64p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
        "dbg_part_queries": (C.c_int, [H, C.c_int, u64p, u64p, C.POINTER(vp)]),
        "dbg_part_answer": (C.c_int, [H, C.c_int, vp, C.c_uint64, vp]),
        "dbg_part_apply": (C.c_int, [H, C.c_int, C.c_int, vp]),
        "dbg_multipass_finish": (C.c_int, [H]),
        "dbg_export_marked": (C.c_int, [H, C.c_uint32, C.c_uint64, u64p, vp, vp, vp]),
        "dbg_part_sizes": (C.c_int, [H, C.c_int, u64p, u64p, u64p]),
        "dbg_export_part": (C.c_int, [H, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
        "dbg_part_device_views": (C.c_int, [H, C.c_int] + [C.POINTER(vp)] * 2 + [C.POINTER(C.c_int)] + [C.POINTER(vp)] * 5),
        "dbg_part_keys_hi": (C.c_int, [H, C.c_int, vp, C.POINTER(vp)]),
        "dbg_take_reads": (C.c_int, [H, vp, C.c_uint64, vp, vp, C.c_uint64, u64p]),
        "dbg_part_prune": (C.c_int, [H, C.c_int, C.c_double, u64p]),
        "dbg_part_select": (C.c_int, [H, C.c_int, C.c_uint32, C.c_uint32, vp, C.c_uint64, u64p]),
        "dbg_part_gather": (C.c_int, [H, C.c_int, vp, C.c_uint64, vp, vp, vp, vp, vp, vp, vp]),
        "dbg_part_mark": (C.c_int, [H, C.c_int, vp, C.c_uint64, C.c_uint32, vp]),
        "dbg_part_clear": (C.c_int, [H, C.c_int, C.c_uint32]),
        "dbg_part_cross_targets": (C.c_int, [H, C.c_int, u64p, vp, C.c_uint64]),
        "dbg_part_segments": (C.c_int, [H, C.c_int, vp, C.c_uint64, vp, vp, vp, vp, vp, vp]),
        "dbg_part_pflags": (C.c_int, [H, C.c_int, C.POINTER(vp)]),
        "dbg_part_segment_text": (C.c_int, [H, C.c_int, vp, C.c_uint64, vp, vp, C.c_uint64]),
        "dbg_scan_reads_for_keys": (C.c_int, [H, C.c_int, vp, vp, C.c_uint64, vp, vp]),
        "dbg_set_orders": (C.c_int, [H, vp]),
        "dbg_shard_apply": (C.c_int, [H, vp]),
        "dbg_import_graph": (C.c_int, [H, C.c_int, C.c_int, u64p, vp, vp, vp, vp, vp]),
        "dbg_device_keys_hi": (C.c_int, [H, C.POINTER(vp)]),
        "dbg_reads_device": (C.c_int, [H, C.POINTER(vp), u64p, C.POINTER(vp), u64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.dbg_abi_version() != ABI_VERSION:
        raise OSError(f"{LIB_PATH}: ABI version {lib.dbg_abi_version()} != binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def default_device():
    for var in ("DBG_DEVICE", "LOCAL_RANK"):
        if os.environ.get(var, "") != "":
            return int(os.environ[var])
    return 0


class Graph:
    """One handle of the C ABI: a read set plus the graph built from it, resident on one GPU."""

    def __init__(self, device=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        self._device_index = default_device() if device is None else int(device)
        rc = self._lib.dbg_create(self._device_index, C.byref(self._h))
        if rc != DBG_OK:
            self._h = None
            raise DbgError(rc, "dbg_create failed: no usable MI355X visible (the device path has no CPU fallback)")
        self._keep = []  # buffers the device borrows
        self.generation = 0  # bumped whenever the handle's graph or reads are replaced (debruijn.output_contigs checks it)
        self.walks = 0       # bumped by every walk: the device contigs of an earlier walk are gone (ContigList.sorted_fasta checks it)
        for var, opt in (("DBG_ENGINE", "engine"), ("DBG_BUCKET_BITS", "bucket_bits"), ("DBG_LDS_SLOTS", "lds_slots"),
                         ("DBG_WALK_JUMP_MIN", "walk_jump_min_nodes"), ("DBG_STAMP64", "stamp64")):
            if os.environ.get(var, "") != "":
                self.set_option(opt, int(os.environ[var]))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dbg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != DBG_OK:
            msg = self._lib.dbg_last_error(self._h).decode("utf-8", "replace")
            raise (AlphabetError if rc == DBG_E_ALPHABET else DbgError)(rc, msg)

    def set_option(self, name, value):
        self._chk(self._lib.dbg_set_option(self._h, name.encode(), int(value)))

    # ---- reads
    def set_reads(self, bases, offsets):
        """bases: bytes-like of concatenated reads; offsets: uint64[n_reads+1]."""
        b = np.frombuffer(bases, dtype=np.uint8) if not isinstance(bases, np.ndarray) else bases
        b = np.ascontiguousarray(b, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        assert o.ndim == 1 and o.size >= 1 and int(o[-1]) == b.size
        self.generation += 1
        self._chk(self._lib.dbg_set_reads(self._h, _ptr(b), _ptr(o), o.size - 1))

    def set_reads_fasta(self, path_or_bytes):
        """Parses a FASTA file on the GPU (read_reads semantics); no Python strings are created."""
        if isinstance(path_or_bytes, np.ndarray):
            raw = np.ascontiguousarray(path_or_bytes, dtype=np.uint8).reshape(-1)
        elif isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
            raw = np.frombuffer(path_or_bytes, dtype=np.uint8)
        else:
            raw = np.fromfile(path_or_bytes, dtype=np.uint8)
        self.generation += 1
        self._chk(self._lib.dbg_set_reads_fasta(self._h, _ptr(raw) if raw.size else None, raw.size))

    def set_reads_device(self, bases_ptr, n_bytes, offsets_ptr, n_reads, keepalive=()):
        self._keep = list(keepalive)
        self.generation += 1
        self._chk(self._lib.dbg_set_reads_device(self._h, C.c_void_p(bases_ptr), n_bytes, C.c_void_p(offsets_ptr),
                                                 n_reads))

    def synth_reads(self, seed, genome_len, n_reads, read_len, err_rate=0.0, first_read=0):
        thr = int(round(float(err_rate) * (1 << 24)))
        self.generation += 1
        self._chk(self._lib.dbg_synth_reads(self._h, seed, genome_len, first_read, n_reads, read_len, thr))

    def reads_checksum(self):
        out = C.c_uint64()
        self._chk(self._lib.dbg_reads_checksum(self._h, C.byref(out)))
        return out.value

    def take_reads(self, indices):
        """-> (chars uint8, offsets uint64[n + 1]): the selected reads, concatenated, gathered on the device."""
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        off = np.zeros(idx.size + 1, dtype=np.uint64)
        total = C.c_uint64()
        self._chk(self._lib.dbg_take_reads(self._h, _ptr(idx) if idx.size else None, idx.size, _ptr(off), None, 0, C.byref(total)))
        chars = np.empty(total.value, dtype=np.uint8)
        if total.value:
            self._chk(self._lib.dbg_take_reads(self._h, _ptr(idx), idx.size, None, _ptr(chars), chars.size, C.byref(total)))
        return chars, off

    def copy_reads(self):
        s = self.sizes()
        b = np.empty(s["n_bytes"], dtype=np.uint8)
        o = np.empty(s["n_reads"] + 1, dtype=np.uint64)
        self._chk(self._lib.dbg_copy_reads(self._h, _ptr(b), _ptr(o)))
        return b, o

    # ---- pipeline
    def build(self, k, table_capacity_hint=0):
        self.generation += 1
        self._chk(self._lib.dbg_build(self._h, int(k), int(table_capacity_hint)))

    def refine_edge_order(self):
        self._chk(self._lib.dbg_refine_edge_order(self._h))

    def export_orders(self):
        """Successor codes by rank, as (n_nodes, max_degree) arrays (0xFF beyond the out-degree for the generic
        layout): Counter.most_common order and first-seen order."""
        sz = self.sizes()
        n, d = sz["n_nodes"], sz["max_degree"]
        if d == 4:  # packed: 2 bits per rank
            a, b = np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.uint8)
            self._chk(self._lib.dbg_export_orders(self._h, _ptr(a), _ptr(b)))
            sh = np.array([0, 2, 4, 6], dtype=np.uint8)[None, :]
            return (a[:, None] >> sh) & 3, (b[:, None] >> sh) & 3
        a, b = np.empty((n, d), dtype=np.uint8), np.empty((n, d), dtype=np.uint8)
        self._chk(self._lib.dbg_export_orders(self._h, _ptr(a), _ptr(b)))
        return a, b

    def prune(self, threshold):
        self._chk(self._lib.dbg_prune(self._h, float(threshold)))

    def remove_tips(self):
        self._chk(self._lib.dbg_remove_tips(self._h))

    def mark_pull_reads(self):
        self._chk(self._lib.dbg_mark_pull_reads(self._h))

    def walk(self, final_mode, max_chars=0):
        self.walks += 1
        self._chk(self._lib.dbg_walk(self._h, 1 if final_mode else 0, int(max_chars)))

    def sizes(self):
        s = Sizes()
        self._chk(self._lib.dbg_get_sizes(self._h, C.byref(s)))
        return s.as_dict()

    def stats(self):
        s = Stats()
        self._chk(self._lib.dbg_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    # ---- exports
    def alphabet(self):
        """(bytes: code -> character, bits per symbol) of the built graph: b"ACTG"/2 for DNA reads, else byte order/5."""
        buf = (C.c_char * 32)()
        n, bits = C.c_int(), C.c_int()
        self._chk(self._lib.dbg_get_alphabet(self._h, buf, C.byref(n), C.byref(bits)))
        return bytes(buf.raw[:n.value]), bits.value

    def export_keepmask(self):
        m = np.empty(self.sizes()["n_nodes"], dtype=np.uint32)
        self._chk(self._lib.dbg_export_keepmask(self._h, _ptr(m)))
        return m

    def export_nodes(self, keys=True, stamps=True, counts=True, flags=True):
        sz = self.sizes()
        n, d = sz["n_nodes"], sz["max_degree"]
        a = np.empty(n, dtype=np.uint64) if keys else None
        b = np.empty(n, dtype=np.uint64) if stamps else None
        c = np.empty((n, d), dtype=np.uint32) if counts else None
        d = np.empty(n, dtype=np.uint8) if flags else None
        self._chk(self._lib.dbg_export_nodes(self._h, _ptr(a), _ptr(b), _ptr(c), _ptr(d)))
        return a, b, c, d

    def export_keys_hi(self):
        """Upper words of the node k-mers (k > 31: k-mer = hi * 2**64 + lo); zeros for k <= 31."""
        hi = np.empty(self.sizes()["n_nodes"], dtype=np.uint64)
        self._chk(self._lib.dbg_export_keys_hi(self._h, _ptr(hi)))
        return hi

    def export_succ(self):
        sz = self.sizes()
        n = sz["n_nodes"]
        s = np.empty((n, sz["max_degree"]), dtype=np.uint32)
        self._chk(self._lib.dbg_export_succ(self._h, _ptr(s)))
        return s

    def export_csr(self):
        sz = self.sizes()
        rp = np.empty(sz["n_nodes"] + 1, dtype=np.uint64)
        col = np.empty(sz["n_edges"], dtype=np.uint32)
        cnt = np.empty(sz["n_edges"], dtype=np.uint32)
        self._chk(self._lib.dbg_export_csr(self._h, _ptr(rp), _ptr(col), _ptr(cnt)))
        return rp, col, cnt

    def export_dict_order(self):
        """Node ids in first-occurrence (dict) order: the argsort of the stamps, sorted on the device."""
        o = np.empty(self.sizes()["n_nodes"], dtype=np.uint32)
        self._chk(self._lib.dbg_export_dict_order(self._h, _ptr(o)))
        return o

    def export_pull_ranks(self):
        r = np.empty(self.sizes()["n_nodes"], dtype=np.uint64)
        self._chk(self._lib.dbg_export_pull_ranks(self._h, _ptr(r)))
        return r

    def export_marked(self, flag, keys=True):
        """Rows, keys, keys_hi of the nodes carrying ``flag`` (F_PULLED: pull order; F_BRANCH: dict order), selected
        and sorted on the device.  keys=False: rows only (graphs keyed by reference hold no packed k-mers)."""
        n = C.c_uint64()
        self._chk(self._lib.dbg_export_marked(self._h, int(flag), 0, C.byref(n), None, None, None))
        m = int(n.value)
        rows = np.empty(m, np.uint32)
        kk, hi = (np.empty(m, np.uint64), np.empty(m, np.uint64)) if keys else (None, None)
        if m:
            self._chk(self._lib.dbg_export_marked(self._h, int(flag), m, C.byref(n), _ptr(rows), _ptr(kk) if keys else None,
                                                  _ptr(hi) if keys else None))
        return rows, kk, hi

    def export_pull_reads(self):
        f = np.empty(self.sizes()["n_reads"], dtype=np.uint8)
        self._chk(self._lib.dbg_export_pull_reads(self._h, _ptr(f)))
        return f

    def export_contig_index(self):
        n = self.sizes()["n_contigs"]
        off = np.empty(n + 1, dtype=np.uint64)
        score = np.empty(n, dtype=np.uint64)
        stamp = np.empty(n, dtype=np.uint64)
        seq = np.empty(n, dtype=np.uint32)
        self._chk(self._lib.dbg_export_contig_index(self._h, _ptr(off), _ptr(score), _ptr(stamp), _ptr(seq)))
        return off, score, stamp, seq

    def export_contig_text(self, index, length):
        """Text of one contig (bytes) -- also when the whole text was larger than the walk's max_chars."""
        buf = np.empty(int(length), dtype=np.uint8)
        self._chk(self._lib.dbg_export_contig_text(self._h, int(index), _ptr(buf), int(length)))
        return buf.tobytes()

    def export_contigs(self):
        sz = self.sizes()
        n, nc = sz["n_contigs"], sz["contig_chars"]
        off = np.empty(n + 1, dtype=np.uint64)
        chars = np.empty(nc, dtype=np.uint8)
        score = np.empty(n, dtype=np.uint64)
        stamp = np.empty(n, dtype=np.uint64)
        seq = np.empty(n, dtype=np.uint32)
        self._chk(self._lib.dbg_export_contigs(self._h, _ptr(off), _ptr(chars), _ptr(score), _ptr(stamp), _ptr(seq)))
        return off, chars, score, stamp, seq


    def export_sorted_fasta(self):
        """The contigs of the last (materialised) walk, sorted by score like the reference's driver, as FASTA text
        (II_assembleFromReads.py:64-69) -> (bytes, order uint32[n_contigs])."""
        n = C.c_uint64()
        self._chk(self._lib.dbg_export_sorted_fasta(self._h, None, None, 0, C.byref(n)))
        buf = np.empty(n.value, dtype=np.uint8)
        order = np.empty(self.sizes()["n_contigs"], dtype=np.uint32)
        self._chk(self._lib.dbg_export_sorted_fasta(self._h, _ptr(order) if order.size else None, _ptr(buf) if buf.size else None,
                                                    buf.size, C.byref(n)))
        return buf.tobytes(), order

    # ---- read-support scores (IV_sortOutputs.py:10-15)
    def support_read_scores(self, read_chars, read_off, scores, is_float, contig_chars, contig_off):
        """-> (float64 scores [n_contigs], uint32 float-typed hits [n_contigs]); numpy arrays in, see include/dbg.h."""
        n_reads, n_contigs = read_off.size - 1, contig_off.size - 1
        out = np.zeros(n_contigs, dtype=np.float64)
        fh = np.zeros(n_contigs, dtype=np.uint32)
        self._chk(self._lib.dbg_support_read_scores(
            self._h, _ptr(read_chars) if read_chars.size else None, _ptr(read_off), n_reads, _ptr(scores) if n_reads else None,
            _ptr(is_float) if n_reads else None, _ptr(contig_chars) if contig_chars.size else None, _ptr(contig_off), n_contigs,
            _ptr(out) if n_contigs else None, _ptr(fh) if n_contigs else None))
        return out, fh

    # ---- multi-pass build: graphs beyond one 32-bit id space (BASELINE.json configs[3])
    def build_multipass(self, k, n_passes):
        self.generation += 1
        self._chk(self._lib.dbg_build_multipass(self._h, int(k), int(n_passes)))
        self._mp_virtual = int(n_passes)  # virtual shards = parts on one GPU

    # ---- ranks x passes: a rank of a sharded build that builds its shard in parts (multi_gpu.sharded_build_multipass)
    def shard_build_multipass(self, k, n_shards, my_shard, n_passes, w0, w1, st, recv_counts, stamp_base, sender_bucket_counts):
        """Received records (torch tensors, split by the senders' level-1 groups) -> n_passes parts on this handle; part p
        is virtual shard my_shard * n_passes + p.  Successors owned by other ranks stay open (part_queries).  The arrays hold
        len(recv_counts) messages (n_shards, or a multiple when the ranks sent their records in parts: shard_extract_part)."""
        self.generation += 1
        n_senders = len(recv_counts)
        rc = (C.c_uint64 * n_senders)(*[int(x) for x in recv_counts])
        sb = (C.c_uint64 * n_senders)(*[int(x) for x in stamp_base])
        flat = [int(x) for row in sender_bucket_counts for x in row]
        if n_shards < 1 or len(stamp_base) != n_senders or len(flat) != n_senders * (512 // n_shards):
            raise DbgError(-1, "shard_build_multipass: one stamp base per sender and one count per (sender, owned level-1 group)")
        sbc = (C.c_uint64 * len(flat))(*flat)
        self._keep = [w0, w1, st]
        self._mp_virtual = n_shards * n_passes
        self._chk(self._lib.dbg_shard_build_multipass_from(self._h, int(k), int(n_shards), int(my_shard), int(n_passes), n_senders,
                                                           C.c_void_p(w0.data_ptr()), C.c_void_p(w1.data_ptr()),
                                                           C.c_void_p(st.data_ptr()), int(st.element_size()), rc, sb, sbc))
        self._keep = []

    def part_queries(self, part):
        """-> (q_starts, q_counts, keys tensor): the part's open successor k-mers grouped by owning virtual shard."""
        nv = getattr(self, "_mp_virtual", 0)
        if not nv:
            raise DbgError(-1, "part_queries: a multi-pass build must run first")
        qs, qc = (C.c_uint64 * nv)(), (C.c_uint64 * nv)()
        pk = C.c_void_p()
        self._chk(self._lib.dbg_part_queries(self._h, int(part), qs, qc, C.byref(pk)))
        qs, qc = [int(x) for x in qs], [int(x) for x in qc]
        total = max([a + b for a, b in zip(qs, qc) if b] + [0])
        self._q_words = 2 if self.sizes()["k"] > 31 else 1  # two-word k-mers: a query is a (lo, hi) pair
        return qs, qc, device_tensor(pk.value, total * self._q_words, "int64", self.sizes_device())

    def part_answer(self, part, keys):
        import torch
        n = keys.numel() // self.query_words()
        ans = torch.empty(n, dtype=torch.int32, device=keys.device)
        if n:
            self._chk(self._lib.dbg_part_answer(self._h, int(part), C.c_void_p(keys.data_ptr()), n, C.c_void_p(ans.data_ptr())))
        return ans

    def part_apply(self, part, owner, answers):
        self._chk(self._lib.dbg_part_apply(self._h, int(part), int(owner), C.c_void_p(answers.data_ptr()) if answers.numel() else None))

    def multipass_finish(self):
        self._chk(self._lib.dbg_multipass_finish(self._h))

    def part_count(self):
        n = C.c_int()
        self._chk(self._lib.dbg_part_count(self._h, C.byref(n)))
        return n.value

    def part_sizes(self, part):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self._lib.dbg_part_sizes(self._h, int(part), C.byref(a), C.byref(b), C.byref(c)))
        return {"n_nodes": a.value, "n_edges": b.value, "first_node_id": c.value}

    def export_part(self, part):
        """-> dict(keys, stamps, flags, row_ptr, col, col_part, cnt) of one part (numpy)."""
        sz = self.part_sizes(part)
        n, ne = sz["n_nodes"], sz["n_edges"]
        out = {"keys": np.empty(n, np.uint64), "stamps": np.empty(n, np.uint64), "flags": np.empty(n, np.uint8),
               "row_ptr": np.empty(n + 1, np.uint64), "col": np.empty(ne, np.uint32), "col_part": np.empty(ne, np.uint8),
               "cnt": np.empty(ne, np.uint32), "keys_hi": np.zeros(n, np.uint64)}
        self._chk(self._lib.dbg_export_part(self._h, int(part), *[_ptr(out[x]) for x in
                                                                   ("keys", "stamps", "flags", "row_ptr", "col", "col_part", "cnt")]))
        self._chk(self._lib.dbg_part_keys_hi(self._h, int(part), _ptr(out["keys_hi"]) if n else None, None))
        return out

    def part_tensors(self, part):
        """Zero-copy torch views of one part: keys int64, stamps int32/int64, flags uint8, row_ptr int32 [n + 1],
        col int32, col_part uint8, cnt int32."""
        p = [C.c_void_p() for _ in range(7)]
        sb = C.c_int()
        self._chk(self._lib.dbg_part_device_views(self._h, int(part), C.byref(p[0]), C.byref(p[1]), C.byref(sb), C.byref(p[2]),
                                                  C.byref(p[3]), C.byref(p[4]), C.byref(p[5]), C.byref(p[6])))
        sz, dev = self.part_sizes(part), self.sizes_device()
        n, ne = sz["n_nodes"], sz["n_edges"]
        if n == 0:  # a part whose level-1 groups hold no record has no arrays on the device
            import torch
            d = torch.device("cuda", dev) if isinstance(dev, int) else torch.device(dev)
            e = lambda dt: torch.empty(0, dtype=dt, device=d)  # noqa: E731
            return {"keys": e(torch.int64), "stamps": e(torch.int64), "flags": e(torch.uint8),
                    "row_ptr": torch.zeros(1, dtype=torch.int32, device=d), "col": e(torch.int32), "col_part": e(torch.uint8),
                    "cnt": e(torch.int32)}
        hi = C.c_void_p()
        self._chk(self._lib.dbg_part_keys_hi(self._h, int(part), None, C.byref(hi)))
        extra = {"keys_hi": device_tensor(hi.value, n, "int64", dev)} if hi.value else {}
        return {**extra, "keys": device_tensor(p[0].value, n, "int64", dev),
                "stamps": device_tensor(p[1].value, n, "int32" if sb.value == 4 else "int64", dev),
                "flags": device_tensor(p[2].value, n, "uint8", dev), "row_ptr": device_tensor(p[3].value, n + 1, "int32", dev),
                "col": device_tensor(p[4].value, ne, "int32", dev), "col_part": device_tensor(p[5].value, ne, "uint8", dev),
                "cnt": device_tensor(p[6].value, ne, "int32", dev)}

    # ---- traversal of a graph in parts (part_traversal.py): per-part primitives; ids and rows are torch tensors on the device
    def _dev(self):
        import torch
        return torch.device("cuda", self.sizes_device())

    def part_prune(self, part, threshold):
        n = C.c_uint64()
        self._chk(self._lib.dbg_part_prune(self._h, int(part), float(threshold), C.byref(n)))
        return n.value

    def part_select(self, part, mask, want):
        """-> int32 tensor of the local ids (as uint32) with (flags & mask) == want, ascending."""
        import torch
        n = C.c_uint64()
        self._chk(self._lib.dbg_part_select(self._h, int(part), int(mask), int(want), None, 0, C.byref(n)))
        ids = torch.empty(n.value, dtype=torch.int32, device=self._dev())
        if n.value:
            self._chk(self._lib.dbg_part_select(self._h, int(part), int(mask), int(want), C.c_void_p(ids.data_ptr()), n.value, C.byref(n)))
        return ids

    def part_gather(self, part, ids, what=("keys", "keys_hi", "stamps", "counts", "succ_owner", "succ_local", "pflags")):
        """rows of the nodes `ids` (int32 device tensor): dict of device tensors for the names in `what`."""
        import torch
        n, dev = ids.numel(), self._dev()
        shapes = {"keys": (torch.int64, n), "keys_hi": (torch.int64, n), "stamps": (torch.int64, n), "counts": (torch.int32, 4 * n),
                  "succ_owner": (torch.uint8, 4 * n), "succ_local": (torch.int32, 4 * n), "pflags": (torch.uint8, n)}
        out = {w: torch.empty(shapes[w][1], dtype=shapes[w][0], device=dev) for w in what}
        if n:
            ptr = lambda w: C.c_void_p(out[w].data_ptr()) if w in out else None  # noqa: E731
            self._chk(self._lib.dbg_part_gather(self._h, int(part), C.c_void_p(ids.data_ptr()), n, ptr("keys"), ptr("keys_hi"),
                                                ptr("stamps"), ptr("counts"), ptr("succ_owner"), ptr("succ_local"), ptr("pflags")))
        return out

    def part_mark(self, part, ids, bits, newly=False):
        import torch
        n = ids.numel()
        flag = torch.zeros(n, dtype=torch.uint8, device=self._dev()) if newly else None
        if n:
            self._chk(self._lib.dbg_part_mark(self._h, int(part), C.c_void_p(ids.data_ptr()), n, int(bits),
                                              C.c_void_p(flag.data_ptr()) if newly else None))
        return flag

    def part_clear(self, part, bits):
        self._chk(self._lib.dbg_part_clear(self._h, int(part), int(bits)))

    def part_cross_targets(self, part):
        """-> (counts per virtual shard, int32 tensor of target local ids grouped by virtual shard)."""
        import torch
        nv = getattr(self, "_mp_virtual", 0)
        counts = (C.c_uint64 * max(nv, 1))()
        self._chk(self._lib.dbg_part_cross_targets(self._h, int(part), counts, None, 0))
        counts_l = [int(x) for x in counts][:nv]
        total = sum(counts_l)
        t = torch.empty(total, dtype=torch.int32, device=self._dev())
        if total:
            self._chk(self._lib.dbg_part_cross_targets(self._h, int(part), counts, C.c_void_p(t.data_ptr()), total))
        return counts_l, t

    def part_segments(self, part, entries):
        import torch
        n, dev = entries.numel(), self._dev()
        out = {"kind": torch.empty(n, dtype=torch.uint8, device=dev), "next_owner": torch.empty(n, dtype=torch.uint8, device=dev),
               "next_local": torch.empty(n, dtype=torch.int32, device=dev), "hops": torch.empty(n, dtype=torch.int32, device=dev),
               "score": torch.empty(n, dtype=torch.int64, device=dev), "last": torch.empty(n, dtype=torch.int32, device=dev)}
        if n:
            self._chk(self._lib.dbg_part_segments(self._h, int(part), C.c_void_p(entries.data_ptr()), n,
                                                  *[C.c_void_p(out[w].data_ptr()) for w in ("kind", "next_owner", "next_local", "hops", "score", "last")]))
        return out

    def part_segment_text(self, part, entries, off):
        """characters of the segments at `entries` (int32 device tensor); off: int64 device tensor [n + 1] -> uint8 device tensor"""
        import torch
        total = int(off[-1].item()) if off.numel() else 0
        chars = torch.empty(total, dtype=torch.uint8, device=self._dev())
        if entries.numel() and total:
            self._chk(self._lib.dbg_part_segment_text(self._h, int(part), C.c_void_p(entries.data_ptr()), entries.numel(),
                                                      C.c_void_p(off.data_ptr()), C.c_void_p(chars.data_ptr()), total))
        return chars

    def scan_reads_for_keys(self, k, keys, keys_hi=None, first_seen=True):
        """-> (read_flags uint8[n_reads], first_seen uint64[n_keys, 4] or None); numpy in and out (dbg_scan_reads_for_keys)."""
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        hi = None if keys_hi is None else np.ascontiguousarray(keys_hi, dtype=np.uint64)
        rf = np.zeros(self.sizes()["n_reads"], dtype=np.uint8)
        fs = np.full((keys.size, 4), np.iinfo(np.uint64).max, dtype=np.uint64) if first_seen else None
        self._chk(self._lib.dbg_scan_reads_for_keys(self._h, int(k), _ptr(keys) if keys.size else None, _ptr(hi) if hi is not None and hi.size else None,
                                                    keys.size, _ptr(rf) if rf.size else None, _ptr(fs) if fs is not None and fs.size else None))
        return rf, fs

    def set_orders(self, order):
        order = np.ascontiguousarray(order, dtype=np.uint8)
        self._chk(self._lib.dbg_set_orders(self._h, _ptr(order)))

    # ---- multi-GPU sharding (buffers are torch tensors on this handle's device; see multi_gpu.py)
    def shard_extract(self, k, n_shards):
        """-> (send_counts list, (w0, w1, st) tensors viewing library memory, grouped by owner).
        w0 holds ``shard_record_layout()[0]`` words per record (1, or 4 for the two-word k-mers' records by value)."""
        counts = (C.c_uint64 * n_shards)()
        p0, p1, p2 = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.generation += 1  # the extraction frees the graph of an earlier build
        self._chk(self._lib.dbg_shard_extract(self._h, int(k), int(n_shards), counts, C.byref(p0), C.byref(p1),
                                              C.byref(p2)))
        counts = [int(c) for c in counts]
        n = sum(counts)
        dev = self.sizes_device()
        words, st_bytes = self.shard_record_layout()
        # k <= 31: super-k-mer records (w0, w1, 32-bit stamp); k > 31: records by value (4 words of bases, meta, 32-bit stamp)
        # or, with "wide_engine" 0, k-mer instances (lo, hi | next << 62, 64-bit meta)
        return counts, (device_tensor(p0.value, words * n, "int64", dev), device_tensor(p1.value, n, "int64", dev),
                        device_tensor(p2.value, n, "int32" if st_bytes == 4 else "int64", dev))

    def shard_extract_part(self, k, n_shards, part, n_parts):
        """shard_extract for slice ``part`` of ``n_parts`` of this rank's reads (k <= 31); the tensors of a part stay valid
        until that part is extracted again."""
        counts = (C.c_uint64 * n_shards)()
        p0, p1, p2 = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self.generation += 1
        self._chk(self._lib.dbg_shard_extract_part(self._h, int(k), int(n_shards), int(part), int(n_parts), counts, C.byref(p0),
                                                   C.byref(p1), C.byref(p2)))
        counts = [int(c) for c in counts]
        n = sum(counts)
        dev = self.sizes_device()
        words, st_bytes = self.shard_record_layout()
        return counts, (device_tensor(p0.value, words * n, "int64", dev), device_tensor(p1.value, n, "int64", dev),
                        device_tensor(p2.value, n, "int32" if st_bytes == 4 else "int64", dev))

    def shard_record_layout(self):
        """(64-bit words per record in the first tensor of shard_extract, bytes per stamp)."""
        a, b = C.c_int(), C.c_int()
        self._chk(self._lib.dbg_shard_record_layout(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sizes_device(self):
        return getattr(self, "_device_index", default_device())

    def shard_bucket_counts(self):
        """Records per level-1 bucket (512) of the last shard_extract (k <= 31): owners are contiguous ranges of them."""
        out = (C.c_uint64 * 512)()
        self._chk(self._lib.dbg_shard_bucket_counts(self._h, out))
        return [int(x) for x in out]

    def shard_build(self, k, n_shards, my_shard, w0, w1, st, recv_counts, stamp_base, sender_bucket_counts=None):
        """Received records (torch tensors) -> (q_starts, q_counts, query key tensor grouped by owner).
        sender_bucket_counts[r] = sender r's record counts for the 512 / n_shards level-1 buckets this shard owns
        (from its shard_bucket_counts): the build then starts at the second multisplit level."""
        rc = (C.c_uint64 * n_shards)(*[int(x) for x in recv_counts])
        sb = (C.c_uint64 * n_shards)(*[int(x) for x in stamp_base])
        qs, qc = (C.c_uint64 * n_shards)(), (C.c_uint64 * n_shards)()
        pk = C.c_void_p()
        sbc = None
        if sender_bucket_counts is not None:
            flat = [int(x) for row in sender_bucket_counts for x in row]
            assert len(flat) == 512, "one count per (sender, owned level-1 bucket)"
            sbc = (C.c_uint64 * 512)(*flat)
        self._keep = [w0, w1, st]
        self.generation += 1
        self._chk(self._lib.dbg_shard_build(self._h, int(k), int(n_shards), int(my_shard), C.c_void_p(w0.data_ptr()),
                                            C.c_void_p(w1.data_ptr()), C.c_void_p(st.data_ptr()), rc, sb, qs, qc,
                                            C.byref(pk), sbc, int(st.element_size())))
        self._keep = []
        qs, qc = [int(x) for x in qs], [int(x) for x in qc]
        total = max([a + b for a, b in zip(qs, qc)] + [0])
        # two-word k-mers (records by value): a query is a (lo, hi) pair -- query_words() 64-bit words per query
        self._q_words = 2 if (int(k) > 31 and self.shard_record_layout()[0] == 4) else 1
        return qs, qc, device_tensor(pk.value, total * self._q_words, "int64", self.sizes_device())

    def query_words(self):
        """64-bit words per successor query of the last shard_build / part_queries (2 for two-word k-mers)."""
        return getattr(self, "_q_words", 1)

    def shard_answer(self, keys):
        import torch
        n = keys.numel() // self.query_words()
        ans = torch.empty(n, dtype=torch.int32, device=keys.device)
        self._chk(self._lib.dbg_shard_answer(self._h, C.c_void_p(keys.data_ptr()), n, C.c_void_p(ans.data_ptr())))
        return ans

    def shard_apply(self, answers):
        self.generation += 1
        self._chk(self._lib.dbg_shard_apply(self._h, C.c_void_p(answers.data_ptr()) if answers.numel() else None))

    # ---- gather for traversal (multi_gpu.gather_graph)
    def node_tensors(self):
        """Zero-copy torch views of this handle's node arrays: keys, stamps (int64 [n]), counts, succ (int32 [4 n])."""
        p = [C.c_void_p() for _ in range(5)]
        self._chk(self._lib.dbg_device_views(self._h, *[C.byref(x) for x in p]))
        n, dev = self.sizes()["n_nodes"], self.sizes_device()
        out = {"keys": device_tensor(p[0].value, n, "int64", dev), "stamps": device_tensor(p[2].value, n, "int64", dev),
               "counts": device_tensor(p[1].value, 4 * n, "int32", dev), "succ": device_tensor(p[4].value, 4 * n, "int32", dev)}
        hi = C.c_void_p()
        self._chk(self._lib.dbg_device_keys_hi(self._h, C.byref(hi)))
        if hi.value:  # two-word k-mers
            out["keys_hi"] = device_tensor(hi.value, n, "int64", dev)
        return out

    def reads_tensors(self):
        """Zero-copy torch views of the resident reads: (bases uint8 [n_bytes], offsets int64 [n_reads + 1])."""
        pb, po, nb, nr = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        self._chk(self._lib.dbg_reads_device(self._h, C.byref(pb), C.byref(nb), C.byref(po), C.byref(nr)))
        dev = self.sizes_device()
        return device_tensor(pb.value, nb.value, "uint8", dev), device_tensor(po.value, nr.value + 1, "int64", dev)

    def set_reads_tensors(self, bases, offsets):
        """Adopt reads held in torch tensors on this handle's device (kept alive by the handle)."""
        self.set_reads_device(bases.data_ptr(), bases.numel(), offsets.data_ptr(), offsets.numel() - 1,
                              keepalive=(bases, offsets))

    def import_graph(self, k, shard_nodes, keys, stamps, counts, succ, keys_hi=None):
        """Install the concatenated shard arrays (torch tensors on this device) as this handle's graph."""
        sn = (C.c_uint64 * len(shard_nodes))(*[int(x) for x in shard_nodes])
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else None
        self.generation += 1
        self._graph_keep = (keys, keys_hi, stamps, counts, succ)  # borrowed by the library, not copied
        self._chk(self._lib.dbg_import_graph(self._h, int(k), len(shard_nodes), sn, ptr(keys), ptr(keys_hi), ptr(stamps),
                                             ptr(counts), ptr(succ)))


class _DevView:
    """Zero-copy torch view of library-owned device memory (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_tensor(ptr, n, dtype, device_index):
    import torch
    tdtype = {"int64": torch.int64, "int32": torch.int32, "uint8": torch.uint8}[dtype]
    if n == 0 or not ptr:
        return torch.empty(0, dtype=tdtype, device=f"cuda:{device_index}")
    return torch.as_tensor(_DevView(ptr, n, {"int64": "<i8", "int32": "<i4", "uint8": "|u1"}[dtype]), device=f"cuda:{device_index}")


# ---- 2-bit key <-> str helpers (host side of the boundary) -------------------------------
def decode_keys(keys, k, alphabet=b"ACTG", bits=2, keys_hi=None):
    """uint64 keys -> list of k-character str (alphabet[code] = character; DNA: code = (ascii >> 1) & 3).

    keys_hi: upper words for k-mers wider than one word (k > 32 at 2 bits): k-mer = hi * 2**64 + lo."""
    keys = np.asarray(keys, dtype=np.uint64)
    if keys.size == 0:
        return []
    table = np.frombuffer(alphabet.ljust(1 << bits, b"?"), dtype=np.uint8)
    shifts = bits * (k - 1 - np.arange(k, dtype=np.int64))
    mask = np.uint64((1 << bits) - 1)
    lo_cols = shifts < 64
    codes = np.empty((keys.size, k), dtype=np.intp)
    codes[:, lo_cols] = ((keys[:, None] >> shifts[lo_cols].astype(np.uint64)[None, :]) & mask).astype(np.intp)
    if not lo_cols.all():
        if keys_hi is None:
            raise ValueError("k-mers wider than 64 bits need keys_hi")
        hi = np.asarray(keys_hi, dtype=np.uint64)
        codes[:, ~lo_cols] = ((hi[:, None] >> (shifts[~lo_cols] - 64).astype(np.uint64)[None, :]) & mask).astype(np.intp)
    buf = table[codes].tobytes().decode("latin-1")
    return [buf[i * k:(i + 1) * k] for i in range(keys.size)]


def encode_kmer(s):
    key = 0
    for ch in s.encode("ascii"):
        key = (key << 2) | ((ch >> 1) & 3)
    return key
