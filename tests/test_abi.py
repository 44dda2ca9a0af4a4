"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol of include/dbg.h."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import PKG, ROOT

LIB = os.path.join(PKG, "libdbg_hip.so")


def header_functions():
    text = open(os.path.join(ROOT, "include", "dbg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dbg_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        subprocess.check_call([os.path.join(PKG, "csrc", "build.sh")])
    return ctypes.CDLL(LIB)


def test_header_declares_expected_surface():
    fns = header_functions()
    for must in ("dbg_create", "dbg_destroy", "dbg_set_reads", "dbg_build", "dbg_prune", "dbg_remove_tips",
                 "dbg_mark_pull_reads", "dbg_walk", "dbg_export_nodes", "dbg_export_csr", "dbg_export_contigs"):
        assert must in fns


def test_library_exports_every_declared_symbol(lib):
    for name in header_functions():
        assert hasattr(lib, name), f"{name} declared in include/dbg.h but not exported"


def test_binding_covers_header():
    import _dbg
    assert sorted(_dbg.SYMBOLS) == header_functions()
    assert _dbg.load_library().dbg_abi_version() == _dbg.ABI_VERSION


def test_struct_layouts_match_header(lib):
    import _dbg
    # dbg_sizes_t: 2 x int32 + 16 x uint64; dbg_stats_t: 14 doubles + 4 uint64
    assert ctypes.sizeof(_dbg.Sizes) == 8 + 16 * 8
    assert ctypes.sizeof(_dbg.Stats) == 14 * 8 + 4 * 8


def test_no_gpu_fails_loudly():
    """Without a GPU the product path must raise, never fall back to a CPU implementation."""
    import _dbg
    import debruijn
    try:
        _dbg.Graph().close()
    except _dbg.DbgError:
        pass  # no usable GPU: the case under test
    else:
        pytest.skip("GPU present")  # (asked of the library itself: torch.cuda.is_available() said False on a GPU box)
    with pytest.raises(_dbg.DbgError):
        debruijn.construct_graph(["ACGTACGT"], 3)


def test_product_does_not_import_oracle():
    for fn in os.listdir(PKG):
        if fn.endswith(".py"):
            src = open(os.path.join(PKG, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), f"{fn} must not reference the oracle"


def test_key_codec_roundtrip():
    import numpy as np
    import _dbg
    for s in ("A", "ACGT", "TTTTGGGGCCCCAAAA", "ACGTACGTACGTACGTACGTACGTACGTACG"):
        key = _dbg.encode_kmer(s)
        assert _dbg.decode_keys(np.array([key], dtype=np.uint64), len(s)) == [s]
    alpha = b"ACDEFGHIKLMNPQRSTVWY"  # peptides: 5 bits per character, codes in byte order
    key = 0
    for ch in b"EVQLV":
        key = (key << 5) | alpha.index(ch)
    assert _dbg.decode_keys(np.array([key], dtype=np.uint64), 5, alpha, 5) == ["EVQLV"]


@pytest.mark.gpu
def test_hidden_gpu_fails_loudly_on_a_gpu_box():
    """The GPU-present twin of test_no_gpu_fails_loudly: on a box that HAS a GPU the library works in this process, and a
    child process that cannot see the device (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES hide it) must get DbgError from the
    product path -- no CPU fallback takes over."""
    import sys
    import _dbg
    _dbg.Graph().close()  # this process sees the GPU
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import _dbg, debruijn\n"
            "try:\n"
            "    debruijn.construct_graph(['ACGTACGT'], 3)\n"
            "except _dbg.DbgError as e:\n"
            "    print('DbgError', e); sys.exit(42)\n"
            "sys.exit(0)\n") % PKG
    env = dict(os.environ, ROCR_VISIBLE_DEVICES="-1", HIP_VISIBLE_DEVICES="-1", CUDA_VISIBLE_DEVICES="-1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 42, (p.returncode, p.stdout[-500:], p.stderr[-500:])
    assert "no usable MI355X" in p.stdout or "dbg error" in p.stdout
