import glob
import json
import os
import sys

import pytest

try:  # torch bundles a HIP runtime of its own: it must be in the process before anything loads the system one
    import torch  # noqa: F401  (see _dbg.load_library; some tests open the library with ctypes directly)
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "py-debruijn_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as fh:
        return json.load(fh)


def golden_case_names():
    """construct_graph/output_contigs cases (one JSON each), excluding driver/fuzz/synth/support-score files."""
    names = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.json"))):
        n = os.path.basename(p)[:-5]
        if n.startswith(("driver_", "fuzz_", "synth_", "aux_", "support_")):
            continue
        names.append(n)
    return names


def case_reads(case):
    """Reads of a golden case: stored inline, or regenerated from the committed generator params."""
    inp = case["inputs"]
    if "reads" in inp:
        return list(inp["reads"])
    import synth
    g = inp["generator"]
    arr = synth.reads_ascii(g["seed"], g["genome_len"], g["n_reads"], g["read_len"], g["err_rate"])
    assert synth.checksum(arr) == inp["reads_checksum"], "synthetic generator drifted from the fixture"
    return [row.tobytes().decode("ascii") for row in arr]
