#!/usr/bin/env python3
"""FASTA ingest on the device at BASELINE size (GPU box): builds the text of 10M x 150 bp records in host memory,
times dbg_set_reads_fasta (H2D + parse) and checks the reads against the generator."""
import sys
import time

import numpy as np

sys.path.insert(0, "py-debruijn_amd")
import _dbg
import synth

n, L = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 10_000_000, 150
g = _dbg.Graph()
g.synth_reads(1, n * 5, n, L, 0.01)
bases, off = g.copy_reads()
want_sum = g.reads_checksum()
hdr = np.frombuffer(b">r\n", dtype=np.uint8)
rec = np.empty((n, 3 + L + 1), dtype=np.uint8)
rec[:, :3] = hdr
rec[:, 3:3 + L] = bases.reshape(n, L)
rec[:, -1] = 10
text = rec.reshape(-1)
print("text bytes", text.size, flush=True)
g2 = _dbg.Graph()
for _ in range(3):
    t = time.perf_counter()
    g2.set_reads_fasta(text)
    dt = time.perf_counter() - t
    print("set_reads_fasta ms", round(dt * 1e3, 1), "GB/s of text", round(text.size / dt / 1e9, 1), flush=True)
assert g2.sizes()["n_reads"] == n and g2.reads_checksum() == want_sum
print("ok")
