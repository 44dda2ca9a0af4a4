#!/usr/bin/env python3
"""The drop-in surface at 1M x 150 bp (GPU box): where the host time goes when the graph no longer fits Python dicts."""
import contextlib
import io
import sys
import time

sys.path.insert(0, "py-debruijn_amd")
import _dbg
import debruijn as prod

n, L, k = int(float(sys.argv[1]) * 1e6) if len(sys.argv) > 1 else 1_000_000, 150, 31
dev = _dbg.Graph()
dev.synth_reads(1, n * 5, n, L, 0.01)
reads = prod.DeviceReads.__new__(prod.DeviceReads)
reads._graph, reads._n, reads._host = dev, n, None
for rep in range(2):
    with contextlib.redirect_stdout(io.StringIO()):
        t0 = time.perf_counter()
        g, pull, branch, pulled, ect = prod.construct_graph(reads, k, threshold=2)
        t1 = time.perf_counter()
        contigs = prod.output_contigs(g, branch, pulled)
        t2 = time.perf_counter()
st = dev.stats()
print(f"{n} reads: construct_graph {t1 - t0:.2f} s, output_contigs {t2 - t1:.2f} s; {len(g[0])} vertices, {len(contigs)} contigs; "
      f"device: build {st['ms_build_total']:.1f} prune {st['ms_prune']:.1f} tips {st['ms_tips']:.1f} pull {st['ms_pull_reads']:.1f} walk {st['ms_walk']:.1f} ms")
if isinstance(contigs, prod.LazyContigs):  # text above the device budget: fetched per contig
    t0 = time.perf_counter()
    longest = max(range(len(contigs)), key=contigs.lengths.__getitem__)
    txt = contigs[longest]
    t1 = time.perf_counter()
    sample = contigs[0:1000]
    t2 = time.perf_counter()
    print(f"lazy contigs: {sum(contigs.lengths)} chars in total; longest {len(txt)} chars fetched in {t1 - t0:.3f} s (builds the "
          f"lifting tables), then 1000 contigs in {t2 - t1:.3f} s")
