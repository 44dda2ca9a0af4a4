#!/usr/bin/env python3
"""BASELINE.json configs[0] through the drop-in surface (GPU box): 10k x 100 bp, k = 21, threshold 2 -- the case BASELINE.md
times the reference on (construct_graph 4.15 s, output_contigs 14.85 s at 1 % errors, one Xeon core)."""
import contextlib
import io
import sys
import time

sys.path.insert(0, "py-debruijn_amd")
import debruijn as prod
import synth

for err in (0.0, 0.01):
    reads = synth.reads_list(1, 100_000, 10_000, 100, err)
    for rep in range(2):
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            g, pull, branch, pulled, ect = prod.construct_graph(reads, 21, threshold=2)
            t1 = time.perf_counter()
            contigs = prod.output_contigs(g, branch, pulled)
            t2 = time.perf_counter()
        dev = g[0]._graph.stats()
        t3 = time.perf_counter(); exp = g[0]._graph.export_contigs(); t4 = time.perf_counter()
    print(f"err {err}: construct_graph {t1 - t0:.3f} s (device build {dev['ms_build_total']:.2f} ms), output_contigs {t2 - t1:.3f} s, "
          f"{len(g[0])} vertices, {len(contigs)} contigs, longest {max(map(len, contigs))} bp; device walk {dev['ms_walk']:.1f} ms, export {1e3 * (t4 - t3):.0f} ms")
