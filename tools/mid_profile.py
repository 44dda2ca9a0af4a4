#!/usr/bin/env python3
"""cProfile of the drop-in construct_graph at 1M x 150 bp (GPU box)."""
import contextlib, cProfile, io, pstats, sys
sys.path.insert(0, "py-debruijn_amd")
import _dbg
import debruijn as prod
n, L, k = 1_000_000, 150, 31
dev = _dbg.Graph()
dev.synth_reads(1, n * 5, n, L, 0.01)
reads = prod.DeviceReads.__new__(prod.DeviceReads)
reads._graph, reads._n, reads._host = dev, n, None
with contextlib.redirect_stdout(io.StringIO()):
    prod.construct_graph(reads, k, threshold=2)
    pr = cProfile.Profile()
    pr.enable()
    prod.construct_graph(reads, k, threshold=2)
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
