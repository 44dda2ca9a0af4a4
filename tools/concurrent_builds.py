#!/usr/bin/env python3
"""How much of the step is latency that another stream could fill?  Two handles (two streams), each building BASELINE.json
configs[1] over and over from its own thread, against one handle alone: builds per second of the card."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import _dbg  # noqa: E402

N = int(os.environ.get("BUILDS", "12"))
K = int(os.environ.get("K", "31"))
hs = []
for i in range(2):
    g = _dbg.Graph()
    g.synth_reads(1 + i, 50000000, 10000000, 150, 0.01)
    g.build(K)
    hs.append(g)


def loop(g):
    for _ in range(N):
        g.build(K)


t0 = time.perf_counter()
loop(hs[0])
one = (time.perf_counter() - t0) / N
ts = [threading.Thread(target=loop, args=(g,)) for g in hs]
t0 = time.perf_counter()
for t in ts:
    t.start()
for t in ts:
    t.join()
two = (time.perf_counter() - t0) / (2 * N)
print(f"k = {K}: one handle {one * 1e3:.2f} ms per build; two handles at once {two * 1e3:.2f} ms per build of the card "
      f"({one / two:.2f}x the throughput)")
