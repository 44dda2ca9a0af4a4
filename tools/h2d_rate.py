import sys, time
sys.path.insert(0, "py-debruijn_amd")
import _dbg
g = _dbg.Graph(); g.synth_reads(1, 50_000_000, 10_000_000, 150, 0.01)
b, o = g.copy_reads()
g2 = _dbg.Graph()
for _ in range(3):
    t = time.perf_counter(); g2.set_reads(b, o); dt = time.perf_counter() - t
    print("set_reads wall ms", round(dt * 1e3, 1), "ms_h2d", round(g2.stats()["ms_h2d"], 1), "GB/s", round(b.nbytes / dt / 1e9, 1))
t = time.perf_counter(); g2.build(31); print("build", round((time.perf_counter() - t) * 1e3, 1))
