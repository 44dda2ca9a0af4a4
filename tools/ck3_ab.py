import sys
sys.path.insert(0, "py-debruijn_amd")
import torch; torch.zeros(1, device="cuda")
import _dbg
g = _dbg.Graph(); g.synth_reads(1, 50000000, 10000000, 150, 0.01)
for ck in (2, 3, 2, 3):
    g.set_option("count_kernel", ck)
    g.build(31); g.build(31)
    st = g.stats(); print("k=31 count_kernel", ck, "count ms", round(st["ms_count"], 3), "build", round(st["ms_build_total"], 3), g.sizes()["n_nodes"], st["n_queries"], flush=True)
g.set_option("count_kernel", 2)
g2 = _dbg.Graph(); g2.synth_reads(1, 50000000, 10000000, 150, 0.0)
for ck in (2, 3):
    g2.set_option("count_kernel", ck); g2.build(31); g2.build(31)
    st = g2.stats(); print("error-free count_kernel", ck, "count ms", round(st["ms_count"], 3), "build", round(st["ms_build_total"], 3), flush=True)
