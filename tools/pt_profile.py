import sys, time, cProfile, pstats
sys.path.insert(0, "py-debruijn_amd")
import torch; torch.zeros(1, device="cuda")
import _dbg, part_traversal
g = _dbg.Graph(); g.synth_reads(1, 50000000, 10000000, 150, 0.01)
for rep in range(2):
    g.build_multipass(31, 4)
    t = part_traversal.PartTraversal(g, 31)
    pr = cProfile.Profile() if rep else None
    if pr: pr.enable()
    for name, fn in (("prune", lambda: t.prune(2)), ("pull", t.pull_out_reads), ("tips", t.remove_tips), ("walk", t.walk_index)):
        t0 = time.perf_counter(); fn(); print(name, round((time.perf_counter() - t0) * 1e3, 1), end="; ")
    print()
    if pr:
        pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(28)
