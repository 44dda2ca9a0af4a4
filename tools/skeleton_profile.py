import sys, time, math
sys.path.insert(0, "py-debruijn_amd")
import numpy as np, torch; torch.zeros(1, device="cuda")
import _dbg, part_traversal as pt
g = _dbg.Graph(); g.synth_reads(1, 50000000, 10000000, 150, 0.01)
g.build_multipass(31, 4)
t = pt.PartTraversal(g, 31)
t.prune(2); t.pull_out_reads(); t.remove_tips()
orig = pt.rank_skeleton
def timed(e, k, keep=False):
    T = {}
    def lap(name, t0):
        torch.cuda.synchronize(); T[name] = T.get(name, 0) + round((time.perf_counter() - t0) * 1e3, 2)
    n = e["gid"].numel(); print("entries", n)
    t0 = time.perf_counter(); o = torch.argsort(e["gid"]); lap("argsort", t0)
    t0 = time.perf_counter(); e = {c: x[o] for c, x in e.items()}; lap("permute 8 cols", t0)
    t0 = time.perf_counter()
    remote = e["kind"] == pt.K_REMOTE
    want = torch.where(remote, e["next"], e["gid"])
    j = torch.searchsorted(e["gid"], want).clamp_(max=n - 1); lap("searchsorted", t0)
    t0 = time.perf_counter(); ok = bool((e["gid"][j] == want).all()); lap("check", t0)
    t0 = time.perf_counter()
    go_on = remote & (e["kind"][j] != pt.K_PULLED)
    hops = e["hops"] + go_on.to(torch.int64)
    score = e["score"] + torch.where(go_on, e["exit"], torch.zeros_like(e["exit"]))
    dead = e["kind"] == pt.K_CYCLE
    done = ~go_on
    jump = torch.where(go_on, j, torch.arange(n, device=e["gid"].device))
    act = torch.nonzero(~done).reshape(-1); lap("setup", t0)
    print("active", act.numel())
    rounds = 0
    t0 = time.perf_counter()
    for _ in range(max(1, math.ceil(math.log2(n + 1))) + 1):
        if act.numel() == 0: break
        tj = jump[act]
        h2, s2, d2, dn, j2 = hops[act] + hops[tj], score[act] + score[tj], dead[act] | dead[tj], done[tj], jump[tj]
        hops[act], score[act], dead[act], done[act], jump[act] = h2, s2, d2, dn, j2
        act = act[~dn]; rounds += 1
    lap("jumping", t0); print("rounds", rounds)
    t0 = time.perf_counter()
    emit = (e["start"] == 1) & (e["kind"] != pt.K_PULLED) & done & ~dead
    em = torch.nonzero(emit).reshape(-1)
    em = em[torch.argsort(e["stamp"][em])]
    out = {"stamp": e["stamp"][em].cpu().numpy().astype(np.uint64), "length": (hops[em] + k).cpu().numpy(), "score": score[em].cpu().numpy()}
    lap("emit + D2H", t0); print(T)
    return out, None
pt.rank_skeleton = timed
for rep in range(2):
    t0 = time.perf_counter(); idx = t.walk_index(); torch.cuda.synchronize(); print("walk ms", round((time.perf_counter() - t0) * 1e3, 1), idx["stamp"].size)
