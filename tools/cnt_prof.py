#!/usr/bin/env python3
"""Where a k_sk_count workgroup spends its time (experiment build: DBG_DEFS=-DDBG_CNT_PROF DBG_OUT=... ; run with DBG_LIB=...)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import _dbg

K = int(os.environ.get("SWEEP_K", "31"))
reads = int(os.environ.get("SWEEP_READS", "10000000"))
g = _dbg.Graph()
CK = int(os.environ.get("COUNT_KERNEL", "2"))
if K <= 31:
    g.set_option("count_kernel", CK)
else:
    g.set_option("wcount_kernel", CK)
g.synth_reads(1, reads * 5, reads, 150, 0.01)
lib = _dbg.load_library()
out = (C.c_ulonglong * 64)()
g.build(K)
lib.dbg_debug_cnt_prof(out, 1)
g.build(K)
lib.dbg_debug_cnt_prof(out, 1)
names = {0: "queries out + top barrier", 1: "clear + stage + barrier", 2: "dedupe + barrier", 3: "quad list + barrier",
         4: "insert", 5: "barrier after insert", 6: "dense list", 7: "barrier after list", 8: "lookups: u0 loads",
         17: "lookups: rest", 9: "reservation (thread 0)", 10: "barrier after lookups", 11: "node write",
         12: "barrier after write"}
main = [0, 1, 2, 3, 4, 5, 6, 7, 8, 17, 9, 10, 11, 12]
if K <= 31 and CK == 2:  # k_sk_count2 (dbg_sk2.h)
    names = {0: "queries out + top barrier", 1: "clear + stage + barrier", 3: "dedupe + quad list + barrier", 4: "insert (+hints, pending list)",
             5: "barrier after insert", 13: "prefetch of the next bucket", 14: "reservation issue", 6: "pending lookups", 7: "dense list",
             9: "reservation wait + range (thread 0)", 10: "barrier after list", 11: "node write", 12: "barrier after write"}
    main = [0, 1, 3, 4, 5, 13, 14, 6, 7, 9, 10, 11, 12]
if K > 31 and CK == 2:  # k_wsk_count2 (dbg_wsk2.h)
    names = {0: "sub-range setup", 13: "top barrier + query atomics + clear", 14: "(first round staged before)", 15: "stage stores", 1: "barrier after stage",
             3: "dedupe + quad list + query flush + barrier", 4: "insert (+hint) + wave sum", 5: "barrier after insert",
             6: "reservation issue + prefetch + dense list", 9: "reservation wait + range (thread 0)", 10: "barrier after list",
             11: "node write (+ next bucket staged)", 12: "deferred lookups"}
    main = [0, 13, 14, 15, 1, 3, 4, 5, 6, 9, 10, 11, 12]
tot = sum(out[i] for i in main)
st = g.stats()
print(f"{out[31]} workgroups, {tot / out[31]:.0f} clocks each; count {st['ms_count']:.2f} ms, {st['n_buckets']} buckets")
for i in main:
    print(f"  {names[i]:28s} {100.0 * out[i] / tot:5.1f} %  = {st['ms_count'] * out[i] / tot:6.2f} ms")
sub = {20: "(since previous subtick)", 13: "u0 find loop / wide: clear", 14: "u0 wave_alloc_n<4> / wide: stage stores", 15: "u0 miss staging / wide: dd clear"}
for i, n in sub.items():
    print(f"    sub {n:28s} {100.0 * out[i] / tot:5.1f} %  = {st['ms_count'] * out[i] / tot:6.2f} ms")
print(f"  events: staged rounds {out[21]}, rounds staged at the top {out[22]}")
print("  insert (or, -DDBG_CNT_PROF=2 and k_wsk_count2: node write + deferred lookups) clocks per wave (share of wave 0):", " ".join(f"{out[32 + w] / max(1, out[32]):.2f}" for w in range(16)))
print(f"  quads per round {out[48] / max(1, out[21] + out[22]):.0f}, records per round {out[49] / max(1, out[21] + out[22]):.0f}")
if out[51]:
    print(f"  -DDBG_CNT_PROF=3: wave 0 waits at the top barrier {st['ms_count'] * out[50] / tot:.2f} ms for the last wave; the last wave was one of "
          + ", ".join(f"{2 * i}-{2 * i + 1}: {100.0 * out[52 + i] / out[51]:.0f} %" for i in range(8)))
