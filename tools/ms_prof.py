#!/usr/bin/env python3
"""Where a multisplit scatter workgroup spends its time (experiment build: DBG_DEFS=-DDBG_MS_PROF, DBG_LIB=...)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "py-debruijn_amd"))
import _dbg

g = _dbg.Graph()
g.synth_reads(1, 40_000_000, 8_000_000, 150, 0.01)
lib = _dbg.load_library()
out = (C.c_ulonglong * 8)()
g.build(31)
lib.dbg_debug_ms_prof(out, 1)
g.build(31)
lib.dbg_debug_ms_prof(out, 1)
names = ["locate+cursors", "zero hist", "load+rank", "scan", "lds scatter", "write out", "cursor update"]
tot = sum(out[i] for i in range(7))
print(f"{out[7]} workgroups, {tot / out[7]:.0f} clocks each; partition {g.stats()['ms_partition']:.2f} ms")
for i, n in enumerate(names):
    print(f"  {n:16s} {100.0 * out[i] / tot:5.1f} %  {out[i] / out[7]:9.0f} clocks per workgroup")
