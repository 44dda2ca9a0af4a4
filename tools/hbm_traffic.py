#!/usr/bin/env python3
"""profiles/rNN_hbm_traffic_pmc.json from the FETCH_SIZE and WRITE_SIZE passes of tools/profile_round.sh.

Per kernel and launch: FETCH_SIZE doubled (MI355X_MICROARCH.md: on gfx950 it tallies 128-byte requests at 64 bytes),
WRITE_SIZE as read, both in GB; plus the hash of the kernel sources the numbers were measured on -- bench.py reports
`roofline.traffic` only while the sources still have that hash.
usage: tools/hbm_traffic.py <pmc_FETCH_SIZE.json> <pmc_WRITE_SIZE.json> > profiles/r02_hbm_traffic_pmc.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # kernel_source_hash

fetch = json.load(open(sys.argv[1]))
write = json.load(open(sys.argv[2]))
rows = []
for name in sorted(set(fetch) | set(write)):
    f, w = fetch.get(name, {}), write.get(name, {})
    fk, wk = f.get("FETCH_SIZE", 0.0), w.get("WRITE_SIZE", 0.0)  # KB per launch (mean over the sampled launches)
    rows.append({"kernel": name, "launches_sampled": int(f.get("dispatches", w.get("dispatches", 0))),
                 "FETCH_SIZE_KB_per_launch": fk, "WRITE_SIZE_KB_per_launch": wk,
                 "hbm_read_GB_corrected_x2": 2 * fk * 1024 / 1e9, "hbm_write_GB": wk * 1024 / 1e9})
rows.sort(key=lambda r: -(r["hbm_read_GB_corrected_x2"] + r["hbm_write_GB"]))
print(json.dumps({"kernel_source_hash": bench.kernel_source_hash(), "command": "python3 bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-extras under rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes)",
                  "kernels": rows}, indent=1))
