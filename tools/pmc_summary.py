#!/usr/bin/env python3
"""Sum rocprofv3 --pmc csv rows per kernel and counter: tools/pmc_summary.py <dir> [kernel substring]."""
import csv
import glob
import json
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"][:60]
            if len(sys.argv) > 2 and sys.argv[2] not in name:
                continue
            rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[name].add(r["Dispatch_Id"])
out = {k: {"dispatches": len(disp[k]), **{c: v / len(disp[k]) for c, v in sorted(cs.items())}} for k, cs in rows.items()}
print(json.dumps(out, indent=1))
