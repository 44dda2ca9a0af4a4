#!/bin/bash
# Instruction counts of k_sk_count per phase: the kernel cut short after phase N ("phase_limit", ablation) under
# rocprofv3 --pmc; differences between consecutive runs are the phases' own instructions.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/phase_pmc
mkdir -p "$OUT"
for pl in 1 2 3 4 5 0; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pl$pl" -- python3 tools/sweep.py "[{\"phase_limit\": $pl}]" > "$OUT/pl$pl.log" 2>&1
  echo "phase_limit $pl rc=$?"
  python3 tools/pmc_summary.py "$OUT/pl$pl" > "$OUT/pl$pl.json"
done
python3 - <<'PY'
import json
prev = None
names = {1: "clear + stage", 2: "+ dedupe + insert", 3: "+ dense list", 4: "+ lookups", 5: "+ reservation", 0: "+ write + queries (all)"}
for pl in (1, 2, 3, 4, 5, 0):
    d = json.load(open(f"gpurun_out/phase_pmc/pl{pl}.json"))
    v = [x for k, x in d.items() if "k_sk_count" in k][0]
    cur = {c: v[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS")}
    delta = {c: (cur[c] - (prev[c] if prev else 0)) / 1e9 for c in cur}
    print(f"{names[pl]:28s} VALU {cur['SQ_INSTS_VALU']/1e9:6.3f}e9 (+{delta['SQ_INSTS_VALU']:.3f})  SALU {cur['SQ_INSTS_SALU']/1e9:6.3f}e9 (+{delta['SQ_INSTS_SALU']:.3f})  LDS {cur['SQ_INSTS_LDS']/1e9:6.3f}e9 (+{delta['SQ_INSTS_LDS']:.3f})")
    prev = cur
PY
