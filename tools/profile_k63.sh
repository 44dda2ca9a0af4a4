#!/bin/bash
# Kernel stats + instruction-mix / LDS counters of the k = 63 bench (the two-word engine), like profile_round.sh.
set -uo pipefail
TAG=${1:-r02_k63}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --k 63 --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_under_rocprof.log" 2>&1
echo "stats rc=$?"
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --k 63 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"
  python3 tools/pmc_summary.py "$OUT/pmc_$name" > "$OUT/pmc_$name.json"
done
cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
