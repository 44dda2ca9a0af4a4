#!/bin/bash
# Collects the evidence bench.py's roofline block cites (run on the GPU box through gpurun):
#   kernel-trace stats of the default bench command, then separate --pmc passes for HBM traffic and LDS counters.
# Output: gpurun_out/prof_<tag>/...; tools/pmc_summary.py turns the counter csv files into per-kernel json.
set -uo pipefail
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_under_rocprof.log" 2>&1
echo "stats rc=$?"
for pass in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  name=$(echo "$pass" | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/pmc_$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/pmc_$name.log" 2>&1
  echo "pmc $name rc=$?"
  python3 tools/pmc_summary.py "$OUT/pmc_$name" > "$OUT/pmc_$name.json"
done
cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
python3 tools/hbm_traffic.py "$OUT/pmc_FETCH_SIZE.json" "$OUT/pmc_WRITE_SIZE.json" > "$OUT/hbm_traffic_pmc.json"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_line.log" 2>&1
echo "bench rc=$?"
