#!/bin/bash
# kernel-trace stats of the default bench command + the plain bench line (GPU box, through gpurun)
set -uo pipefail
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench_under_rocprof.log" 2>&1
echo "stats rc=$?"
cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
timeout -k 10 500 python3 bench.py > "$OUT/bench_line.log" 2>&1
echo "bench rc=$?"
