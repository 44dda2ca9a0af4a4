#!/bin/bash
# kernel-trace stats of one rank of the sharded step (tools/shard_mp_profile.py: sharded_build + sharded_build_multipass on
# 10 M reads over nccl with one rank), GPU box, through gpurun:  tools/profile_shard_step.sh <tag>
set -uo pipefail
TAG=${1:-r03_shard}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 LOCAL_RANK=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/shard_mp_profile.py > "$OUT/under_rocprof.log" 2>&1
echo "stats rc=$?"
cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
