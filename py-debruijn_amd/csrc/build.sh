#!/bin/bash
# Builds the C-ABI shared library for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${DBG_OUT:-../libdbg_hip.so}   # DBG_OUT / DBG_DEFS: experimental variants next to the product build
# -amdgpu-atomic-optimizer-strategy=None: the optimiser turns "one lane adds to a global cursor" into a wave-aggregated add
# followed at once by a broadcast of the result -- a wait for the memory-side atomic (1-2 us) right where it is issued.
# The kernels aggregate by hand where it matters and want the reservation's answer as late as possible.
"$HIPCC" -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None ${DBG_DEFS:-} \
    -o "$OUT" dbg_hip.hip
echo "built $OUT"
