#!/bin/bash
# Builds the C-ABI shared library for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${DBG_OUT:-../libdbg_hip.so}   # DBG_OUT / DBG_DEFS: experimental variants next to the product build
"$HIPCC" -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wall -Wno-unused-function ${DBG_DEFS:-} \
    -o "$OUT" dbg_hip.hip
echo "built $OUT"
