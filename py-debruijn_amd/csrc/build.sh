#!/bin/bash
# Builds the C-ABI shared library for gfx950 (cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -Wall -Wno-unused-function \
    -o ../libdbg_hip.so dbg_hip.hip
echo "built $(cd .. && pwd)/libdbg_hip.so"
