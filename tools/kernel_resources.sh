#!/bin/bash
# Resource usage (VGPR / SGPR / spills / LDS / scratch) of every kernel in the built library, and optionally the ISA
# statistics of one of them:  tools/kernel_resources.sh [name-substring]
set -euo pipefail
LIB=${LIB:-$(dirname "$0")/../py-debruijn_amd/libdbg_hip.so}
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$LIB" $T/stripped.so 2>/dev/null || $LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$LIB"
$LLVM/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co
$LLVM/llvm-readelf --notes $T/k.co > $T/notes.txt
python3 - "$T/notes.txt" "${1:-}" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
for blk in txt.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name: continue
    n = name.group(1)
    if pat and pat not in n: continue
    g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [0, "?"])[1]
    print(f"{n[:110]:110s} vgpr {g('vgpr_count'):>3} sgpr {g('sgpr_count'):>3} sgpr_spill {g('sgpr_spill_count'):>3} vgpr_spill {g('vgpr_spill_count'):>3} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size'):>4}")
PY
if [ -n "${1:-}" ]; then
  $LLVM/llvm-objdump -d $T/k.co > $T/dis.txt
  python3 - "$T/dis.txt" "$1" <<'PY'
import re, sys, collections
txt = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r"^[0-9a-f]+ <([^>]+)>:\n(.*?)(?=^\n|\Z)", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name: continue
    ops = [l.split()[0] for l in body.splitlines() if l.strip() and not l.strip().startswith("//")]
    c = collections.Counter()
    for o in ops:
        c["total"] += 1
        if o.startswith("v_readlane") or o.startswith("v_writelane"): c["lane_spill"] += 1
        if o.startswith("s_cbranch") or o.startswith("s_branch"): c["branch"] += 1
        if o.startswith("s_waitcnt"): c["waitcnt"] += 1
        if o.startswith("ds_"): c["lds"] += 1
        if o.startswith("v_"): c["vector"] += 1
        elif o.startswith("s_"): c["scalar"] += 1
        if o.startswith("global_") or o.startswith("buffer_") or o.startswith("flat_"): c["vmem"] += 1
        if o.startswith("scratch_"): c["scratch"] += 1
    print(name[:100], dict(c))
PY
fi
rm -rf $T
