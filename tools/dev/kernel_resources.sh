#!/bin/bash
# usage: tools/dev/kernel_resources.sh [file.hip ...]  (default: every kernel file) -> per kernel: VGPRs, scratch bytes per lane, occupancy
# (waves per SIMD), LDS -- compiled with exactly the flags opensmile_amd/csrc/Makefile uses for that file (make -n), plus
# -Rpass-analysis=kernel-resource-usage. EXTRA="..." appends experiment flags. CPU only.
cd "$(dirname "$0")/../../opensmile_amd/csrc" || exit 1
[ $# -eq 0 ] && set -- $(ls lld_*.hip)
for f in "$@"; do
  echo "== $f"
  cmd=$(make -n -B "${f%.hip}.o" 2>/dev/null | grep hipcc | head -1 | sed "s| -o ${f%.hip}.o| -o /tmp/kres.o|")
  [ -z "$cmd" ] && { echo "no rule"; continue; }
  $cmd ${EXTRA:-} -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re
name = None; rec = {}
for line in sys.stdin:
    m = re.search(r"remark: (?:[^ ]+: )?\s*(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|AGPRs|LDS Size \[bytes/block\]|TotalSGPRs): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        name = v; rec[name] = {}
    elif name: rec[name][k.split(" ")[0]] = v
for n, r in rec.items():
    print("%-72s vgpr %4s agpr %3s sgpr %4s scratch %4s occ %s" % (n[:72], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy")))
'
done
