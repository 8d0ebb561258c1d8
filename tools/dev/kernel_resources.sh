#!/bin/bash
# usage: tools/dev/kernel_resources.sh file.hip ...  -> per kernel: VGPRs, scratch bytes per lane, occupancy (waves per SIMD), LDS
cd "$(dirname "$0")/../../opensmile_amd/csrc" || exit 1
for f in "$@"; do
  echo "== $f"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kres.o 2>&1 | python3 -c '
import sys, re
name = None; rec = {}
for line in sys.stdin:
    m = re.search(r"remark: (?:[^ ]+: )?\s*(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|AGPRs): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        name = v; rec[name] = {}
    elif name: rec[name][k.split(" ")[0]] = v
for n, r in rec.items():
    print("%-60s vgpr %4s agpr %3s scratch %4s occ %s" % (n[:60], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy")))
'
done
