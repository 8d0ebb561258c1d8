#!/bin/bash
# usage: tools/dev/valu_lines.sh <file.hip> <mangled kernel prefix> [bucket]  -> static SALU (scalar ALU, branches included; not s_waitcnt / s_nop) instruction counts of one kernel by source
# line, summed over buckets of `bucket` lines (default 10) of the INNERMOST inlined location (Makefile flags + -gline-tables-only). CPU only.
cd "$(dirname "$0")/../../opensmile_amd/csrc" || exit 1
cmd=$(make -n -B "${1%.hip}.o" 2>/dev/null | grep hipcc | head -1 | sed "s| -c | --cuda-device-only -gline-tables-only -S |; s| -o ${1%.hip}.o| -o /tmp/_vl.s|")
$cmd 2>/dev/null
awk -v k="$2" 'index($0, k) == 1 && /:/ {p=1} p{print} /s_endpgm/{if(p){exit}}' /tmp/_vl.s > /tmp/_vlk.s
python3 - "${3:-10}" <<'PY'
import re, collections, sys
B = int(sys.argv[1])
files = {}
for l in open('/tmp/_vl.s'):
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m: files[m.group(1)] = m.group(2)
cur = None; cnt = collections.Counter(); tot = 0
for l in open('/tmp/_vlk.s'):
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(m.group(1), m.group(1)), int(m.group(2)) // B * B); continue
    t = l.strip().split()
    if t and t[0].startswith("s_") and not t[0].startswith("s_waitcnt") and not t[0].startswith("s_nop"): cnt[cur] += 1; tot += 1
print("SALU instructions (static):", tot)
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:30]: print("  %-28s lines %4d..%4d: %5d  %4.1f %%" % (k[0], k[1], k[1] + B - 1, v, 100.0 * v / tot))
PY
