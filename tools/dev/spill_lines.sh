#!/bin/bash
# usage: tools/dev/spill_lines.sh <file.hip> <mangled kernel prefix>  -> scratch (spill) instructions of one kernel by source line
# (compiled with the Makefile's flags for that file + -gline-tables-only). CPU only.
cd "$(dirname "$0")/../../opensmile_amd/csrc" || exit 1
cmd=$(make -n -B "${1%.hip}.o" 2>/dev/null | grep hipcc | head -1 | sed "s| -c | --cuda-device-only -gline-tables-only -S |; s| -o ${1%.hip}.o| -o /tmp/_sl.s|")
$cmd 2>/dev/null
awk -v k="$2" 'index($0, k) == 1 && /:/ {p=1} p{print} /s_endpgm/{if(p){exit}}' /tmp/_sl.s > /tmp/_slk.s
python3 - <<'PY'
import re, collections
files = {}
for l in open('/tmp/_sl.s'):
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m: files[m.group(1)] = m.group(2)
cur = None; cnt = collections.Counter(); valu = 0
for l in open('/tmp/_slk.s'):
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(m.group(1), m.group(1)), int(m.group(2))); continue
    t = l.strip().split()
    if not t: continue
    if t[0].startswith('scratch_'): cnt[cur] += 1
    if t[0].startswith('v_'): valu += 1
print("VALU instructions (static):", valu, " scratch instructions:", sum(cnt.values()))
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1])[:25]: print("  %-28s line %4d: %d" % (k[0], k[1], v))
PY
