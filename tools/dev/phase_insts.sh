#!/bin/bash
# usage: tools/dev/phase_insts.sh <file.hip> <mangled-name prefix> [MACRO] [extra hipcc flags]: static instruction counts of one kernel between
# the phase markers of its body (the IPHASE / CPHASE ... macros compiled as assembler comments; the loops are unrolled, so
# static = dynamic per pass up to the predicated regions). CPU only: what a change does to the instruction count before a GPU run.
F=$1; K=$2; M=${3:-IPHASE}; shift; shift; [ $# -gt 0 ] && shift
cd "$(dirname "$0")/../../opensmile_amd/csrc" || exit 1
sed "s|^#define $M(i)\$|#define $M(i) asm volatile(\"; PHASEMARK \" #i)|" $F > _tmp_mark.hip
# (a marker macro defined in a header the file includes behind an #ifndef: the forced include defines it first)
printf '#define %s(i) asm volatile("; PHASEMARK " #i)\n' "$M" > _tmp_mark_def.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --cuda-device-only -S -include _tmp_mark_def.h "$@" -o /tmp/_mark.s _tmp_mark.hip 2>/dev/null
rm -f _tmp_mark.hip _tmp_mark_def.h
awk -v k="$K" 'index($0, k) == 1 && /:/ {p=1} p{print} /s_endpgm/{if(p){exit}}' /tmp/_mark.s > /tmp/_mark_k.s
python3 - <<'PY'
import re
ph = 'prologue'; cnt = {}; order = []
for l in open('/tmp/_mark_k.s'):
    m = re.search(r'; PHASEMARK (\d+)', l)
    if m:
        ph = 'after mark ' + m.group(1); continue
    t = l.strip().split()
    if not t or t[0].startswith(';') or t[0].startswith('.') or t[0].endswith(':'): continue
    op = t[0]
    c = cnt.setdefault(ph, dict(valu=0, f64=0, pk=0, mov=0, dpp=0, salu=0, branch=0, ds=0, vmem=0, flat=0, scratch=0, nop=0))
    if ph not in order: order.append(ph)
    if op.startswith('v_'):
        c['valu'] += 1
        if 'f64' in op: c['f64'] += 1
        if op.startswith('v_pk_'): c['pk'] += 1
        if op.startswith('v_mov') or op.startswith('v_accvgpr'): c['mov'] += 1
        if 'dpp' in op or 'dpp' in l: c['dpp'] += 1
    elif op.startswith('s_cbranch') or op == 's_branch': c['branch'] += 1
    elif op == 's_nop': c['nop'] += 1
    elif op.startswith('s_'): c['salu'] += 1
    elif op.startswith('ds_'): c['ds'] += 1
    elif op.startswith('scratch_'): c['scratch'] += 1
    elif op.startswith('flat_'): c['flat'] += 1
    elif op.startswith('global_') or op.startswith('buffer_'): c['vmem'] += 1
tot = {}
for p in order:
    print('%-16s' % p, ' '.join('%s %4d' % kv for kv in cnt[p].items()))
    if p != 'prologue':
        for k, v in cnt[p].items(): tot[k] = tot.get(k, 0) + v
print('%-16s' % 'per pass', ' '.join('%s %4d' % kv for kv in tot.items()))
PY
