#!/bin/bash
# usage: tools/dev/mfcc512_static.sh   (CPU only) -- static figures of the headline instance's pass loop: VALU instructions on the
# steady-state path, how many are 64-bit encoded (VOP3 / DPP / SDWA / literal: ~4.2 issue cycles instead of ~2.8), SGPR-spill
# restores (v_readlane), registers. Uses tools/ubench/valu_replay_gen.py's extraction.
cd "$(dirname "$0")/../.." || exit 1
D=/tmp/mfcc512_static; rm -rf $D
python tools/ubench/valu_replay_gen.py $D > /dev/null || exit 1
python3 - $D <<'PY'
import json, sys, re, subprocess, collections
d = sys.argv[1]
info = json.load(open(d + "/valu_replay_info.json"))
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", d + "/valu_replay_base.co"], capture_output=True, text=True).stdout
n4 = n8 = 0; by8 = collections.Counter()
for l in out.splitlines():
    m = re.match(r'\s+(v_\S+)\s.*//\s*([0-9A-F]+):((?:\s[0-9A-F]{8})+)', l)
    if not m: continue
    if len(m.group(3).split()) == 1: n4 += 1
    else: n8 += 1; by8[m.group(1)] += 1
print("VALU on the steady path:", info["valu_in_stream"], " 32-bit:", n4, " 64-bit:", n8, " s_nop:", info["s_nop_in_stream"], " dft stages:", info["dft_stage1_valu"], info["dft_stage2_valu"])
print("model 2.8 x n32 + 4.2 x n64 =", round(2.8 * n4 + 4.2 * n8), "cycles per pass")
print("64-bit forms:", dict(by8.most_common(12)))
asm = open(d + "/lld_mfcc512.s").read()
k = "_ZN8smilehip11lld_mfcc512ILi13ELb1ELb1ELb1ELb0ELi6ELb1EEEvNS_9LldParamsENS_13Fast512TablesE"
i = asm.index(".amdhsa_kernel " + k)
blk = asm[i:asm.index(".end_amdhsa_kernel", i)]
for key in ("next_free_vgpr", "next_free_sgpr", "accum_offset", "private_segment_fixed_size"):
    m = re.search(r"\.amdhsa_" + key + r"\s+(\S+)", blk)
    print(" ", key, m.group(1) if m else "?")
j = asm.index("; Kernel info:", asm.index(k + ":")) if "; Kernel info:" in asm[asm.index(k + ":"):] else -1
m = re.search(r"; ScratchSize: (\d+).*?; Occupancy: (\d+)", asm[asm.index(k + ":"):], re.S)
print("  scratch", m.group(1), "occupancy", m.group(2))
PY
