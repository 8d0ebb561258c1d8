#!/usr/bin/env python3
"""VALU replay of the headline kernel's own pass loop: the issue-time floor of ITS instruction mix.

    tools/ubench/valu_replay_gen.py <outdir>          (CPU only: hipcc cross-compiles)

Compiles opensmile_amd/csrc/lld_mfcc512.hip with the Makefile's flags to assembly (+ line tables), takes the pass loop of the bench's
instance -- lld_mfcc512<13, true, true, true, false, 6, true> -- and writes stand-alone gfx950 assembly kernels that execute ONLY the
loop's vector-ALU instructions (the steady-state path: the clamped-index regression block and the clamped prefetch block, which a pass
in the middle of an utterance does not execute, are left out), with the registers the compiler allocated -- so every dependency
between them is the kernel's own -- in a counted loop. No memory, LDS or scalar instruction survives (s_nop does: the wait states
are part of the issue time), so what the kernel measures is the time the SIMDs need to ISSUE the pass's VALU stream, at the
kernel's own occupancy (two blocks of eight waves per CU = 4 waves per SIMD).

Variants (one .s each; tools/ubench/valu_replay.hip runs them):
  base        the VALU stream as it is
  mfma16      + 16 v_mfma_f32_32x32x2_f32 per pass (one radix-16 stage of the pass's four frames as a dense 32x32 real matrix product:
              M = 32 output parts, N = 32 columns = 2 frames x 16, K = 32 -> 8 instructions per 2 frames) spread evenly over the stream,
              on the 16 accumulation registers the kernel's 112 VGPRs leave free at 4 waves per SIMD -- does the matrix pipe run BESIDE
              the vector ALU of the other waves?
  mfma16_nodft2   the same with the second radix-16 stage's vector instructions removed (what the stage would cost on the matrix pipe
              instead) and 16 v_mov_b32 added for the operand exchange
  mfma32_nodft    both stages removed, 32 MFMA per pass, 32 v_mov_b32 added
  nodft2 / nodft  the streams without the stage(s) and without MFMA (how much vector time the stages are)
  split_*     wave-specialised: waves 0-3 of every block (one per SIMD, two per SIMD with the two resident blocks) run the VALU stream, waves 4-7 a loop of 32 MFMA per
              iteration; valu_only / mfma_only = the other half exits at once, both = side by side. both ~ max(valu_only, mfma_only)
              would mean the two pipes serve DIFFERENT waves concurrently
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "opensmile_amd", "csrc", "lld_mfcc512.hip")
KERNEL = "_ZN8smilehip11lld_mfcc512ILi13ELb1ELb1ELb1ELb0ELi6ELb1EEEvNS_9LldParamsENS_13Fast512TablesE"
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Xclang -target-feature "
         "-Xclang -load-store-opt --cuda-device-only -gline-tables-only -S").split()
DFT_LINES = set(range(93, 104)) | set(range(111, 142))   # dft4 / dft16 of lld_mfcc512.hip (innermost inlined location); cmul (105-109) is left
# in: the line tables cannot tell the 4 cmul inside a dft16 from the 15 twiddle products between the stages, which stay on the vector ALU


def kernel_body(asm_path):
    lines = open(asm_path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l) or re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"', l)
        if m:
            files[m.group(1)] = m.group(2)
    return lines[start:end + 1], files


def pass_loop(body):
    """(first, last) line index of the pass loop: the depth-1 loop with the most instructions"""
    heads = [(i, re.match(r"^(\.LBB\d+_\d+):", l).group(1)) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:.*Loop Header: Depth=1", l)]
    best = None
    for i, lab in heads:
        back = [k for k in range(i, len(body)) if re.match(r"\s+s_c?branch\S*\s+" + re.escape(lab) + r"\s*$", body[k])]
        if back and (best is None or back[-1] - i > best[1] - best[0]):
            best = (i, back[-1])
    return best


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/valu_replay"
    os.makedirs(out, exist_ok=True)
    asm = os.path.join(out, "lld_mfcc512.s")
    subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, SRC, "-o", asm], check=True, stderr=subprocess.DEVNULL)
    body, files = kernel_body(asm)
    lo, hi = pass_loop(body)
    loop = body[lo:hi + 1]
    # ---- blocks: label -> lines; the two blocks a steady-state pass skips
    blocks, order, cur = {}, [], "head"
    for l in loop:
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %bb\.(\d+):", l)
        if m:
            cur = m.group(0).split(":")[0]
        if cur not in blocks:
            blocks[cur] = []
            order.append(cur)
        blocks[cur].append(l)
    def n_valu(ls):
        return sum(1 for l in ls if l.strip().startswith("v_"))
    def has(ls, pat):
        return any(re.search(pat, l) for l in ls)
    skipped = []
    for b in order:
        ls = blocks[b]
        # delta_exact: the block with the division sequence (v_div_scale) and ds_bpermute; the clamped prefetch: v_min_i32 chains
        if n_valu(ls) > 60 and (has(ls, r"v_div_scale") or (has(ls, r"v_min_i32") and not has(ls, r"v_fmac"))):
            skipped.append((b, n_valu(ls)))
    skip = {b for b, _ in skipped}
    # ---- the stream: (instruction text, source line, is_dft)
    stream, cur_line, cur_file = [], 0, ""
    seen_transpose_write = False
    n_ds_read_after = 0
    for b in order:
        if b in skip:
            continue
        for l in blocks[b]:
            m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
            if m:
                cur_file, cur_line = files.get(m.group(1), ""), int(m.group(2))
                continue
            t = l.strip()
            if t.startswith("ds_write_b64"):
                seen_transpose_write = True
            if not t or t.startswith(";") or t.startswith("."):
                continue
            op = t.split()[0]
            if op.startswith("v_") or op == "s_nop":
                is_dft = op.startswith("v_") and cur_file.endswith("lld_mfcc512.hip") and cur_line in DFT_LINES
                stream.append((t.split(";")[0].rstrip(), cur_line, is_dft, 2 if (is_dft and seen_transpose_write) else (1 if is_dft else 0)))
    n_v = sum(1 for s in stream if s[0].startswith("v_"))
    n_d1 = sum(1 for s in stream if s[3] == 1)
    n_d2 = sum(1 for s in stream if s[3] == 2)
    used_s = set()
    for s in stream:
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", s[0]):
            if m.group(3):
                used_s.add(int(m.group(3)))
            else:
                used_s.update(range(int(m.group(1)), int(m.group(2)) + 1))
    ctr = next(k for k in range(99, 1, -1) if k not in used_s and k not in (0, 1))
    used_v = [int(x) for s in stream for x in re.findall(r"\bv(\d+)\b", s[0])] + [int(x) for s in stream for m in re.finditer(r"\bv\[(\d+):(\d+)\]", s[0]) for x in m.groups()]
    max_v = max(used_v)
    info = {"pass_loop_lines": len(loop), "valu_in_stream": n_v, "s_nop_in_stream": len(stream) - n_v, "dft_stage1_valu": n_d1, "dft_stage2_valu": n_d2,
            "skipped_blocks": skipped, "max_vgpr": max_v, "counter_sgpr": ctr}

    def emit(name, drop, n_mfma, n_mov, split=None):
        ins = [s[0] for s in stream if s[3] not in drop]
        n_valu_v = sum(1 for t in ins if t.startswith("v_"))
        extra = []
        for q in range(max(n_mfma, n_mov)):              # interleaved
            if q < n_mfma:
                extra.append("v_mfma_f32_32x32x2_f32 a[0:15], v126, v127, a[0:15]")
            if q < n_mov:
                extra.append("v_mov_b32_e32 v125, v124")
        if extra:                                        # spread evenly
            step = len(ins) / float(len(extra))
            merged, k = [], 0
            for i, t in enumerate(ins):
                merged.append(t)
                while k < len(extra) and (k + 0.5) * step <= i + 1:
                    merged.append(extra[k]); k += 1
            merged += extra[k:]
            ins = merged
        with open(os.path.join(out, f"valu_replay_{name}.s"), "w") as f:
            f.write(f'''\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.text
\t.globl\treplay_{name}
\t.p2align\t8
\t.type\treplay_{name},@function
replay_{name}:
\ts_load_dword s{ctr}, s[0:1], 0x0
\tv_readfirstlane_b32 s{ctr - 1}, v0
\ts_lshr_b32 s{ctr - 1}, s{ctr - 1}, 8
\ts_and_b32 s{ctr - 1}, s{ctr - 1}, 1
\ts_waitcnt lgkmcnt(0)
''')
            if split:                                    # waves 4-7 of a block: matrix pipe only; waves 0-3: the VALU stream only (waves go to the SIMDs round robin)
                f.write(f"\ts_cmp_eq_u32 s{ctr - 1}, 1\n\ts_cbranch_scc1 .Lmfma_side\n")
                if split == "mfma_only":
                    f.write("\ts_endpgm\n")
            f.write(".Lloop:\n")
            for t in ins:
                f.write("\t" + t + "\n")
            f.write(f'''\ts_sub_u32 s{ctr}, s{ctr}, 1
\ts_cmp_lg_u32 s{ctr}, 0
\ts_cbranch_scc1 .Lloop
\ts_endpgm
''')
            if split:
                f.write(".Lmfma_side:\n")
                if split == "valu_only":
                    f.write("\ts_endpgm\n")
                f.write(".Lmloop:\n")
                for q in range(32):
                    f.write("\tv_mfma_f32_32x32x2_f32 a[0:15], v126, v127, a[0:15]\n")
                f.write(f"\ts_sub_u32 s{ctr}, s{ctr}, 1\n\ts_cmp_lg_u32 s{ctr}, 0\n\ts_cbranch_scc1 .Lmloop\n\ts_endpgm\n")
            f.write(f'''.Lfunc_end:
\t.size\treplay_{name}, .Lfunc_end-replay_{name}
\t.rodata
\t.p2align\t6
\t.amdhsa_kernel replay_{name}
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_next_free_vgpr 128
\t\t.amdhsa_next_free_sgpr 100
\t\t.amdhsa_accum_offset 112
\t\t.amdhsa_group_segment_fixed_size 0
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel
\t.amdgpu_metadata
---
amdhsa.version: [1, 2]
amdhsa.kernels:
  - .name: replay_{name}
    .symbol: replay_{name}.kd
    .kernarg_segment_size: 8
    .group_segment_fixed_size: 0
    .private_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .wavefront_size: 64
    .sgpr_count: 106
    .vgpr_count: 128
    .agpr_count: 16
    .max_flat_workgroup_size: 512
    .args:
      - {{.size: 4, .offset: 0, .value_kind: by_value}}
...
\t.end_amdgpu_metadata
''')
        co = os.path.join(out, f"valu_replay_{name}.co")
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                        os.path.join(out, f"valu_replay_{name}.s"), "-o", co + ".o"], check=True)
        subprocess.run(["/opt/rocm/lib/llvm/bin/ld.lld", "-shared", co + ".o", "-o", co], check=True)
        os.remove(co + ".o")
        info[name] = {"valu": n_valu_v + n_mov, "mfma": n_mfma}

    emit("base", set(), 0, 0)
    emit("mfma16", set(), 16, 0)
    emit("mfma32", set(), 32, 0)
    emit("nodft2", {2}, 0, 0)
    emit("nodft", {1, 2}, 0, 0)
    emit("mfma16_nodft2", {2}, 16, 16)
    emit("mfma32_nodft", {1, 2}, 32, 32)
    # wave-specialised: waves 0-3 of a block (one per SIMD) run the VALU stream, waves 4-7 32 MFMA per iteration -- alone and together
    emit("split_valu_only", set(), 0, 0, split="valu_only")
    emit("split_mfma_only", set(), 0, 0, split="mfma_only")
    emit("split_both", set(), 0, 0, split="both")
    import json
    json.dump(info, open(os.path.join(out, "valu_replay_info.json"), "w"), indent=1)
    print(json.dumps(info, indent=1))


if __name__ == "__main__":
    main()
