#!/usr/bin/env python3
"""Replay of a stretch of ANY kernel's compiler output as a stand-alone gfx950 kernel (the method of valu_replay_gen.py, which is
specific to lld_mfcc512, made general): what does the stretch cost when only its vector-ALU instructions are issued -- and when its
scalar instructions, scalar loads and their waits come back?

    tools/ubench/stream_replay_gen.py <file stem, e.g. f0> <mangled kernel name> <first label> <last label> <outdir> <name>
                                      [--block N] [--lds BYTES] [--vgprs N] [--valu-only 1] [--skip .LBBa,.LBBb] [--init 'asm line' ...]

Compiles opensmile_amd/csrc/lld_<stem>.hip with the Makefile's flags to assembly, takes the kernel's lines from <first label> up to
(not including) <last label> and writes two code objects (registers as the compiler allocated them: every dependency is the kernel's own):
  <name>_valu   only the v_* instructions and s_nop, in a counted loop
  <name>_scal   + the scalar ALU instructions, scalar loads and s_waitcnt lgkmcnt (vector memory, LDS, branches, barriers and
                anything that writes EXEC stay out); --init lines run once per iteration before the stretch (set up the scalar
                registers the stretch's addresses are built from; {PTR} = the pointer argument (1 MB of zeros), {CTR} = the iteration
                count down, {WG} = the workgroup index: registers the stretch leaves alone)
tools/ubench/stream_replay_run.hip runs them at a given grid / block and prints cycles per vector instruction per SIMD."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "opensmile_amd", "csrc")


def main():
    a = sys.argv[1:]
    stem, kernel, first, last, out, name = a[:6]
    opt = {"--block": "64", "--lds": "0", "--vgprs": "128"}
    inits = []
    i = 6
    while i < len(a):
        if a[i] == "--init":
            inits.append(a[i + 1])
        else:
            opt[a[i]] = a[i + 1]
        i += 2
    out = os.path.abspath(out)
    os.makedirs(out, exist_ok=True)
    cmd = subprocess.run(["make", "-C", CSRC, "-n", "-B", f"lld_{stem}.o"], capture_output=True, text=True).stdout
    cmd = next(l for l in cmd.split("\n") if "hipcc" in l and " -c " in l)
    asm = os.path.join(out, f"lld_{stem}.s")
    cmd = cmd.replace(f" -c lld_{stem}.hip -o lld_{stem}.o", f" -S --cuda-device-only lld_{stem}.hip -o {asm}")
    if not os.path.exists(asm) or os.environ.get("REGEN"):
        subprocess.run(cmd, shell=True, cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    k0 = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    k1 = next(i for i in range(k0, len(lines)) if "s_endpgm" in lines[i])
    body = lines[k0:k1 + 1]
    i0 = next(i for i, l in enumerate(body) if l.startswith(first + ":"))
    i1 = next(i for i, l in enumerate(body) if l.startswith(last + ":")) if last != "end" else len(body)
    valu, scal = [], []
    skip = set(x for x in opt.get("--skip", "").split(",") if x)      # labels of blocks the steady state does not execute
    skipping = False
    for l in body[i0:i1]:
        ml = re.match(r"^(\.LBB\d+_\d+):", l)
        if ml:
            skipping = ml.group(1) in skip
        if skipping:
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if re.search(r"\bexec\b", t) or op.startswith("v_cmpx"):
            continue
        if op.startswith("v_") and not op.startswith("v_mfma"):
            valu.append(t); scal.append(t)
        elif op == "s_nop":
            valu.append(t); scal.append(t)
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            scal.append(t)
        elif op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", t)
            if m:
                scal.append(f"s_waitcnt lgkmcnt({m.group(1)})")
        elif op.startswith("s_") and not re.match(r"s_(c?branch|barrier|endpgm|setpc|swappc|getpc|call|sleep|sethalt|trap|setprio|memtime|memrealtime|dcache)", op):
            scal.append(t)
    # housekeeping registers the stretch does not touch: the pointer argument (a pair), the iteration counter, the workgroup index
    used = set()
    for t in (valu if "--valu-only" in opt else scal) + inits:
        for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", t):
            if m.group(3):
                used.add(int(m.group(3)))
            else:
                used.update(range(int(m.group(1)), int(m.group(2)) + 1))
    free = [k for k in range(100, 5, -1) if k not in used]
    ptr = next(k for k in free if k % 2 == 0 and k + 1 not in used)
    ctr = next(k for k in free if k not in (ptr, ptr + 1))
    wg = next(k for k in free if k not in (ptr, ptr + 1, ctr))
    sub = lambda t: t.replace("{PTR}", f"s[{ptr}:{ptr + 1}]").replace("{CTR}", f"s{ctr}").replace("{WG}", f"s{wg}")
    inits = [sub(t) for t in inits]
    n_valu = sum(1 for t in valu if t.startswith("v_"))
    info = {"kernel": kernel, "first": first, "last": last, "valu": n_valu, "scalar_alu": sum(1 for t in scal if t.startswith("s_") and not t.startswith("s_load") and not t.startswith("s_waitcnt") and not t.startswith("s_nop")),
            "smem": sum(1 for t in scal if t.startswith("s_load")), "waits": sum(1 for t in scal if t.startswith("s_waitcnt")),
            "block": int(opt["--block"]), "lds": int(opt["--lds"]), "vgprs": int(opt["--vgprs"])}
    variants = (("valu", valu),) if "--valu-only" in opt else (("valu", valu), ("scal", scal))
    for variant, ins in variants:
        kn = f"{name}_{variant}"
        with open(os.path.join(out, kn + ".s"), "w") as f:
            f.write(f'''\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
\t.text
\t.globl\t{kn}
\t.p2align\t8
\t.type\t{kn},@function
{kn}:
\ts_mov_b32 s{wg}, s2
\ts_load_dword s{ctr}, s[0:1], 0x0
\ts_load_dwordx2 s[{ptr}:{ptr + 1}], s[0:1], 0x8
\ts_waitcnt lgkmcnt(0)
.Lloop_{kn}:
''')
            if variant == "scal":
                for t in inits:
                    f.write("\t" + t + "\n")
            for t in ins:
                f.write("\t" + t + "\n")
            f.write(f'''\ts_sub_u32 s{ctr}, s{ctr}, 1
\ts_cmp_lg_u32 s{ctr}, 0
\ts_cbranch_scc1 .Lloop_{kn}
\ts_endpgm
.Lfunc_end_{kn}:
\t.size\t{kn}, .Lfunc_end_{kn}-{kn}
\t.rodata
\t.p2align\t6
\t.amdhsa_kernel {kn}
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_next_free_vgpr {opt["--vgprs"]}
\t\t.amdhsa_next_free_sgpr 100
\t\t.amdhsa_accum_offset {opt["--vgprs"]}
\t\t.amdhsa_group_segment_fixed_size {opt["--lds"]}
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel
\t.amdgpu_metadata
---
amdhsa.version: [1, 2]
amdhsa.kernels:
  - .name: {kn}
    .symbol: {kn}.kd
    .kernarg_segment_size: 16
    .group_segment_fixed_size: {opt["--lds"]}
    .private_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .wavefront_size: 64
    .sgpr_count: 106
    .vgpr_count: {opt["--vgprs"]}
    .agpr_count: 0
    .max_flat_workgroup_size: {opt["--block"]}
    .args:
      - {{.size: 4, .offset: 0, .value_kind: by_value}}
      - {{.size: 8, .offset: 8, .value_kind: global_buffer, .address_space: global}}
...
\t.end_amdgpu_metadata
''')
        co = os.path.join(out, kn + ".co")
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c",
                        os.path.join(out, kn + ".s"), "-o", co + ".o"], check=True)
        subprocess.run(["/opt/rocm/lib/llvm/bin/ld.lld", "-shared", co + ".o", "-o", co], check=True)
        os.remove(co + ".o")
    json.dump(info, open(os.path.join(out, name + "_info.json"), "w"), indent=1)
    print(json.dumps(info))


if __name__ == "__main__":
    main()
