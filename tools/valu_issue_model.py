#!/usr/bin/env python3
"""Instruction-issue floor of the hot kernels from their DYNAMIC instruction-class counts.

    tools/valu_issue_model.py > profiles/r06_valu_issue_calibration.json

Inputs (all under profiles/): r06_valu_classes.json (tools/ubench/valu_classes.hip: wall time per wave-instruction per SIMD of every
class at 4 waves per SIMD), r06_pmc_c<N>.txt + r06_pmc64_c<N>.txt (rocprofv3 --pmc passes at the bench's batch sizes: SQ_INSTS_VALU and
its classes per launch), r06_valu_replay.json (tools/ubench/valu_replay_gen.py: lld_mfcc512's own VALU stream replayed).

Per kernel:  class_floor_cycles = sum over classes of count x the CHEAPEST measured cost of the class (a 32-bit encoded form with
independent operands) -- a lower bound of the issue time that no schedule of these instructions can beat, loose by what the
64-bit encodings (VOP3, DPP, SDWA, literals: ~4.2 cycles instead of ~2.8), dependent chains and s_nop wait states add;
issue_frac = class_floor_cycles / (1024 SIMDs x the launch's cycles). For lld_mfcc512 the replay gives the TIGHT floor: its own
937-instruction pass stream, its own registers and dependencies, nothing but VALU: 3.74 cycles per instruction at saturation."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
GHZ, N_SIMD, N_XCD = 2.4, 1024, 8


def parse(path):
    d, cur = {}, None
    if not os.path.exists(path):
        return d
    for l in open(path):
        if l.startswith("=="):
            cur = re.sub(r"^(void )?smilehip::", "", l[2:].strip())
            d.setdefault(cur, {})
            continue
        m = re.match(r"\s+(\S+)\s+avg/dispatch\s+([0-9.]+)\s+\(n=(\d+)\)", l)
        if m and cur:
            d[cur][m.group(1)] = float(m.group(2))
    return d


def main():
    ops = json.load(open(os.path.join(P, "r06_valu_classes.json")))["ops"]
    cyc = lambda *names: min(ops[n]["cycles_at_2.4GHz"] for n in names)
    cost = {"ADD_F32": cyc("v_add_f32", "v_sub_f32"), "MUL_F32": cyc("v_mul_f32"), "FMA_F32": cyc("v_fmac_f32", "v_fma_f32"),
            "TRANS_F32": cyc("v_rcp_f32", "v_log_f32", "v_sqrt_f32"), "CVT": cyc("v_cvt_f32_i32"), "INT32": cyc("v_add_u32", "v_lshlrev_b32"),
            "ADD_F64": cyc("v_add_f64"), "MUL_F64": cyc("v_mul_f64"), "FMA_F64": cyc("v_fma_f64"), "TRANS_F64": cyc("v_rcp_f64"),
            "INT64": cyc("v_fma_f32"),            # v_lshl_add_u64 and friends are VOP3: the 64-bit encoded rate
            "other": cyc("v_mov_b32", "v_mov_b32_dpp")}
    out = {"clock_ghz": GHZ, "class_cost_cycles": cost,
           "class_cost_source": "profiles/r06_valu_classes.json (tools/ubench/valu_classes.hip, 4 waves per SIMD; the cheapest form of each class; "
                                "v_cndmask_b32's 23.6 there is an artefact of the VOP2 form's implicit vcc in a tight loop and is not used)",
           "kernels": {}, "cycles_per_valu_inst": {}}
    # the replay of the round's FINAL kernel when it is there (r06_valu_replay_tuned.json), else of the kernel the round started with
    rep_file = "r06_valu_replay_tuned.json" if os.path.exists(os.path.join(P, "r06_valu_replay_tuned.json")) else "r06_valu_replay.json"
    rep = json.load(open(os.path.join(P, rep_file)))["variants"]
    out["replay_source"] = "profiles/" + rep_file
    quad_rep = {}
    qf = os.path.join(P, "r06_stream_replay_quads.json")
    if os.path.exists(qf):
        names = {"is09_quad_valu": "lld_is09_frame_quad", "compare_quad_valu": "lld_compare_frame_quad", "frame20_quad_valu": "lld_gemaps_frame20_quad", "f0_spec_valu": "lld_f0_spec", "f0_cand9_valu": "lld_f0_cand9"}
        for l in open(qf):
            if l.strip().startswith("{"):
                j = json.loads(l)
                if j.get("kernel") in names:
                    quad_rep[names[j["kernel"]]] = j
    for c in (2, 3, 4, 5):
        a, b = parse(os.path.join(P, f"r06_pmc_c{c}.txt")), parse(os.path.join(P, f"r06_pmc64_c{c}.txt"))
        for k, v in a.items():
            if not v.get("SQ_INSTS_VALU") or "GRBM_GUI_ACTIVE" not in v:
                continue
            cycles = v["GRBM_GUI_ACTIVE"] / N_XCD
            if cycles / (GHZ * 1e6) < 0.3:
                continue
            n = v["SQ_INSTS_VALU"]
            w = b.get(k, {})
            cls = {"ADD_F32": v.get("SQ_INSTS_VALU_ADD_F32", 0), "MUL_F32": v.get("SQ_INSTS_VALU_MUL_F32", 0), "FMA_F32": v.get("SQ_INSTS_VALU_FMA_F32", 0),
                   "TRANS_F32": v.get("SQ_INSTS_VALU_TRANS_F32", 0), "CVT": v.get("SQ_INSTS_VALU_CVT", 0), "INT32": v.get("SQ_INSTS_VALU_INT32", 0),
                   "ADD_F64": w.get("SQ_INSTS_VALU_ADD_F64", 0), "MUL_F64": w.get("SQ_INSTS_VALU_MUL_F64", 0), "FMA_F64": w.get("SQ_INSTS_VALU_FMA_F64", 0),
                   "TRANS_F64": w.get("SQ_INSTS_VALU_TRANS_F64", 0), "INT64": w.get("SQ_INSTS_VALU_INT64", 0)}
            cls["other"] = max(0.0, n - sum(cls.values()))
            floor = sum(cls[x] * cost[x] for x in cls)
            name = k.split("(")[0].split("<")[0].strip()
            r = {"config": c, "valu_insts_per_launch": n, "classes": {x: round(y / n, 4) for x, y in cls.items()},
                 "measured_cycles_per_inst": cycles * N_SIMD / n, "class_floor_cycles_per_inst": floor / n,
                 "issue_frac_class_floor": floor / (cycles * N_SIMD), "ms_per_launch_under_counters": cycles / (GHZ * 1e6)}
            if name == "lld_mfcc512":
                cpi = rep["base"]["cycles_at_2.4GHz"]
                r["replay_floor_cycles_per_inst"] = cpi
                r["issue_frac_replay_floor"] = cpi * n / (cycles * N_SIMD)
                r["replay_note"] = ("tools/ubench/valu_replay_gen.py: the pass loop's own 937 VALU instructions (steady-state path) with the compiler's "
                                    "registers, nothing else, 4 waves per SIMD: %.4f ms per 249 500 passes against the launch's measured time" % rep["base"]["ms_per_bench_launch"])
                out["cycles_per_valu_inst"][name] = cpi
            elif name in quad_rep:
                # the frame loop of a sixteen-lanes-per-frame kernel (or of lld_f0_spec) replayed (tools/ubench/run_stream_replay_quads.sh): its own stream's cost
                cpi = quad_rep[name]["cycles_per_valu_per_simd"]
                r["replay_floor_cycles_per_inst"] = cpi
                r["issue_frac_replay_floor"] = cpi * n / (cycles * N_SIMD)
                r["replay_note"] = ("tools/ubench/stream_replay_gen.py: the frame loop's %d vector instructions (every block, file order) alone at the "
                                    "kernel's 3 waves per SIMD; profiles/r06_stream_replay_quads.json" % quad_rep[name]["valu_per_iter"])
                out["cycles_per_valu_inst"].setdefault(name, floor / n)
                out.setdefault("replay_cycles_per_valu_inst", {})[name] = cpi
            else:
                out["cycles_per_valu_inst"].setdefault(name, floor / n)
            out["kernels"].setdefault(f"c{c}:{k}", r)
    out["mfma_beside_valu"] = {
        "question": "does v_mfma_f32_32x32x2_f32 run beside the vector ALU of the same SIMD (VERDICT r05 item 2b)?",
        "answer": "no: the times ADD. In one stream (16 MFMA spread over the %d-instruction pass) each MFMA costs its full ~75 cycles on top of the VALU "
                  "time; wave-specialised (waves 0-3 of a block VALU only, waves 4-7 MFMA only) the two halves take %.2f ms and %.2f ms alone and %.2f ms "
                  "together. A radix-16 stage as a dense 32 x 32 real product (16 MFMA per pass = 1024 matrix-pipe cycles) would replace 138-144 VALU "
                  "instructions (~520 cycles): slower on either schedule." % (rep["base"]["valu_per_pass"], rep["split_valu_only"]["ms_run"], rep["split_mfma_only"]["ms_run"], rep["split_both"]["ms_run"]),
        "ms_per_bench_launch": {k: v["ms_per_bench_launch"] for k, v in rep.items() if not k.startswith("split")},
        "source": "profiles/" + rep_file + " (the round-5 kernel's: r06_valu_replay.json), counters (SQ_VALU_MFMA_BUSY_CYCLES beside SQ_ACTIVE_INST_VALU) profiles/r06_valu_replay_tuned_pmc.txt / r06_valu_replay_pmc.txt"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
