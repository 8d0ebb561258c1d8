#!/usr/bin/env python3
"""Counter summary (tools/pmc_summary.py's text: per-kernel averages per dispatch of the separate rocprofv3 --pmc passes) ->
the JSON bench.py reads for `roofline.bound_by_counters` / `valu_issue_frac` (profiles/r05_pmc_c<N>.json).

usage: pmc_counters_json.py profiles/r05_pmc_c3.txt [min_ms] > profiles/r05_pmc_c3.json

Per kernel (each alone on the device: counter collection serialises dispatches):
  ms                      GRBM_GUI_ACTIVE / 8 XCDs / 2.4 GHz
  valu_issue_frac         SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / cycles   (a wave64 VALU instruction holds its SIMD's issue port 4 cycles;
                          FP64 and transcendental instructions hold it longer, so this is a LOWER bound on the port's use)
  valu_busy_frac          SQ_ACTIVE_INST_VALU x 4 / 1024 / cycles          (the counter is in quad-cycles: the port's measured busy share)
  lds_pipe_frac           SQ_LDS_IDX_ACTIVE / 256 CUs / cycles
  lds_bank_conflict_frac  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  wait_inst_frac          SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                (share of resident wave time spent waiting on an outstanding instruction)
  waves_per_simd          SQ_WAVE_CYCLES x 4 / 1024 / cycles               (average resident waves)
  hbm_bytes, hbm_frac     (2 x FETCH_SIZE + WRITE_SIZE) KB per dispatch (gfx950 correction of MI355X_MICROARCH.md) against 8 TB/s
  scratch / spill         not a counter: tools/dev/kernel_resources.sh (profiles/r05_kernel_resources.txt)
  valu_issue_floor_frac   (round 6) sum over the instruction classes the SQ counts (ADD / MUL / FMA / TRANS F32 and F64, CVT, INT32 / INT64,
                          the rest) of count x the class's measured issue cost at 4 waves per SIMD (profiles/r06_valu_classes.json,
                          cheapest form) / 1024 SIMDs / cycles: the share of the launch that ISSUING its own instructions needs at least.
                          By construction <= 1 (valu_busy_frac is not: SQ_ACTIVE_INST_VALU counts one quad-cycle per instruction whatever
                          the instruction costs -- a kernel of 2.8-cycle adds reads 1.4). The FP64 classes come from <file>64<...>.txt
                          (profiles/r06_pmc64_c<N>.txt) when it exists.
  bound                   the classification bench.py prints: "hbm" > 60 % of 8 TB/s; "valu issue" floor share > 65 % (busy > 70 % when no
                          class counts exist); "lds pipe" > 60 %; otherwise "latency" with the shares spelled out
"""
import json
import re
import sys

N_SIMD, N_CU, N_XCD, GHZ = 1024, 256, 8, 2.4


def parse(path):
    d, cur = {}, None
    for l in open(path):
        if l.startswith("=="):
            cur = re.sub(r"^(void )?smilehip::", "", l[2:].strip())
            d[cur] = {}
            continue
        m = re.match(r"\s+(\S+)\s+avg/dispatch\s+([0-9.]+)\s+\(n=(\d+)\)", l)
        if m and cur:
            d[cur][m.group(1)] = float(m.group(2))
            d[cur].setdefault("_n", int(m.group(3)))
    return d


def class_costs():
    import os
    try:
        cal = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_valu_issue_calibration.json")))
        return cal["class_cost_cycles"]
    except Exception:
        return None


def main():
    path = sys.argv[1]
    min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
    out = {"source": path, "clock_ghz_assumed": GHZ, "kernels": {}}
    cost = class_costs()
    p64 = parse(path.replace("pmc_c", "pmc64_c")) if "pmc_c" in path else {}
    try:
        p64 = p64 or {}
    except Exception:
        p64 = {}
    for k, v in parse(path).items():
        if "GRBM_GUI_ACTIVE" not in v:
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / N_XCD
        ms = cyc / (GHZ * 1e6)
        if ms < min_ms:
            continue
        lds = v.get("SQ_LDS_IDX_ACTIVE", 0.0)
        hbm = (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024
        r = {
            "ms": ms, "dispatches": v.get("_n"),
            "valu_issue_frac": v.get("SQ_INSTS_VALU", 0.0) * 4 / N_SIMD / cyc,
            "valu_busy_frac": v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / N_SIMD / cyc,
            "lds_pipe_frac": lds / N_CU / cyc,
            "lds_bank_conflict_frac": v.get("SQ_LDS_BANK_CONFLICT", 0.0) / lds if lds else 0.0,
            "wait_inst_frac": v.get("SQ_WAIT_INST_ANY", 0.0) / v["SQ_WAVE_CYCLES"] if v.get("SQ_WAVE_CYCLES") else None,
            "waves_per_simd": v.get("SQ_WAVE_CYCLES", 0.0) * 4 / N_SIMD / cyc,
            "valu_insts": v.get("SQ_INSTS_VALU"), "salu_insts": v.get("SQ_INSTS_SALU"), "lds_insts": v.get("SQ_INSTS_LDS"),
            "vmem_rd_insts": v.get("SQ_INSTS_VMEM_RD"), "vmem_wr_insts": v.get("SQ_INSTS_VMEM_WR"),
            "hbm_bytes": hbm, "hbm_frac": hbm / (ms * 1e-3) / 8e12 if ms else None,
        }
        floor = None
        if cost and v.get("SQ_INSTS_VALU") and "SQ_INSTS_VALU_ADD_F32" in v:
            w = p64.get(k, {})
            cls = {"ADD_F32": v.get("SQ_INSTS_VALU_ADD_F32", 0), "MUL_F32": v.get("SQ_INSTS_VALU_MUL_F32", 0), "FMA_F32": v.get("SQ_INSTS_VALU_FMA_F32", 0),
                   "TRANS_F32": v.get("SQ_INSTS_VALU_TRANS_F32", 0), "CVT": v.get("SQ_INSTS_VALU_CVT", 0), "INT32": v.get("SQ_INSTS_VALU_INT32", 0),
                   "ADD_F64": w.get("SQ_INSTS_VALU_ADD_F64", 0), "MUL_F64": w.get("SQ_INSTS_VALU_MUL_F64", 0), "FMA_F64": w.get("SQ_INSTS_VALU_FMA_F64", 0),
                   "TRANS_F64": w.get("SQ_INSTS_VALU_TRANS_F64", 0), "INT64": w.get("SQ_INSTS_VALU_INT64", 0)}
            cls["other"] = max(0.0, v["SQ_INSTS_VALU"] - sum(cls.values()))
            floor = sum(cls[x] * cost[x] for x in cls) / N_SIMD / cyc
        r["valu_issue_floor_frac"] = floor
        if r["hbm_frac"] and r["hbm_frac"] > 0.6:
            r["bound"] = "hbm"
        elif floor is not None and floor > 0.65:
            r["bound"] = "valu issue (its instructions need >= %.0f %% of the launch at the measured issue cost of their classes)" % (100 * floor)
        elif floor is None and r["valu_busy_frac"] > 0.7:
            r["bound"] = "valu issue (%.0f %% of the SIMDs' issue cycles)" % (100 * r["valu_busy_frac"])
        elif r["lds_pipe_frac"] > 0.6:
            r["bound"] = "lds pipe (%.0f %%)" % (100 * r["lds_pipe_frac"])
        else:
            r["bound"] = "latency (valu issue floor %.0f %%, lds pipe %.0f %%, waves waiting on an instruction %.0f %% of their residence, %.1f waves per SIMD)" % (
                100 * (floor if floor is not None else r["valu_busy_frac"]), 100 * r["lds_pipe_frac"], 100 * (r["wait_inst_frac"] or 0.0), r["waves_per_simd"])
        out["kernels"][k] = r
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
