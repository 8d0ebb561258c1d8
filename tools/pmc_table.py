#!/usr/bin/env python3
"""Per-kernel roofline table from the counter summaries of tools/pmc_all.sh (profiles/r02_pmc_*.txt): duration (GRBM_GUI_ACTIVE
over the 8 XCDs at 2.4 GHz), HBM traffic (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, KB), f32 operations ((ADD + MUL + TRANS) + 2 FMA
wave-instructions x 64 lanes), the fractions of the 8 TB/s and 157 TFLOP/s roofs, LDS pipe activity and the share of it that
is bank conflicts. usage: pmc_table.py profiles/r02_pmc_compare_full.txt [min_ms]"""
import re
import sys


def main():
    path = sys.argv[1]
    min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
    d, cur = {}, None
    for l in open(path):
        if l.startswith("=="):
            cur = l[2:].strip()
            d[cur] = {}
            continue
        m = re.match(r"\s+(\S+)\s+avg/dispatch\s+([0-9.]+)\s+\(n=(\d+)\)", l)
        if m and cur:
            d[cur][m.group(1)] = float(m.group(2))
    print("| kernel | ms | HBM MB | of 8 TB/s | f32 GFLOP | of 157 TF | VALU issue | LDS pipe busy | of it conflicts |")
    print("|---|---|---|---|---|---|---|---|---|")
    for k, v in d.items():
        if "GRBM_GUI_ACTIVE" not in v:
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        ms = cyc / 2.4e6
        if ms < min_ms:
            continue
        hbm = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024
        flop = 64 * (v.get("SQ_INSTS_VALU_ADD_F32", 0) + v.get("SQ_INSTS_VALU_MUL_F32", 0) + v.get("SQ_INSTS_VALU_TRANS_F32", 0)
                     + 2 * v.get("SQ_INSTS_VALU_FMA_F32", 0))
        lds = v.get("SQ_LDS_IDX_ACTIVE", 0)
        name = re.sub(r"^(void )?smilehip::", "", k)
        print(f"| `{name[:44]}` | {ms:.2f} | {hbm / 1e6:.0f} | {hbm / (ms * 1e-3) / 8e12 * 100:.1f} % | {flop / 1e9:.1f} | "
              f"{flop / (ms * 1e-3) / 157e12 * 100:.1f} % | {v.get('SQ_INSTS_VALU', 0) / 1024 * 2.7 / cyc * 100:.0f} % | "
              f"{lds / 256 / cyc * 100:.0f} % | {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(lds, 1) * 100:.0f} % |")


if __name__ == "__main__":
    main()
