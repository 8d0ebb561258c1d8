#!/usr/bin/env python3
"""Per-phase wave residence of the F0 frame kernel (instrumented build: tools/ubench/variant_f0.sh phasef0 -DSMILEHIP_PHASE_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SMILEHIP_LIB"] = os.path.join(ROOT, "tools", "ubench", "build", "libsmilehip_phasef0.so")
import torch  # noqa: E402
from opensmile_amd import capi, synth  # noqa: E402

NAMES = ["rows from global (lld_f0_cand)", "-", "interp + summation + top six", "mean", "candidates + output"]


def main():
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.compare16_f0_config())
    pcm, off = synth.corpus_tiled(1000, 160000, n_unique=32)
    b = capi.Batch(plan, off)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((b.total_frames, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L = capi.load()
    dbg = L.smilehip_debug_phase_f0
    dbg.restype = C.c_int
    dbg.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 2, st)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 16)()
    dbg(buf, 1)
    n = 3
    for _ in range(n):
        b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 2, st)
    torch.cuda.synchronize()
    dbg(buf, 0)
    v = np.array(list(buf)[:len(NAMES)], dtype=np.float64) / n
    tot = v.sum()
    print(f"memtime ticks per frame per wave: {tot / b.total_frames:.0f}")
    for nm, x in zip(NAMES, v):
        print(f"  {nm:26s} {x / b.total_frames:8.0f} ticks/frame  {100 * x / tot:5.1f} %")
    sub = np.array(list(buf)[8:12], dtype=np.float64) / n
    print("inside 'interp + summation + top six' (f0_shs):")
    for nm, x in zip(["spline evaluation + auditory weighting", "harmonic summation", "exact-mean test", "local maxima + top six"], sub):
        print(f"  {nm:40s} {x / b.total_frames:8.0f} ticks/frame  {100 * x / sub.sum():5.1f} %")


if __name__ == "__main__":
    main()
