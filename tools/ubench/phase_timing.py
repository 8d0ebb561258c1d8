#!/usr/bin/env python3
"""Per-phase wave residence of the fused MFCC kernel (instrumented build: tools/ubench/variant.sh phase -DSMILEHIP_PHASE_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SMILEHIP_LIB"] = os.path.join(ROOT, "tools", "ubench", "build", "libsmilehip_phase.so")
import torch  # noqa: E402
from opensmile_amd import capi, synth  # noqa: E402

NAMES = ["loop/tile setup", "prefetch wait + stage", "frame load x window", "dft16 #1 + twiddle", "transposes",
         "dft16 #2", "untangle + power", "prefetch issue", "mel units", "band sums + log", "DCT + store"]


def main():
    ctx = capi.Context(0)
    plan = capi.Plan(ctx)
    pcm, off = synth.corpus_tiled(1000, 160000, n_unique=32)
    b = capi.Batch(plan, off)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((b.total_frames, 39), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L = capi.load()
    dbg = L.smilehip_debug_phase
    dbg.restype = C.c_int
    dbg.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    for _ in range(3):
        b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 39, st)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 16)()
    dbg(buf, 1)
    n = 10
    for _ in range(n):
        b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 39, st)
    torch.cuda.synchronize()
    dbg(buf, 0)
    v = np.array(list(buf)[:11], dtype=np.float64) / n
    passes = b.total_frames / 4
    tot = v.sum()
    print(f"memtime ticks per pass per wave: {tot / passes:.0f}")
    for nm, x in zip(NAMES, v):
        print(f"  {nm:26s} {x / passes:8.0f} ticks/pass  {100 * x / tot:5.1f} %")


if __name__ == "__main__":
    main()
