#!/usr/bin/env python3
"""Per-phase wave residence of the jitter kernel (instrumented build: tools/ubench/variant_f0.sh phasef0 -DSMILEHIP_PHASE_TIMING;
counters 5..9 of the array the F0 frame kernel shares)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SMILEHIP_LIB"] = os.path.join(ROOT, "tools", "ubench", "build", "libsmilehip_phasef0.so")
import torch  # noqa: E402
from opensmile_amd import capi, synth  # noqa: E402

NAMES = ["frame set-up + wave load", "cross-correlations", "peak / amplitudes / waveform / jitter sums", "harmonic + noise energy",
         "output"]


def main():
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.compare16_config())
    pcm, off = synth.corpus_tiled(1000, 160000, n_unique=32)
    b = capi.Batch(plan, off)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((b.total_rows, 130), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L = capi.load()
    dbg = L.smilehip_debug_phase_f0
    dbg.restype = C.c_int
    dbg.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 130, st)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 16)()
    dbg(buf, 1)
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 130, st)
    torch.cuda.synchronize()
    dbg(buf, 0)
    v = np.array(list(buf)[5:10], dtype=np.float64)
    frames = 995 * 1000
    print(f"memtime ticks per frame (one wave per utterance): {v.sum() / frames:.0f}")
    for nm, x in zip(NAMES, v):
        print(f"  {nm:44s} {x / frames:10.0f} ticks/frame  {100 * x / v.sum():5.1f} %")


if __name__ == "__main__":
    main()
