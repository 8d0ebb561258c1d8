#!/usr/bin/env python3
"""Per-phase wave residence of lld_gemaps_harm (instrumented build: tools/ubench/variant_any.sh gemaps phasegm -DSMILEHIP_PHASE_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SMILEHIP_LIB"] = os.path.join(ROOT, "tools", "ubench", "build", "libsmilehip_phasegm.so")
import torch  # noqa: E402
from opensmile_amd import capi, synth  # noqa: E402

NAMES = ["load + window + FFT + magnitudes", "ACF (inverse transform)", "HNR peak search", "harmonic peaks", "log magnitudes + duplicates",
         "formant amplitudes + output"]


def main():
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.egemapsv02_config())
    pcm, off = synth.corpus_tiled(2000, 48000, n_unique=32)
    b = capi.Batch(plan, off)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((b.total_rows, 25), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L = capi.load()
    dbg = L.smilehip_debug_phase_gm
    dbg.restype = C.c_int
    dbg.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 25, st)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 16)()
    dbg(buf, 1)
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 25, st)
    torch.cuda.synchronize()
    dbg(buf, 0)
    v = np.array(list(buf)[:len(NAMES)], dtype=np.float64)
    print(f"lld_gemaps_harm: memtime ticks per voiced frame and wave (all frames: {b.total_frames})")
    for nm, x in zip(NAMES, v):
        print(f"  {nm:36s} {x / b.total_frames:8.0f} ticks/frame  {100 * x / v.sum():5.1f} %")
    names20 = ["frame into LDS, energy2", "FFT", "magnitudes + cSpecResample rows", "mel / auditory spectrum / MFCC", "GeMAPS spectral descriptors"]
    v = np.array(list(buf)[8:8 + len(names20)], dtype=np.float64)
    print(f"lld_gemaps_frame20: memtime ticks per frame and wave")
    for nm, x in zip(names20, v):
        print(f"  {nm:36s} {x / b.total_frames:8.0f} ticks/frame  {100 * x / v.sum():5.1f} %")


if __name__ == "__main__":
    main()
