#!/usr/bin/env python3
"""Per-phase wave residence of the IS09 frame kernel (instrumented build: tools/ubench/variant_any.sh is09 phaseis09 -DSMILEHIP_PHASE_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SMILEHIP_LIB", os.path.join(ROOT, "tools", "ubench", "build", "libsmilehip_phaseis09.so"))
import torch  # noqa: E402
from opensmile_amd import capi, synth  # noqa: E402

NAMES = ["utterance lookup + frame load", "ZCR", "pre-emphasis, window, RMS energy", "forward transform + magnitudes", "mel, log, DCT",
         "ACF + cepstrum (two inverse transforms, 257 double logs)", "cPitchACF"]


def main():
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.is09_lld_config())
    pcm, off = synth.corpus_tiled(2000, 160000, n_unique=32)
    b = capi.Batch(plan, off)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((b.total_rows, 32), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L = capi.load()
    dbg = L.smilehip_debug_phase_is09
    dbg.restype = C.c_int
    dbg.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 32, st)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 16)()
    dbg(buf, 1)
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 32, st)
    torch.cuda.synchronize()
    dbg(buf, 0)
    v = np.array(list(buf)[:len(NAMES)], dtype=np.float64)
    print(f"lld_is09_frame_wave: memtime ticks per frame and wave: {v.sum() / b.total_frames:.0f}")
    for nm, x in zip(NAMES, v):
        print(f"  {nm:58s} {x / b.total_frames:8.0f} ticks/frame  {100 * x / v.sum():5.1f} %")


if __name__ == "__main__":
    main()
