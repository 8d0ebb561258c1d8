#!/usr/bin/env python3
"""Per-phase wave residence of the jitter kernels (instrumented build: tools/ubench/variant_any.sh jitter phasejit
-DSMILEHIP_PHASE_TIMING). usage: phase_timing_jitter.py [utterances] [egemaps]; SMILEHIP_JITTER=utt times the per-utterance form."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SMILEHIP_LIB"] = os.path.join(ROOT, "tools", "ubench", "build", "libsmilehip_phasejit.so")
import torch  # noqa: E402
from opensmile_amd import capi, synth  # noqa: E402

NAMES = ["frame set-up + wave load", "cross-correlations", "peak / amplitudes / waveform / jitter sums", "harmonic + noise energy",
         "output", "workgroup set-up", "between frames (F0 load)"]


def main():
    ctx = capi.Context(0)
    egm = len(sys.argv) > 2 and sys.argv[2] == "egemaps"      # searchRangeRel 0.1 instead of 0.25: a fifth of the period in candidates
    plan = capi.Plan(ctx, capi.egemapsv02_config() if egm else capi.compare16_config())
    n_utt = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    pcm, off = synth.corpus_tiled(n_utt, 48000 if egm else 160000, n_unique=32)
    b = capi.Batch(plan, off)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((b.total_rows, 130), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L = capi.load()
    dbg = L.smilehip_debug_phase_jit
    dbg.restype = C.c_int
    dbg.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 130, st)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 8)()
    dbg(buf, 1)
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 130, st)
    torch.cuda.synchronize()
    dbg(buf, 0)
    v = np.array(list(buf)[0:7], dtype=np.float64)
    frames = max(int(buf[7]), 1)
    print(f"{n_utt} utterances, {frames} voiced frames; memtime ticks (10 ns) of wave residence per voiced frame: {v.sum() / frames:.0f}")
    for nm, x in zip(NAMES, v):
        print(f"  {nm:44s} {x / frames:10.0f} ticks/frame  {100 * x / v.sum():5.1f} %")


if __name__ == "__main__":
    main()
