#!/usr/bin/env python3
"""Per-phase wave residence of lld_f0_sweep (instrumented build: tools/ubench/variant_any.sh f0 phasef0 -DSMILEHIP_PHASE_TIMING)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SMILEHIP_LIB", os.path.join(ROOT, "tools", "ubench", "build", "libsmilehip_phasef0.so"))
import torch  # noqa: E402
from opensmile_amd import capi, synth  # noqa: E402

NAMES = ["pass 1: a block's first 14 bins", "pass 1: wait for the next block", "pass 1: last two bins + next line from LDS",
         "pass 2: wait for the block + line from LDS", "pass 2: recurrences", "pass 2: target points + output", "-", "pass 1 prologue"]


def main():
    ctx = capi.Context(0)
    plan = capi.Plan(ctx, capi.compare16_f0_config())
    pcm, off = synth.corpus_tiled(int(sys.argv[1]) if len(sys.argv) > 1 else 2000, 160000, n_unique=32)
    b = capi.Batch(plan, off)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((b.total_frames, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    L = capi.load()
    dbg = L.smilehip_debug_phase_sweep
    dbg.restype = C.c_int
    dbg.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 2, st)
    torch.cuda.synchronize()
    buf = (C.c_uint64 * 8)()
    dbg(buf, 1)
    n = 3
    for _ in range(n):
        b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), 2, st)
    torch.cuda.synchronize()
    dbg(buf, 0)
    v = np.array(list(buf), dtype=np.float64) / n
    waves = b.total_frames / 64.0
    print(f"frames {b.total_frames}; s_memtime ticks (100 MHz) per wave (64 frames), 33 blocks per pass: total {v.sum() / waves:.0f}")
    for nm, x in zip(NAMES, v):
        print(f"  {nm:46s} {x / waves:9.0f} ticks/wave  {100 * x / v.sum():5.1f} %")


if __name__ == "__main__":
    main()
