"""Throughput of the per-component operators of lld_stage4_kernels.hip (the components the INTERSPEECH 2010 - 2012 sets add) at batch
scale: IS10_paraling's shapes over 1000 x 10 s (998 000 frames of 25 ms, 995 000 of 60 ms). One JSON line per operator."""
import ctypes as C
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opensmile_amd import capi

L = capi.load()
ctx = capi.Context(0)
NF = 998000
g = torch.Generator(device="cuda").manual_seed(1)


def timed(name, fn, frames, bytes_per_frame, steps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    print(json.dumps({"op": name, "frames": frames, "ms": ms, "frames_per_s": frames / ms * 1e3, "alg_GBps": frames * bytes_per_frame / ms / 1e6}))


frames = (torch.randn((NF, 400), device="cuda", generator=g) * 0.2)
out1 = torch.empty((NF, 1), device="cuda")
timed("cIntensity loudness (400-sample frames)", lambda: capi._check(L.smilehip_intensity_frames(ctx._h, frames.data_ptr(), 400, 400, 2, out1.data_ptr(), 1, NF, None)), NF, 8)
del frames
spec = torch.randn((NF // 4, 512), device="cuda", generator=g)
n_out, k_max, nd = C.c_int64(0), C.c_int64(0), C.c_double(0.0)
capi._check(L.smilehip_specresample_geometry(512, 512 / 16000.0, 400 / 16000.0, 1 / 16000.0, 11000.0, C.byref(n_out), C.byref(k_max), C.byref(nd)))
h = k_max.value // 2
ct, st = np.zeros(h * n_out.value, np.float32), np.zeros(h * n_out.value, np.float32)
capi._check(L.smilehip_specresample_tables(512, n_out.value, k_max.value, nd.value, ct.ctypes.data, st.ctypes.data))
d_ct, d_st = torch.from_numpy(ct).cuda(), torch.from_numpy(st).cuda()
res = torch.empty((NF // 4, n_out.value), device="cuda")
timed("cSpecResample 512 -> 275 (a quarter of the frames)", lambda: capi._check(L.smilehip_specresample_table_frames(
    ctx._h, spec.data_ptr(), 512, 512, n_out.value, k_max.value, d_ct.data_ptr(), d_st.data_ptr(), res.data_ptr(), n_out.value, NF // 4, None)),
    NF // 4, 4 * (512 + 275))
lpc = torch.empty((NF // 4, 8), device="cuda")
timed("cLpc p = 8 on 275 samples (a quarter)", lambda: capi._check(L.smilehip_lpc_acf_frames(ctx._h, res.data_ptr(), n_out.value, n_out.value, 8, lpc.data_ptr(), 8, NF // 4, None)),
      NF // 4, 4 * (275 + 8))
del spec
lpc_all = lpc.repeat(4, 1)[:NF].contiguous()
lsp = torch.empty_like(lpc_all)
timed("cLsp p = 8", lambda: capi._check(L.smilehip_lsp_frames(ctx._h, lpc_all.data_ptr(), 8, 8, lsp.data_ptr(), 8, NF, None)), NF, 64)
mel = torch.rand((NF, 8), device="cuda", generator=g) + 1e-3
lg = torch.empty_like(mel)
timed("cVectorOperation log, 8 bands", lambda: capi._check(L.smilehip_vecop_frames(ctx._h, 2, 1.0, 0.0, mel.data_ptr(), 8, 8, lg.data_ptr(), 8, NF, None)), NF, 64)
T = 995
cand = torch.rand((1000 * T, 18), device="cuda", generator=g)
cand[:, :6] = cand[:, :6] * 400 + 60
off = torch.arange(0, 1001, dtype=torch.int64, device="cuda") * T
o2 = torch.empty((1000 * T, 2), device="cuda")
wr = torch.zeros(1000, dtype=torch.int64, device="cuda")
timed("cPitchSmoother F0finEnv + voicing, 1000 streams x 995 frames", lambda: capi._check(L.smilehip_pitch_smoother_rows(
    ctx._h, 6, 0.7, 0, 1, 2 | 8, cand.data_ptr(), 18, off.data_ptr(), 1000, 0, None, 0, o2.data_ptr(), 2, wr.data_ptr(), None)), 1000 * T, 4 * 20)
