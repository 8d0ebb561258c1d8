import numpy as np, sys
sys.path.insert(0, "/root/repo")
from opensmile_amd import capi, synth
ctx = capi.Context(0)
plan = capi.Plan(ctx, capi.compare16_f0_config())
for n_samp in (160000, 48000):
    lens=[n_samp]*8
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    pcm = np.concatenate([synth.utterance(2+i, n) for i, n in enumerate(lens)])
    b = capi.Batch(plan, off)
    out = b.run_host(pcm)
    for i in range(8):
        f0 = out[b.frame_offsets[i]:b.frame_offsets[i+1], 0]
        v = f0 > 0
        starts = np.flatnonzero(v & ~np.concatenate([[False], v[:-1]]))
        ends = np.flatnonzero(v & ~np.concatenate([v[1:], [False]]))
        L = ends - starts + 1
        print(n_samp, i, "frames", len(f0), "voiced %.2f" % v.mean(), "segments", len(starts), "longest", L.max() if len(L) else 0)
