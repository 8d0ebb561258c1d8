#!/usr/bin/env python3
"""Throughput of the IS09_emotion and ComParE_2016 (groups A+B) LLD chains on one GPU
(configs 3/4 of BASELINE.json, per-GPU share). Not the driver's bench line (bench.py);
prints one JSON object per chain: frames/s with PCM resident in HBM."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--sets", default="is09,compare")
    ap.add_argument("--func", action="store_true", help="compare_full / egemaps: also time the set's functionals level")
    ap.add_argument("--seconds", type=float, default=10.0, help="utterance length (config 5 of BASELINE.json: 3)")
    ap.add_argument("--rate", type=int, default=16000, help="sample rate of the synthetic corpus (the big sets run at 8 .. 48 kHz)")
    args = ap.parse_args()
    import torch
    from opensmile_amd import capi, synth
    ctx = capi.Context(0)
    pcm, off = synth.corpus_tiled(args.utts, int(round(args.seconds * args.rate)), n_unique=32, fs=args.rate)
    d_pcm = torch.from_numpy(pcm).cuda()
    for name in args.sets.split(","):
        if name.upper() in ("MFCC12_E_D_A", "MFCC12_0_D_A_Z", "MFCC12_E_D_A_Z", "PLP_E_D_A", "PLP_0_D_A_Z", "PLP_E_D_A_Z"):
            cfg_fn = (lambda n=name.upper(): capi.htk_variant_config(n))
        else:
            cfg_fn = None
        cfg = cfg_fn() if cfg_fn else {"is09": capi.is09_lld_config, "compare": capi.compare16_ab_config, "mfcc": capi.mfcc12_0_d_a_config,
               "plp": capi.plp_0_d_a_config, "f0": capi.compare16_f0_config, "compare_full": capi.compare16_config,
               "egemaps": capi.egemapsv02_config}[name]()
        cfg.sample_rate = float(args.rate)
        plan = capi.Plan(ctx, cfg)
        b = capi.Batch(plan, off)
        n_out = plan.geometry.n_out
        rows = int(b.frame_offsets[-1])
        d_out = torch.empty((max(rows, 1), n_out), dtype=torch.float32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), n_out, st)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            b.run_device(d_pcm.data_ptr(), d_out.data_ptr(), n_out, st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        rec = {"set": name, "rate": args.rate, "utterances": args.utts, "seconds": args.seconds, "frames": b.total_frames, "rows": rows, "cols": n_out,
               "ms_per_step": dt * 1e3, "frames_per_s": b.total_frames / dt}
        if name == "compare_full" and args.func:
            # the functionals level on top (6373 values per utterance), LLD matrix resident
            import ctypes as C
            L = capi.load()
            d_func = torch.empty((args.utts, 6373), dtype=torch.float32, device="cuda")
            def run_func():
                capi._check(L.smilehip_batch_functionals_compare16(plan._h, b._h, C.c_void_p(d_out.data_ptr()), n_out,
                                                                   C.c_void_p(d_func.data_ptr()), 6373, C.c_void_p(st)))
            run_func()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run_func()
            torch.cuda.synchronize()
            df = (time.perf_counter() - t0) / args.steps
            rec.update({"func_ms_per_step": df * 1e3, "lld_plus_func_frames_per_s": b.total_frames / (dt + df),
                        "func_values_per_utt": 6373})
        if name == "egemaps" and args.func:
            # the 88 functionals on top (the smoothed levels stay resident in the batch's scratch)
            import ctypes as C
            L = capi.load()
            d_func = torch.empty((args.utts, 88), dtype=torch.float32, device="cuda")
            def run_func():
                capi._check(L.smilehip_batch_functionals_egemaps(plan._h, b._h, C.c_void_p(d_func.data_ptr()), 88, C.c_void_p(st)))
            run_func()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                run_func()
            torch.cuda.synchronize()
            df = (time.perf_counter() - t0) / args.steps
            rec.update({"func_ms_per_step": df * 1e3, "lld_plus_func_frames_per_s": b.total_frames / (dt + df),
                        "lld_plus_func_utterances_per_s": args.utts / (dt + df), "func_values_per_utt": 88})
        print(json.dumps(rec), flush=True)
        b.close()
        plan.close()


if __name__ == "__main__":
    main()
