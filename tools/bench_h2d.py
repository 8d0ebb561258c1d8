#!/usr/bin/env python3
"""PCIe-inclusive throughput of the MFCC chain: the corpus starts in (pinned) host memory, chunks are
copied to the device on one stream while the previous chunk is processed on another (two batches, two
device buffers), features optionally come back to the host. Not the driver's bench line (that one keeps
PCM resident in HBM, bench.py); DESIGN.md quotes this number next to it."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(utts=1000, chunks=10, reps=5, d2h=False):
    args = argparse.Namespace(utts=utts, chunks=chunks, reps=reps, d2h=d2h)
    import torch
    from opensmile_amd import capi, synth
    ctx = capi.Context(0)
    plan = capi.Plan(ctx)
    per = args.utts // args.chunks
    pcm, off = synth.corpus_tiled(per, 160000, n_unique=min(32, per))
    h_pcm = [torch.from_numpy(pcm.copy()).pin_memory() for _ in range(args.chunks)]
    batches = [capi.Batch(plan, off) for _ in range(2)]
    frames = batches[0].total_frames
    d_pcm = [torch.empty(len(pcm), dtype=torch.int16, device="cuda") for _ in range(2)]
    d_out = [torch.empty((frames, 39), dtype=torch.float32, device="cuda") for _ in range(2)]
    h_out = [torch.empty((frames, 39), dtype=torch.float32).pin_memory() for _ in range(2)] if args.d2h else None
    copy_s, comp_s = torch.cuda.Stream(), torch.cuda.Stream()
    ev_in = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]

    def run_once():
        for i in range(args.chunks):
            k = i & 1
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(ev_done[k])            # the buffer's previous chunk has been consumed
                d_pcm[k].copy_(h_pcm[i], non_blocking=True)
                ev_in[k].record(copy_s)
            comp_s.wait_event(ev_in[k])
            batches[k].run_device(d_pcm[k].data_ptr(), d_out[k].data_ptr(), 39, comp_s.cuda_stream)
            if args.d2h:
                with torch.cuda.stream(comp_s):
                    h_out[k].copy_(d_out[k], non_blocking=True)
            ev_done[k].record(comp_s)
        torch.cuda.synchronize()

    run_once()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        run_once()
    dt = (time.perf_counter() - t0) / args.reps
    total = frames * args.chunks
    return {"workload": f"MFCC12_0_D_A, {args.chunks} chunks x {per} x 10 s, PCM from pinned host memory, copy/compute overlapped"
                        + (", features back to pinned host memory" if args.d2h else ""),
            "frames": total, "ms": dt * 1e3, "frames_per_s": total / dt,
            "h2d_GBps": 2.0 * len(pcm) * args.chunks / dt / 1e9}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=1000)
    ap.add_argument("--chunks", type=int, default=10)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--d2h", action="store_true", help="also copy the feature matrix back to pinned host memory")
    args = ap.parse_args()
    print(json.dumps(measure(args.utts, args.chunks, args.reps, args.d2h)))


if __name__ == "__main__":
    main()
