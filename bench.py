#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config 2 (default), or --config 3 | 4 | 5.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]

--config 3: IS09_emotion LLD (16 + 16 delta) on 10 000 x 10 s; --config 4: the whole ComParE_2016 LLD level (130 columns) on
12 500 x 10 s (the per-GPU share of 100 000 utterances over 8 GPUs); --config 5: eGeMAPSv02 LLD + the 88 functionals on
125 000 x 3 s (the per-GPU share of 10^6). Same JSON schema, the config's own workload / roofline kernel / CPU baseline
(the real SMILExtract on that conf). Steps and warm-up default to what finishes in about a minute.

A "step" is one pass of the fused LLD chain (MFCC12_0_D_A: int16 PCM ->
13 MFCC + delta + accel) over one batch of synthetic audio: 1000 utterances x
10 s, 16 kHz mono, 25 ms / 10 ms => 998 000 frames per GPU (configs[1]). PCM is
resident in HBM before the timed region starts; outputs stay in HBM.

N > 1, one rank per GPU over RCCL: either the driver launches the ranks (python -m
torch.distributed.run ... bench.py --gpus N: RANK / WORLD_SIZE are in the environment) or
`python bench.py --gpus N` launches them itself the same way (launch_ranks below; with fewer
devices than ranks it says so and falls back to gloo with ranks sharing devices -- a plumbing
check, not a measurement). WORLD_SIZE must equal --gpus. Utterances shard embarrassingly, every
rank runs its own 1000 x 10 s batch ("weak" scaling, no data-path collective); the only
collective in the path -- the gather of the feature matrices to rank 0 -- is
opensmile_amd/gather.py and is reported next to `value` (gather_ms, value_incl_gather), never in it.

The default N = 1 line also carries (a) "configs": BASELINE.json's configs 3, 4 and 5 at their
per-GPU sizes, 2-3 timed steps each, each with its own roofline and real-binary CPU baseline
(--no-configs skips them), (b) "h2d_inclusive" / "to_host": the same chain with the PCM starting
in pinned host memory / the features ending there (SURVEY 8d's GPU timing protocol).

Rank 0 prints ONE JSON line (see the driver contract) with two extra objects:
  roofline     -- dominant kernel (fused MFCC): algorithmic bytes per launch
                  (372 B/frame) / average launch duration
                  measured with HIP events on the launch stream, vs 8 TB/s
  cpu_baseline -- the REAL reference (oracle/_ref/SMILExtract) timed on this
                  box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_UTT = 1000          # config 2: 1000 x 10 s
UTT_SAMPLES = 160000
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3
ALG_BYTES_PER_FRAME_MAIN = 2 * 160 + 4 * 13      # int16 hop in + 13 f32 out
ALG_BYTES_PER_FRAME_CHAIN = 2 * 160 + 4 * 39     # incl. delta/accel columns
ALG_FLOP_PER_FRAME = 1.52e4

# --config: (capi config factory, utterances per GPU, samples per utterance, default steps, default warm-up, workload text,
#            roofline kernel name, algorithmic bytes per 10 ms frame of that kernel, what the bytes are, reference conf, output
#            option of the reference binary, frames per file of the reference run)
CONFIGS = {
    3: dict(cfg="is09_lld_config", utts=10000, samples=160000, steps=10, warmup=3,
            workload="IS09_emotion LLD (MFCC 1-12, RMS energy, ZCR, voiceProb, F0 via cAcf / cPitchACF; 16 LLD + 16 delta) on "
                     "10 000 x 10 s synthetic 16 kHz mono int16 per GPU, 25 ms / 10 ms, PCM resident in HBM",
            kernel="lld_is09_frame_quad (+ lld_pitch_smooth)", pmc_kernels=["lld_is09_frame", "lld_pitch_smooth"], alg_bytes=2 * 160 + 4 * 16,
            alg_note="int16 hop in + 16 f32 pre-smoothing columns out per frame (SURVEY 8d counts 448 B for the whole chain incl. deltas)",
            conf="is09-13/IS09_emotion.conf", opt="-lldhtkoutput"),
    4: dict(cfg="compare16_config", utts=12500, samples=160000, steps=4, warmup=1,
            workload="ComParE_2016 whole LLD level (130 columns: F0 group incl. Viterbi + jitter / shimmer, groups A + B, deltas) on "
                     "12 500 x 10 s synthetic 16 kHz mono int16 per GPU (config 4's share of 100 000 utterances over 8 GPUs), "
                     "PCM resident in HBM",
            kernel="lld_compare_frame_quad (+ RASTA scan, group A)", pmc_kernels=["lld_compare_frame", "lld_compare_rasta", "lld_compare_groupA"],
            alg_bytes=2 * 160 + 4 * (4 + 55),
            alg_note="int16 hop in + 59 f32 pre-smoothing columns of groups A + B out per 20 ms frame",
            conf="compare16/ComParE_2016.conf", opt="-lldhtkoutput"),
    5: dict(cfg="egemapsv02_config", utts=125000, samples=48000, steps=3, warmup=1,
            workload="eGeMAPSv02 LLD (25 columns) + 88 functionals per utterance on 125 000 x 3 s synthetic 16 kHz mono int16 per GPU "
                     "(config 5's share of 10^6 utterances over 8 GPUs), PCM resident in HBM",
            kernel="lld_gemaps_frame20 + lld_gemaps_lpc + lld_gemaps_formants", pmc_kernels=["lld_gemaps_frame20", "lld_gemaps_lpc", "lld_gemaps_formants"],
            alg_bytes=2 * 160 + 4 * (12 + 222 + 12 + 10),
            alg_note="int16 hop in + 12 f32 raw descriptors + 222 f32 of cSpecResample's input + 12 LP + 10 formant values out per 20 ms frame",
            conf="egemaps/v02/eGeMAPSv02.conf", opt="-htkoutput"),
}


# Algorithmic bytes per 10 ms frame of the kernels that can dominate a step of configs 3-5: what that STAGE of the chain has to
# read and write if nothing is wasted -- the int16 hop where the kernel reads the samples, the values it takes from the stage in
# front of it and the values it hands on. The three kernels of the F0 front end (spec -> sweep -> cand) are one stage (the rows they
# pass each other are not algorithmic: SURVEY 8d): each is priced with the stage's bytes, 320 in + 6 candidates x (f0, score) + 3
# frame values out.
KERNEL_ALG_BYTES = {
    "lld_jitter_runs": (2 * 160 + 4 + 4 * 3, "int16 hop + final F0 in, jitterLocal / jitterDDP / shimmerLocal out"),
    "lld_f0_jitter": (2 * 160 + 4 + 4 * 3, "int16 hop + final F0 in, jitterLocal / jitterDDP / shimmerLocal out"),
    "lld_f0_spec": (2 * 160 + 4 * 15, "F0 front end (spec + sweep + cand are one stage): int16 hop in, 6 x (f0, score) + 3 frame values out"),
    "lld_f0_sweep": (2 * 160 + 4 * 15, "F0 front end (one stage with spec and cand), as lld_f0_spec"),
    "lld_f0_cand9": (2 * 160 + 4 * 15, "F0 front end (one stage with spec and sweep), as lld_f0_spec"),
    "lld_f0_cand": (2 * 160 + 4 * 15, "F0 front end (one stage with spec and sweep), as lld_f0_spec"),
    "lld_f0_viterbi": (4 * 12 + 4 * 3, "6 x (f0, score) in, final F0 + voicing + score out"),
    "lld_compare_frame_quad": (2 * 160 + 4 * (4 + 55), "int16 hop in + 59 f32 pre-smoothing columns of groups A + B out"),
    "lld_compare_frame_wave3t": (2 * 160 + 4 * (4 + 55), "int16 hop in + 59 f32 pre-smoothing columns of groups A + B out"),
    "lld_is09_frame_quad": (2 * 160 + 4 * 16, "int16 hop in + 16 f32 pre-smoothing columns out"),
    "lld_chain_tiled": (None, "smoothing / delta chain: every column of the level read once and written once (8 B per output cell)"),
    "lld_gemaps_harm": (2 * 160 + 4 + 4 * 9, "int16 hop (60 ms frames) + final F0 in, 9 harmonic / formant-amplitude values out"),
    "lld_gemaps_frame20_quad": (2 * 160 + 4 * (12 + 222), "int16 hop in + 12 raw descriptors + 222 f32 of cSpecResample's input out"),
    "lld_gemaps_frame20": (2 * 160 + 4 * (12 + 222), "int16 hop in + 12 raw descriptors + 222 f32 of cSpecResample's input out"),
    "lld_gemaps_lpc": (4 * 222 + 4 * 12, "222 f32 resampled spectrum in, 12 LP values out"),
    "lld_gemaps_formants": (4 * 12 + 4 * 10, "12 LP values in, 10 formant values out"),
}
# SURVEY 8(d)'s per-frame figure of the WHOLE chain of a config: the int16 hop in, every column of the output level out
CHAIN_ALG_BYTES = {3: 2 * 160 + 4 * 32, 4: 2 * 160 + 4 * 130, 5: 2 * 160 + 4 * 25}


def cpu_baseline(max_seconds=25.0, conf_rel="mfcc/MFCC12_0_D_A.conf", opt="-O", n_samples=UTT_SAMPLES, frames_per_file=998, max_files=14000):
    """Time the real reference binary (one process per file, HTK output to
    /dev/shm, log level 0) on all host cores; bounded sample, scaled to frames/s."""
    from oracle import lldo
    from opensmile_amd import synth
    exe = os.path.join(lldo.REF_DIR, "SMILExtract")
    conf = os.path.join(lldo.REF_DIR, "config", *conf_rel.split("/"))
    cores = os.cpu_count() or 1
    if not os.path.exists(exe) and conf_rel != "mfcc/MFCC12_0_D_A.conf":
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference", "sample": "oracle/_ref/SMILExtract not built"}
    if not os.path.exists(exe):
        # fall back to the C restatement ("port"), single thread
        cfg = lldo.default_cfg()
        pcm = synth.utterance(2, UTT_SAMPLES)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min(max_seconds, 10.0):
            lldo.mfcc_chain(cfg, pcm)
            n += 1
        dt = time.perf_counter() - t0
        return {"value": n * 998 / dt, "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": f"{n} x 10 s utterances through oracle/lld_oracle.c (1 thread)"}
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=base) as td:
        n_unique = 8
        for i in range(n_unique):
            lldo.write_wav(os.path.join(td, f"u{i}.wav"), synth.utterance(2 + i, n_samples))
        # calibrate on a few files, 1 core
        t0 = time.perf_counter()
        n_cal = 8
        for i in range(n_cal):
            subprocess.run([exe, "-C", conf, "-I", os.path.join(td, f"u{i % n_unique}.wav"),
                            opt, os.path.join(td, "cal.htk"), "-l", "0"],
                           cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        per_file = (time.perf_counter() - t0) / n_cal
        one_core = frames_per_file / per_file
        # ~10-15 s of wall clock on all cores; outputs reuse 4*cores file names
        n_files = int(max(cores * 4, min(max_files, min(12.0, max_seconds / 2.0) / per_file * cores)))
        jobs = "\n".join(f"{i % n_unique} {i % (4 * cores)}" for i in range(n_files))
        cmd = (f"xargs -P {cores} -L 1 sh -c '{exe} -C {conf} -I {td}/u$0.wav {opt} {td}/o$1.htk "
               f"-l 0 >/dev/null 2>&1'")
        t0 = time.perf_counter()
        subprocess.run(cmd, shell=True, input=jobs.encode(), cwd=td, check=True)
        dt = time.perf_counter() - t0
        produced = sum(1 for f in os.listdir(td) if f.startswith("o") and f.endswith(".htk"))
        done = n_files if produced >= min(n_files, 4 * cores) else 0
    return {"value": done * frames_per_file / dt, "unit": "frames/s", "cores": cores, "kind": "reference",
            "sample": (f"{done} x {n_samples / 16000.0:g} s files, one SMILExtract -C {conf_rel} process per file, {cores} in parallel "
                       f"(xargs -P), {opt} /dev/shm/*.htk -l 0; {dt:.1f} s wall"),
            "one_core_value": one_core}


def _counters(config):
    """Counter evidence of the latest round's --pmc passes (tools/pmc_counters_json.py -> profiles/r06_pmc_c<N>.json): what bounds the
    config's kernels by the SQ counters, and the VALU issue fraction of the roofline kernel(s). None when the file is absent."""
    for rnd in ("r06", "r05"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_c{config}.json")
        try:
            return json.load(open(path)), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def _counter_fields(config, kernels):
    """roofline.bound_by_counters / valu_issue_frac / ... of the named kernels, from the committed counter summary (never a literal)."""
    cj, src = _counters(config)
    if not cj:
        return {"bound_by_counters": None, "valu_issue_frac": None, "counters_source": None}
    mine = [v for k, v in cj.get("kernels", {}).items() if any(n in k for n in kernels)]
    if not mine:
        return {"bound_by_counters": None, "valu_issue_frac": None, "counters_source": src}
    w = sum(v["ms"] for v in mine) or 1.0
    avg = lambda key: sum(v.get(key, 0.0) * v["ms"] for v in mine) / w
    top = max(mine, key=lambda v: v["ms"])
    return {"bound_by_counters": top.get("bound"), "valu_issue_floor_frac": avg("valu_issue_floor_frac") if all(v.get("valu_issue_floor_frac") is not None for v in mine) else None,
            "valu_issue_frac": avg("valu_issue_frac"), "valu_busy_frac": avg("valu_busy_frac"),
            "lds_pipe_frac": avg("lds_pipe_frac"), "lds_bank_conflict_frac": avg("lds_bank_conflict_frac"),
            "wait_inst_frac": avg("wait_inst_frac"), "counters_source": src}


def _issue_roofline(config, kernel, kernel_ms_per_step, launches_per_step):
    """The instruction-issue roofline of the dominant kernel: VALU wave-instructions per launch (SQ_INSTS_VALU of the --pmc pass at
    this batch size) x the calibrated issue cycles per instruction of a saturated SIMD (profiles/r06_valu_issue_calibration.json:
    tools/ubench/valu_mix.hip, the 4-waves-per-SIMD rate of the kernel class's own instruction mix) / 1024 SIMDs = the cycles the
    launch needs if every SIMD issued back to back; over the launch's measured duration at the calibration's clock = frac."""
    cj, src = _counters(config)
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "r06_valu_issue_calibration.json")))
    except Exception:
        cal = None
    if not cj or not cal or not kernel:
        return None
    mine = [v for k, v in cj.get("kernels", {}).items() if k.split("(")[0].split("<")[0].split("::")[-1].strip() == kernel and v.get("valu_insts")]
    if not mine:
        return None
    insts = sum(v["valu_insts"] * (v.get("dispatches") or 1) for v in mine) / sum((v.get("dispatches") or 1) for v in mine)   # per launch
    cpi = cal.get("cycles_per_valu_inst", {}).get(kernel, cal.get("cycles_per_valu_inst", {}).get("default"))
    ghz = cal.get("clock_ghz", 2.4)
    if not cpi:
        return None
    floor_ms = insts * cpi / 1024.0 / (ghz * 1e6)
    per_launch_ms = kernel_ms_per_step / max(launches_per_step, 1)
    out = {"valu_wave_insts_per_launch": insts, "cycles_per_inst_saturated": cpi, "clock_ghz": ghz, "floor_ms_per_launch": floor_ms,
           "measured_ms_per_launch": per_launch_ms, "frac": floor_ms / per_launch_ms if per_launch_ms else None,
           "source": f"{src} (SQ_INSTS_VALU) x profiles/r06_valu_issue_calibration.json"}
    # the kernel's OWN instruction stream replayed alone (tools/ubench/stream_replay_gen.py), where it has been: the tight floor
    rcpi = cal.get("replay_cycles_per_valu_inst", {}).get(kernel)
    if rcpi:
        out["replay_cycles_per_inst"] = rcpi
        out["replay_floor_ms_per_launch"] = insts * rcpi / 1024.0 / (ghz * 1e6)
        out["replay_frac"] = out["replay_floor_ms_per_launch"] / per_launch_ms if per_launch_ms else None
    return out


def measured_accuracy(config, host_rows, frame_offsets, pcm, sample_offsets, func_rows=None, n_check=32):
    """The timed run's OWN output against the CPU oracle (oracle/lldo.py: the C restatement pinned bit for bit on the reference
    binary), outside the timed region: the first `n_check` utterances (the corpus tiles 32 distinct ones: all of them) and the
    batch's last one. Config 2 (the fast kernel, its own transform order): max abs error and the per-frame-scaled error
    max_t max_i |d[t,i]| / max_i |ref[t,i]| (gate 1e-5, tests/tolerance.py); configs 3-5 (reference-order kernels): the number
    of cells whose bits equal the oracle's (gate: all)."""
    from oracle import lldo
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import tolerance
    chain = {2: lambda p: lldo.mfcc_chain(lldo.default_cfg(), p), 3: lldo.is09_chain, 4: lldo.compare_lld_chain,
             5: lldo.egemaps_lld_chain}[config]
    n_utt = len(frame_offsets) - 1
    which = sorted(set(list(range(min(n_check, n_utt))) + [n_utt - 1]))
    cells = same = 0
    max_abs = fse = 0.0
    shape_ok = True
    col_max = None                                   # config 2, SURVEY 8(d) (i): max abs error per feature column
    refs, gots = [], []
    for u in which:
        ref = np.asarray(chain(pcm[sample_offsets[u]:sample_offsets[u + 1]]), np.float32)
        got = host_rows(int(frame_offsets[u]), int(frame_offsets[u + 1]))
        if got.shape != ref.shape:
            shape_ok = False
            continue
        eq = (got.view(np.uint32) == ref.view(np.uint32)) | ((got == 0) & (ref == 0))
        cells += eq.size
        same += int(eq.sum())
        if ref.size:
            d = np.abs(got.astype(np.float64) - ref.astype(np.float64))
            d = d[np.isfinite(d)]
            max_abs = max(max_abs, float(d.max()) if d.size else 0.0)
            if config == 2:
                fse = max(fse, tolerance.frame_scaled_err(got, ref, block=13))
                cm = np.abs(got.astype(np.float64) - ref.astype(np.float64)).max(axis=0)
                col_max = cm if col_max is None else np.maximum(col_max, cm)
                refs.append(ref); gots.append(got)
    res = {"utterances_checked": len(which), "cells": cells, "cells_bit_identical": same, "max_abs_err": max_abs,
           "reference": "oracle/lldo.py (C restatement of the reference, pinned bit for bit on oracle/_ref/SMILExtract by "
                        "tests/test_oracle_pin*.py), on the timed run's own output buffer after the timed region",
           "row_counts_equal": shape_ok}
    if config == 2:
        res["frame_scaled_err"] = fse
        if col_max is not None:
            # per-column maxima; the share of cells with |d| <= 1e-5 max(|ref|, s_col), s_col = the
            # column's 99th percentile of |ref| (reported, not gated: the gate is (ii), the per-frame-scaled error)
            R, G = np.concatenate(refs).astype(np.float64), np.concatenate(gots).astype(np.float64)
            s_col = np.percentile(np.abs(R), 99, axis=0)
            ok = np.abs(G - R) <= 1e-5 * np.maximum(np.abs(R), s_col[None, :])
            res["max_abs_err_per_column"] = [float(f"{v:.3g}") for v in col_max]
            res["pass_rate_1e-5_of_max_ref_scol"] = float(ok.mean())
        res["gate"] = "frame_scaled_err <= 1e-5 and row counts =="
        res["pass"] = bool(shape_ok and fse <= 1e-5)
    else:
        res["gate"] = "cells_bit_identical == cells and row counts =="
        res["pass"] = bool(shape_ok and same == cells)
    if func_rows is not None:                        # config 5: the 88 functionals per utterance
        fc = fs = 0
        for u in which:
            ref = np.asarray(lldo.egemaps_func(pcm[sample_offsets[u]:sample_offsets[u + 1]]), np.float32)
            if ref.shape[0] == 0:
                continue
            got = func_rows(u).reshape(1, -1)
            eq = (got.view(np.uint32) == ref.view(np.uint32)) | ((got == 0) & (ref == 0))
            fc += eq.size
            fs += int(eq.sum())
        res["functionals_cells"], res["functionals_bit_identical"] = fc, fs
        res["pass"] = bool(res["pass"] and fc == fs)
    return res


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command under torch.distributed.run, one
    per GPU (LOCAL_RANK picks the device). Fewer devices than ranks: gloo, ranks share devices (the line then says so)."""
    import torch
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if n_dev < args.gpus:
        env.setdefault("SMILEHIP_DIST_BACKEND", "gloo")
        print(f"bench.py: {args.gpus} ranks asked for, {n_dev} device(s) visible: ranks share devices, backend "
              f"{env['SMILEHIP_DIST_BACKEND']} (plumbing check, not a scaling measurement)", file=sys.stderr, flush=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def init_ranks(args):
    """rank, world, the torch.distributed module (or None) -- one rank per GPU; asserts WORLD_SIZE == --gpus"""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}), "
                         f"or run `python bench.py --gpus {args.gpus}` without a launcher")
    if world == 1:
        if not args.plumbing_only:
            torch.cuda.set_device(0)
        return rank, world, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # one rank per GPU over RCCL ("nccl" on ROCm). SMILEHIP_DIST_BACKEND=gloo lets ranks share a GPU (launch_ranks on a
    # box with fewer devices than ranks; tools/smoke_multirank.sh)
    backend = os.environ.get("SMILEHIP_DIST_BACKEND", "nccl")
    if args.plumbing_only:
        dist.init_process_group("gloo")
        return rank, world, dist
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return rank, world, dist


def rank_report(dist, world, plumbing_only=False):
    """what every rank ran on: backend, RCCL version, device index + bus id per rank"""
    import torch
    if plumbing_only:
        me = "cpu"
    else:
        d = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(d)
        me = f"cuda:{d} {getattr(pr, 'pci_bus_id', '?')}:{getattr(pr, 'pci_device_id', '?')} {pr.name}"
    devs = [me]
    if world > 1:
        devs = [None] * world
        dist.all_gather_object(devs, me)
    rep = {"backend": (dist.get_backend() if world > 1 else None), "rank_devices": devs}
    try:
        rep["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        rep["rccl_version"] = None
    if world > 1 and not plumbing_only:
        rep["distinct_devices"] = len(set(devs))
    return rep


def open_comm(dist, world):
    """The product's own gather library (include/smilehip_comm.h) for the path's one collective: an RCCL communicator joined through
    the launcher's process group (rank 0's unique id is broadcast; torch.distributed moves nothing else). -> (Comm or None, text).
    Ranks that share a device (the gloo plumbing fallback of launch_ranks) cannot form an RCCL communicator: torch.distributed P2P then."""
    import torch
    if world > 1 and dist.get_backend() != "nccl":
        return None, f"torch.distributed point-to-point ({dist.get_backend()}: ranks share devices, no RCCL communicator)"
    try:
        from opensmile_amd import comm as smcomm
        cm = smcomm.Comm.from_process_group(dist if world > 1 else None, torch.cuda.current_device())
        return cm, "libsmilehip_comm.so (smilehip_comm_allgather_count + smilehip_comm_gather_rows: one grouped ncclSend / ncclRecv per gather)"
    except Exception as e:
        return None, f"torch.distributed point-to-point (libsmilehip_comm.so failed: {type(e).__name__}: {e})"


def gather_rows(cm, dist, local, stream):
    """every rank's rows to rank 0 (stream-ordered on `stream`): through the library, or torch.distributed when there is no communicator"""
    import torch
    if cm is None:
        from opensmile_amd import gather
        return gather.gather_features(local, dst=0)
    counts = cm.allgather_count(local.shape[0], stream)
    out = torch.empty((int(counts.sum()), local.shape[1]), dtype=torch.float32, device="cuda") if cm.rank == 0 else None
    cm.gather_rows(local.data_ptr() if local.numel() else None, counts, int(local.shape[1]), out.data_ptr() if out is not None else None, stream)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5), help="BASELINE.json config (default 2: the headline)")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)   # config 2: the clocks ramp over the first ~20 launches
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--utts", type=int, default=None, help="utterances per GPU (default: the config's)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the driver's contract): every rank its own 1000 x 10 s; strong: ONE ragged corpus of "
                         "--utts utterances (5..15 s) sharded over the ranks by frame count (gather.shard_utterances)")
    ap.add_argument("--no-configs", action="store_true", help="leave BASELINE configs 3-5 out of the default line")
    ap.add_argument("--no-h2d", action="store_true", help="leave the PCIe-inclusive figures out of the default line")
    ap.add_argument("--with-gather", action="store_true",
                    help="run the gather section at N = 1 too (a world-size-1 communicator: rank 0's own block, the same calls)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="NO kernel is launched and nothing is measured: only the rank plumbing runs (launcher, rendezvous, barrier, "
                         "max-over-ranks, frame sum, gather, JSON) on the gloo backend with CPU tensors -- what the CPU test of the "
                         "N > 1 path uses; the line's metric says so and its value is null")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    if args.plumbing_only:
        return main_plumbing(args)
    if args.config != 2:
        res = run_config(args, args.config, init_ranks(args), with_cpu=not args.no_cpu_baseline)
        if res is not None:
            print(json.dumps(res), flush=True)
        return finish_ranks()
    # N > 1: >= 200 steps, so that the timed region (one barrier on each side) is >= 80 ms of kernels
    args.steps = (200 if args.gpus > 1 else 100) if args.steps is None else args.steps
    args.warmup = 30 if args.warmup is None else args.warmup
    args.utts = N_UTT if args.utts is None else args.utts

    import torch
    from opensmile_amd import capi, synth

    rank, world, dist = init_ranks(args)
    dev = torch.cuda.current_device()

    ctx = capi.Context(dev)
    plan = capi.Plan(ctx)                       # MFCC12_0_D_A
    n_out = plan.geometry.n_out
    if args.scaling == "weak":
        # this rank's own batch: 32 seeded utterances of the corpus contract tiled to --utts (generating 1000 x 10 s in numpy
        # takes minutes; the chain's work per frame does not depend on the data)
        pcm, off = synth.corpus_tiled(args.utts, UTT_SAMPLES, n_unique=32)
    else:
        # ONE corpus for the whole job, ragged (utterance u lasts 5 + (7 u mod 11) s), sharded by frame count: the LPT
        # partition of opensmile_amd/gather.py; this rank packs only its own utterances
        from opensmile_amd import gather
        lens = [(5 + (7 * u) % 11) * 16000 for u in range(args.utts)]
        mine = gather.shard_utterances([plan.num_frames(n) for n in lens], world)[rank]
        base = {n: synth.utterance(2 + (n // 16000), n) for n in sorted(set(lens))}      # one waveform per length
        pcm = np.concatenate([base[lens[u]] for u in mine]) if mine else np.zeros(0, np.int16)
        off = np.concatenate([[0], np.cumsum([lens[u] for u in mine])]).astype(np.int64)
    batch = capi.Batch(plan, off)
    frames = batch.total_frames
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((max(frames, 1), n_out), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        batch.run_device(d_pcm.data_ptr(), d_out.data_ptr(), n_out, stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up, before the W warm-up steps: the device's clocks ramp over the first ~20 launches of a 0.4 ms kernel (a
    # 20-step run right after start-up measured 0.49 ms per step, the steady state is 0.43), so the chain is run untimed
    # for ~0.15 s first -- what a production batch loop is in after its first few batches. Reported in the line.
    ramp_n, r0 = 0, time.perf_counter()
    while ramp_n < 64 or time.perf_counter() - r0 < 0.15:
        step()
        ramp_n += 1
        if ramp_n % 32 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    ramp_ms = (time.perf_counter() - r0) * 1e3
    for _ in range(args.warmup):
        step()
    plan.set_timing(True)                        # HIP events around each launch, same stream
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ms_main, ms_delta = plan.last_timing()
    plan.set_timing(False)

    dts = [dt]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        all_t = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(all_t, t)
        dts = [float(x.item()) for x in all_t]
        dt = max(dts)
        fr = torch.tensor([frames], dtype=torch.int64, device="cuda")
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        total_frames = int(fr.item())
    else:
        total_frames = frames

    # the path's one collective: gather feature matrices to rank 0 (next to `value`, never in it)
    gather_ms, gather_via = None, None
    if world > 1 or args.with_gather:
        cm, gather_via = open_comm(dist, world)
        gather_rows(cm, dist, d_out, stream)         # (first call: communicator set-up, receive buffers)
        barrier()
        g0 = time.perf_counter()
        gathered = gather_rows(cm, dist, d_out, stream)
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        del gathered
        if cm is not None:
            cm.close()
    ranks = rank_report(dist, world)

    if rank == 0:
        value = total_frames * args.steps / dt
        # BASELINE.json's metric: "...; max abs err vs CPU ref" -- measured on the timed run's own output, after the timed region
        try:
            accuracy = measured_accuracy(2, lambda a, b: d_out[a:b].cpu().numpy(), batch.frame_offsets, pcm, off)
        except Exception as e:
            accuracy = {"pass": False, "error": f"{type(e).__name__}: {e}"}
        if not accuracy.get("pass"):
            value = None                          # a fast kernel whose results differ from the reference's is not a measurement
        # the frame kernel computes the two regression stages itself (smilehip_batch_delta_fused): its launch then reads the hop and
        # writes all 39 columns -- 476 B per frame -- and the window-chain kernel is gone from the step
        fused = bool(getattr(batch, "delta_fused", False)) and ms_delta < 0.2 * ms_main
        alg_per_frame = ALG_BYTES_PER_FRAME_CHAIN if fused else ALG_BYTES_PER_FRAME_MAIN
        alg_bytes = alg_per_frame * frames
        achieved = alg_bytes / (ms_main * 1e-3) / 1e9
        # HBM bytes per launch of the dominant kernel cannot be counted inside this process: the figure is the one the
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS command produced (tools/profile_run.sh), kept in profiles/
        traffic, traffic_source = None, None
        tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tfile) and args.utts == N_UTT and args.scaling == "weak":
            try:
                tj = json.load(open(tfile))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "profiles/pmc_traffic.json: " + tj.get("source", "rocprofv3 --pmc passes of this command")
            except Exception:
                traffic = None
        res = {
            "metric": "MFCC frames/sec (16kHz, 25ms/10ms)", "value": value, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MFCC12_0_D_A (13 MFCC + delta + accel) on 1000 x 10 s synthetic 16 kHz mono "
                                   "int16 per GPU, 25 ms / 10 ms, PCM resident in HBM",
                       "utterances_per_gpu": args.utts if args.scaling == "weak" else None,
                       "utterances_total": args.utts * world if args.scaling == "weak" else args.utts,
                       "frames_rank0": frames, "out_cols": n_out,
                       "ranks_seen": world, "rank_ms_per_step": {"min": min(dts) / args.steps * 1e3, "max": max(dts) / args.steps * 1e3},
                       "corpus": ("32 seeded utterances of the SURVEY 8(d) contract tiled to 1000 per GPU (work per frame is "
                                  "data-independent)") if args.scaling == "weak" else
                                 "one ragged corpus (5..15 s utterances), LPT-sharded by frame count over the ranks",
                       "clock_ramp": {"untimed_launches_before_warmup": ramp_n, "ms": ramp_ms,
                                      "note": "set-up: the chain run untimed for >= 0.15 s so that the device clocks are at their "
                                              "steady state when the W warm-up steps begin; the timed region is exactly K steps"},
                       "parallelism": f"utterance-sharded x{world}"},
            # The roof that bounds this kernel is FP32 vector issue (+ the per-CU LDS pipe), not HBM -- arithmetic intensity
            # 41 FLOP/B against a ridge of 20, and the counters agree (profiles/) -- so `bound`,
            # `achieved`, `peak`, `frac` are the FP32 figures and the HBM figures the north star asks for stand beside them.
            "roofline": {"bound": "fp32_valu", "kernel": "fused MFCC + delta regression (R0-R7, R13)" if fused else "fused MFCC (R0-R7)",
                         "achieved": ALG_FLOP_PER_FRAME * frames / (ms_main * 1e-3) / 1e12,
                         "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (ALG_FLOP_PER_FRAME * frames / (ms_main * 1e-3)) / (FP32_PEAK_TFLOPS * 1e12),
                         "alg_flop_per_frame": ALG_FLOP_PER_FRAME,
                         "hbm_achieved": achieved, "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s", "hbm_frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,      # from profiles/ (a separate --pmc pass), not this run
                         **_counter_fields(2, ["lld_mfcc512"]),
                         "alg_bytes_per_frame": alg_per_frame,
                         "alg_flop_note": "SURVEY 8d's count for R0-R7; the two regression stages add ~2e2 per frame, not counted",
                         "kernel_ms": ms_main, "delta_kernel_ms": ms_delta},
            "accuracy": accuracy, "max_abs_err": accuracy.get("max_abs_err"), "frame_scaled_err": accuracy.get("frame_scaled_err"),
            "ranks": ranks,
        }
        if gather_ms is not None:
            res["gather_ms"] = gather_ms
            res["value_incl_gather"] = total_frames * args.steps / (dt + args.steps * gather_ms * 1e-3)
            res["gather_note"] = "every step's feature matrices (frames x 39 f32 per rank) gathered to rank 0, not overlapped"
            res["gather_via"] = gather_via
        if world == 1 and not args.no_h2d:
            try:
                del d_out, d_pcm
                res["h2d_inclusive"], res["to_host"] = pcie_figures()
            except Exception as e:
                res["h2d_inclusive"] = {"value": None, "note": f"failed: {e}"}
            try:
                res["file_to_file"] = file_to_file_figure()
            except Exception as e:
                res["file_to_file"] = {"value": None, "note": f"failed: {type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline must never take the bench line down
                res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference",
                                       "sample": f"failed: {e}"}
        if world == 1 and not args.no_configs:
            # BASELINE.json's configs 3-5 at their per-GPU sizes, the same protocol, their own short step counts
            res["configs"] = {}
            sub = argparse.Namespace(**vars(args))
            sub.steps = sub.warmup = sub.utts = None
            for k in (3, 4, 5):
                try:
                    torch.cuda.empty_cache()
                    res["configs"][str(k)] = run_config(sub, k, (rank, world, dist), with_cpu=not args.no_cpu_baseline, nested=True)
                except Exception as e:
                    res["configs"][str(k)] = {"value": None, "error": str(e)}
        print(json.dumps(res), flush=True)
    finish_ranks()


def finish_ranks():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def pcie_figures(utts=N_UTT, chunks=10, reps=5):
    """SURVEY 8(d)'s other two figures for the headline workload: PCM starting in pinned host memory (copy of chunk i + 1
    overlapped with the chain on chunk i), and the same with the features copied back to pinned host memory."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_h2d", os.path.join(ROOT, "tools", "bench_h2d.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = mod.measure(utts, chunks, reps, d2h=False)
    b = mod.measure(utts, chunks, reps, d2h=True)
    fmt = lambda r: {"value": r["frames_per_s"], "unit": "frames/s", "ms": r["ms"], "h2d_GBps": r["h2d_GBps"], "workload": r["workload"]}
    return fmt(a), fmt(b)


def file_to_file_figure(files=8000, reps=3):
    """The route a user runs: `files` x 10 s 16-bit mono WAV files on /dev/shm -> opensmile_amd/smilextract_hip --set
    mfcc12_0_d_a (one process, one GPU: header walks, page-locked staging, copies, kernels, one HTK file per input) -- wall clock of
    the whole process (start-up included) and from the first ingest to the last sink (SMILEHIP_TIMING); fastest of `reps` runs."""
    import re
    import shutil
    from opensmile_amd import synth
    from oracle import lldo
    d = "/dev/shm/smilehip_f2f" if os.path.isdir("/dev/shm") else tempfile.mkdtemp()
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d + "/in"); os.makedirs(d + "/out")
    try:
        uniq = [synth.utterance(2 + i, UTT_SAMPLES) for i in range(32)]
        first = f"{d}/in/u00000.wav"
        paths = []
        for i in range(files):
            p = f"{d}/in/u{i:05d}.wav"
            if i < 32:
                lldo.write_wav(p, uniq[i])
            else:
                shutil.copyfile(f"{d}/in/u{i % 32:05d}.wav", p)
            paths.append(p)
        open(d + "/list.txt", "w").write("\n".join(paths) + "\n")
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "opensmile_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""), SMILEHIP_TIMING="1")
        cmd = [os.path.join(ROOT, "opensmile_amd", "smilextract_hip"), "--set", "mfcc12_0_d_a", "-filelist", d + "/list.txt", "-outdir", d + "/out", "-O", "1"]
        subprocess.run(cmd, env=env, capture_output=True, text=True)              # (first run: page cache, code objects)
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            r = subprocess.run(cmd, env=env, capture_output=True, text=True)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"value": None, "note": "smilextract_hip failed: " + r.stderr[-300:]}
            if best is None or dt < best[0]:
                best = (dt, r.stderr)
        wall, err = best
        m = re.search(r"since the first ingest ([0-9.]+) s", err)
        net = float(m.group(1)) if m else None
        n_out = sum(1 for f in os.listdir(d + "/out") if f.endswith(".htk"))
        frames = files * 998
        # the written files against the oracle (the fast kernel: per-frame-scaled error), a sample of them
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import tolerance
        fse = 0.0
        for i in range(0, files, max(1, files // 16)):
            got = lldo.read_htk(f"{d}/out/u{i:05d}.htk")[0]
            ref = lldo.mfcc_chain(lldo.default_cfg(), uniq[i % 32])
            fse = max(fse, tolerance.frame_scaled_err(got, ref, block=13) if got.shape == ref.shape else 1.0)
        return {"value": frames / wall, "unit": "frames/s", "wall_s": wall, "value_net_of_startup": frames / net if net else None,
                "net_s": net, "files": files, "files_written": n_out, "frame_scaled_err_of_written_files": fse,
                "stages": (err.strip().splitlines() or [""])[-1],
                "workload": f"{files} x 10 s WAV files on /dev/shm -> smilextract_hip --set mfcc12_0_d_a -> {files} HTK files on /dev/shm, one "
                            "process, start-up included in `value`"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main_plumbing(args):
    """--plumbing-only: every step of the N-rank protocol except the device work (see the flag's help)."""
    import torch
    rank, world, dist = init_ranks(args)
    steps = 3 if args.steps is None else args.steps
    frames = 998 * (10 + rank)                       # ragged on purpose: the gather is a gather-v
    local = torch.full((frames, 39), float(rank))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        local += 1.0
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dts, total = [dt], frames
    gathered, pieces = None, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        all_t = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(all_t, t)
        dts = [float(x.item()) for x in all_t]
        fr = torch.tensor([frames], dtype=torch.int64)
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        total = int(fr.item())
        from opensmile_amd import gather
        out = gather.gather_features(local, dst=0)
        if rank == 0:
            gathered = [int(o.shape[0]) for o in out]
            assert all(float(o[0, 0]) == r + steps for r, o in enumerate(out))
        # ... and in pieces: the call order of config 4's gather (opensmile_amd/comm.py: PieceGather; on devices the pieces travel
        # through libsmilehip_comm.so, here through gloo -- the rows of each piece are the library's own arithmetic either way)
        from opensmile_amd import comm as smcomm
        pg = smcomm.PieceGather(local, dist, piece_rows=4096)
        for k in range(pg.pieces):
            pg.piece(k)
        whole = pg.finish()
        if rank == 0:
            pieces = {"pieces": pg.pieces, "rows": int(whole.shape[0]),
                      "equal": bool(torch.equal(whole, torch.cat(out)))}
    ranks = rank_report(dist, world, plumbing_only=True)
    if rank == 0:
        print(json.dumps({"metric": "PLUMBING ONLY -- no kernel launched, nothing measured", "value": None, "unit": "frames/s",
                          "n_gpus": world, "steps": steps, "warmup": 0, "ms_per_step": None, "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": None, "data": "none",
                          "config": {"workload": "rank plumbing of bench.py on CPU tensors (gloo)", "ranks_seen": len(dts),
                                     "frames_total": total, "gathered_rows": gathered, "gathered_in_pieces": pieces},
                          "ranks": ranks}), flush=True)
    finish_ranks()


def run_config(args, config, ranks, with_cpu=True, nested=False):
    """config 3 | 4 | 5: the same protocol (warm-up, K timed steps between barrier + synchronize, max over ranks) on that
    config's chain through smilehip_lld_run; weak scaling, every rank its own batch. Returns rank 0's result object."""
    import ctypes as C

    import torch
    from opensmile_amd import capi, synth
    c = CONFIGS[config]
    steps = c["steps"] if args.steps is None else args.steps
    warmup = c["warmup"] if args.warmup is None else args.warmup
    utts = c["utts"] if args.utts is None else args.utts
    rank, world, dist = ranks
    ctx = capi.Context(torch.cuda.current_device())
    plan = capi.Plan(ctx, getattr(capi, c["cfg"])())
    n_out = plan.geometry.n_out
    pcm, off = synth.corpus_tiled(utts, c["samples"], n_unique=32)
    batch = capi.Batch(plan, off)
    frames, rows = batch.total_frames, int(batch.frame_offsets[-1])
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.empty((max(rows, 1), n_out), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    L = capi.load()
    d_func = torch.empty((utts, 88), dtype=torch.float32, device="cuda") if config == 5 else None

    def step():
        batch.run_device(d_pcm.data_ptr(), d_out.data_ptr(), n_out, stream)
        if config == 5:     # the functionals level on the smoothed levels the run left in the batch's scratch
            capi._check(L.smilehip_batch_functionals_egemaps(plan._h, batch._h, C.c_void_p(d_func.data_ptr()), 88, C.c_void_p(stream)))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    plan.set_timing(True)
    capi.kernel_timing(True)                     # two HIP events around EVERY launch of the step, on the launch's own stream
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ms_main, ms_rest = plan.last_timing()
    by_kernel = capi.kernel_timing_report()      # {name: (launches, summed ms)} over the timed region
    capi.kernel_timing(False)
    plan.set_timing(False)
    dts = [dt]
    total_frames = frames
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        all_t = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(all_t, t)
        dts = [float(x.item()) for x in all_t]
        dt = max(dts)
        fr = torch.tensor([frames], dtype=torch.int64, device="cuda")
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        total_frames = int(fr.item())
    # the path's one collective: what a caller wants in one place goes to rank 0, through the product's own library
    # (include/smilehip_comm.h) -- config 5's 88 functionals per utterance (44 MB per rank) and config 3's LLD matrix in one grouped
    # send / receive; config 4's LLD level (6.5 GB per rank) in pieces on the communicator's own stream BESIDE the next step's
    # kernels (two output buffers). Next to `value`, never in it.
    gather_ms, gathered, gather_via, overlap = None, None, None, None
    if world > 1 or args.with_gather:
        cm, gather_via = open_comm(dist, world)
        if config in (3, 5) or cm is None:
            g_src = d_func if config == 5 else d_out
            gather_rows(cm, dist, g_src, stream)             # (first call: communicator set-up, receive buffers)
            barrier()
            g0 = time.perf_counter()
            got = gather_rows(cm, dist, g_src, stream)
            barrier()
            gather_ms = (time.perf_counter() - g0) * 1e3
            gathered = "88 functionals per utterance" if config == 5 else f"LLD matrix (rows x {n_out})"
            if rank == 0 and cm is not None:
                overlap = {"rank0_block_equal": bool(torch.equal(got[:g_src.shape[0]], g_src))}
            del got
        else:
            from opensmile_amd import comm as smcomm
            piece_rows = 1 << 18                             # 136 MB of 130-column rows per rank and piece
            bufs = [d_out, torch.empty_like(d_out)]
            pg = smcomm.PieceGather(bufs[0], cm, piece_rows=piece_rows)      # counts exchanged once; rank 0 holds the gathered level
            def issue(buf):
                pg.rebind(buf)
                for k in range(pg.pieces):
                    pg.piece(k, after_stream=stream)
            barrier()
            g0 = time.perf_counter()
            for i in range(steps):
                cm.wait(stream)                              # the gathers issued so far (steps <= i - 2) are done before step i overwrites their buffer
                if i > 0:
                    issue(bufs[(i - 1) % 2])                 # step i - 1's rows travel while step i's kernels run
                batch.run_device(d_pcm.data_ptr(), bufs[i % 2].data_ptr(), n_out, stream)
            issue(bufs[(steps - 1) % 2])
            cm.wait(stream)
            barrier()
            dt_ov = time.perf_counter() - g0
            dts_ov = [dt_ov]
            if world > 1:
                t = torch.tensor([dt_ov], dtype=torch.float64, device="cuda")
                all_t = [torch.zeros_like(t) for _ in range(world)]
                dist.all_gather(all_t, t)
                dts_ov = [float(x.item()) for x in all_t]
            gathered = f"LLD level (rows x {n_out}) of every step, in pieces of {piece_rows} rows per rank on the communicator's stream beside the next step's kernels"
            overlap = {"ms_per_step_with_overlapped_gather": max(dts_ov) / steps * 1e3, "pieces_per_step": pg.pieces, "piece_rows": piece_rows,
                       "value_incl_gather_overlapped": total_frames * steps / max(dts_ov)}
            if rank == 0:
                last = bufs[(steps - 1) % 2]
                overlap["rank0_block_equal"] = bool(torch.equal(pg.out[:last.shape[0]], last))
            del pg, bufs
        if cm is not None:
            cm.close()
    if rank == 0:
        # the dominant kernel = the largest share of the step's summed kernel time (kernels on side streams overlap: the shares
        # are of the SUM, which exceeds the step's wall time where streams run beside each other)
        k_sum = sum(v[1] for v in by_kernel.values()) or 1.0
        dom = max(by_kernel, key=lambda k: by_kernel[k][1]) if by_kernel else None
        dom_ms = by_kernel[dom][1] / steps if dom else ms_main           # per step (a kernel launched per chunk: all its launches)
        dom_alg, dom_note = KERNEL_ALG_BYTES.get(dom, (None, None)) if dom else (None, None)
        if dom_alg is None:                                              # not in the table: the whole chain's bytes
            dom_alg, dom_note = CHAIN_ALG_BYTES[config], dom_note or "the whole chain's bytes per frame (SURVEY 8d): int16 hop in, every output column out"
        achieved = dom_alg * frames / (dom_ms * 1e-3) / 1e9
        achieved_step = CHAIN_ALG_BYTES[config] * frames / (dt / steps) / 1e9
        cf = _counter_fields(config, [dom] if dom else c["pmc_kernels"])
        try:
            accuracy = measured_accuracy(config, lambda a, b: d_out[a:b].cpu().numpy(), batch.frame_offsets, pcm, off,
                                         func_rows=(lambda u: d_func[u].cpu().numpy()) if config == 5 else None)
        except Exception as e:
            accuracy = {"pass": False, "error": f"{type(e).__name__}: {e}"}
        res = {
            "metric": {3: "IS09_emotion LLD frames/sec (16kHz, 25ms/10ms)", 4: "ComParE_2016 LLD frames/sec (16kHz, 20ms+60ms/10ms)",
                       5: "eGeMAPSv02 LLD+functionals frames/sec (16kHz, 20ms+60ms/10ms)"}[config],
            "value": (total_frames * steps / dt) if accuracy.get("pass") else None, "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "accuracy": accuracy, "max_abs_err": accuracy.get("max_abs_err"), "cells_bit_identical": accuracy.get("cells_bit_identical"),
            "cells_checked": accuracy.get("cells"),
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": c["workload"], "baseline_config": config, "utterances_per_gpu": utts,
                       "utterances_total": utts * world, "utterances_per_s": utts * world * steps / dt,
                       "frames_rank0": frames, "rows_rank0": rows, "out_cols": n_out,
                       "ranks_seen": world, "rank_ms_per_step": {"min": min(dts) / steps * 1e3, "max": max(dts) / steps * 1e3},
                       "corpus": "32 seeded utterances of the SURVEY 8(d) contract tiled (work per frame is data-independent except "
                                 "for the voiced / unvoiced pattern, which the 32 cover)",
                       "parallelism": f"utterance-sharded x{world}"},
            # `bound` is what the counters of the dominant kernel say (profiles/r06_pmc_c<N>.json, collected at this batch size by
            # separate --pmc passes), never a literal; the HBM figures (achieved / peak / frac) are the contract's ruler and stand
            # beside the instruction-issue figures below, which are the ruler that fits these kernels
            "roofline": {"bound": cf.get("bound_by_counters") or "unknown (no counter file for this kernel)", **cf,
                         "kernel": dom, "kernel_share_of_summed_kernel_time": (by_kernel[dom][1] / k_sum) if dom else None,
                         "kernel_launches_per_step": (by_kernel[dom][0] / steps) if dom else None,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_step": achieved_step / HBM_PEAK_GBS, "achieved_step": achieved_step,
                         "alg_bytes_per_frame_step": CHAIN_ALG_BYTES[config],
                         "traffic": None, "traffic_source": None,
                         "alg_bytes_per_frame": dom_alg, "alg_bytes_note": dom_note, "kernel_ms": dom_ms,
                         "frame_kernels": c["kernel"], "frame_kernels_ms": ms_main, "rest_of_step_ms": ms_rest,
                         "issue": _issue_roofline(config, dom, dom_ms, by_kernel[dom][0] / steps if dom else 1),
                         "kernels_ms_per_step": {k: round(v[1] / steps, 4) for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])}},
        }
        if gather_via is not None:
            res["gathered"], res["gather_via"] = gathered, gather_via
        if gather_ms is not None:
            res["gather_ms"] = gather_ms
            res["value_incl_gather"] = total_frames * steps / (dt + steps * gather_ms * 1e-3)
        if overlap is not None:
            res["gather_check"] = overlap
        # HBM bytes of the roofline kernel(s) per launch: counted by separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, each
        # alone) of this command at a smaller batch, kept per frame in profiles/ and scaled to this batch's frames
        tfile = os.path.join(ROOT, "profiles", f"pmc_traffic_c{config}.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                by = tj.get("by_kernel", {})
                mine = [v for k, v in by.items() if k.split("<")[0] == dom]
                # the roofline kernel(s)' own HBM bytes per launch (what `achieved` is to be held against), and the whole step's
                res["roofline"]["traffic"] = sum(v["fetch_bytes_per_frame"] + v["write_bytes_per_frame"] for v in mine) * frames if mine else None
                res["roofline"]["traffic_step"] = tj["hbm_bytes_per_frame"] * frames
                res["roofline"]["traffic_source"] = f"profiles/pmc_traffic_c{config}.json: " + tj.get("source", "")
                res["roofline"]["traffic_by_kernel_bytes_per_frame"] = {
                    k: round(v["fetch_bytes_per_frame"] + v["write_bytes_per_frame"], 1) for k, v in by.items()}
            except Exception:
                pass
        if world == 1 and with_cpu:
            try:
                fpf = plan.num_frames(c["samples"])
                res["cpu_baseline"] = cpu_baseline(conf_rel=c["conf"], opt=c["opt"], n_samples=c["samples"], frames_per_file=fpf,
                                                   max_seconds=12.0 if nested else 25.0, max_files=3000 if nested else 14000)
            except Exception as e:
                res["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "reference", "sample": f"failed: {e}"}
        return res
    return None


if __name__ == "__main__":
    main()
