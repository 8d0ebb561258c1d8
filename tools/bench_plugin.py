#!/usr/bin/env python3
"""Throughput of the plugin path inside the unmodified reference binary, on one file (not a bench line).
Three runs of oracle/_ref/SMILExtract on the same wav: (a) plain CPU binary (plugin disabled), (b) the per-component
overrides (one upload / launch / download per frame per component), (c) the same unmodified file with SMILEHIP_PLUGIN_FUSE=1
(whole file in one batch, the chain's components hand out its rows), (d) the fused source component cHipLldSource.
Process wall time minus the time of an empty-input run of the same mode (start-up: config parsing, dlopen, HIP init) is
reported as well. Prints one JSON object per (config, mode)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--skip-per-component", action="store_true", help="per-component mode only for MFCC and ComParE (it takes ~25 s per run)")
    ap.add_argument("--configs", default="", help="comma-separated substrings of the config names to run (default: all)")
    ap.add_argument("--no-per-component", action="store_true", help="leave the per-component mode out altogether (an hour-long file takes minutes in it)")
    args = ap.parse_args()
    from oracle import lldo
    from opensmile_amd import synth
    exe = os.path.join(lldo.REF_DIR, "SMILExtract")
    plugdir = os.path.join(ROOT, "opensmile_amd", "plugin")
    env0 = dict(os.environ)
    env0.pop("SMILEHIP_PLUGIN_FUSE", None)
    env0["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), lldo.REF_DIR, env0.get("LD_LIBRARY_PATH", "")])
    n = int(args.seconds * 16000)
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        wav, tiny = os.path.join(td, "in.wav"), os.path.join(td, "tiny.wav")
        lldo.write_wav(wav, synth.utterance(5, n))
        lldo.write_wav(tiny, synth.utterance(5, 1600))
        cases = [
            ("MFCC12_0_D_A", os.path.join(lldo.REF_DIR, "config", "mfcc/MFCC12_0_D_A.conf"), "-O", [], int(args.seconds * 100) - 2),
            ("ComParE_2016 lld", os.path.join(lldo.REF_DIR, "config", "compare16/ComParE_2016.conf"), "-lldhtkoutput", [], int(args.seconds * 100) - 5),
            ("IS09_emotion lld", os.path.join(lldo.REF_DIR, "config", "is09-13/IS09_emotion.conf"), "-lldhtkoutput", [], int(args.seconds * 100) - 2),
            ("eGeMAPSv02 lld", os.path.join(lldo.REF_DIR, "config", "egemaps/v02/eGeMAPSv02.conf"), "-lldhtkoutput", [], int(args.seconds * 100) - 5),
        ]
        fused_sets = {"ComParE_2016 lld": "compare16_lld", "IS09_emotion lld": "is09_lld", "eGeMAPSv02 lld": "egemapsv02_lld"}
        for name, conf, opt, extra, frames in cases:
            if args.configs and not any(c in name for c in args.configs.split(",")):
                continue
            modes = [("cpu_binary", {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, conf, extra),
                     ("plugin_per_component", {"SMILEHIP_PLUGIN_FUSE": "0", "SMILEHIP_PLUGIN_BLOCK": "0"}, conf, extra),
                     ("plugin_block_per_tick", {"SMILEHIP_PLUGIN_FUSE": "0"}, conf, extra)]
            modes.append(("plugin_default_unmodified_conf (fused)", {}, conf, extra))       # (round 5: fused is what an unmodified file gets)
            if name == "MFCC12_0_D_A":
                modes.append(("plugin_fused_source", {}, os.path.join(plugdir, "conf", "MFCC12_0_D_A_hip.conf"), ["-featureSet", "mfcc12_0_d_a"]))
            if name in fused_sets:                          # the whole LLD level from ONE source component, the reference's sinks
                modes.append(("plugin_fused_source", {}, os.path.join(plugdir, "conf", "LLD_hip.conf"), ["-featureSet", fused_sets[name]]))
                if args.skip_per_component and name != "ComParE_2016 lld":
                    modes = [m for m in modes if m[0] != "plugin_per_component"]
            if args.no_per_component:
                modes = [m for m in modes if m[0] != "plugin_per_component"]
            for mode, envx, c, ex in modes:
                env = dict(env0)
                env.update(envx)
                res = {}
                for tag, f in (("startup", tiny), ("file", wav)):
                    best = 1e9
                    for _ in range(2):
                        t0 = time.perf_counter()
                        r = subprocess.run([exe, "-C", c, "-I", f, opt, os.path.join(td, "o.htk"), "-l", "0"] + ex, cwd=plugdir, env=env,
                                           capture_output=True, text=True)
                        best = min(best, time.perf_counter() - t0)
                        if r.returncode != 0:
                            print(json.dumps({"config": name, "mode": mode, "error": r.stderr[-300:]}), flush=True)
                            break
                    res[tag] = best
                net = max(res["file"] - res["startup"], 1e-9)
                print(json.dumps({"config": name, "mode": mode, "file_seconds": args.seconds, "frames": frames, "wall_s": round(res["file"], 4),
                                  "startup_s": round(res["startup"], 4), "frames_per_s_wall": round(frames / res["file"], 1),
                                  "frames_per_s_net": round(frames / net, 1)}), flush=True)


if __name__ == "__main__":
    main()
