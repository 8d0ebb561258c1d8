#!/usr/bin/env python3
"""Throughput of the in-process route (not a bench line): a host program pushes a long signal into a running openSMILE instance through
the reference's C API -- smile_extaudiosource_write_data in 1 s pieces (cExternalAudioSource::writeData,
src/iocore/externalAudioSource.cpp:132-160), vectors back through cExternalSink's callback -- once with the plain library
(oracle/_ref/libSMILEapi.so, no ./plugins in the working directory) and once with the plugin loaded (block-per-tick overrides: an
input that never ends cannot take the whole-file batch). Graphs: MFCC12_0_D_A's chain (tests/conf/mfcc_smileapi.conf) and the LLD
part of config/is09-13/IS09_emotion.conf with its wave source / file sinks replaced by the external source / sink (text substitution
at run time). Prints one JSON object per (graph, mode): seconds from the first write to smile_run's return, frames per second, and
whether the plugin's vectors equal the plain library's (IS09: bit for bit; MFCC: per-frame-scaled error)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

EXT_SOURCE = """[componentInstances:cComponentManager]
instance[extsource].type=cExternalAudioSource
[extsource:cExternalAudioSource]
writer.dmLevel=wave
sampleRate=16000
channels=1
nBits=16
blocksize_sec=1.0
buffersize_sec=5.0
fieldName=pcm
"""
EXT_SINK = """[componentInstances:cComponentManager]
instance[extsink].type=cExternalSink
[extsink:cExternalSink]
reader.dmLevel=lld;lld_de
"""


def is09_conf(ref_dir, td):
    base = os.path.join(ref_dir, "config", "is09-13")
    lines = open(os.path.join(base, "IS09_emotion.conf")).read().split("\n")
    out = []
    for l in lines:
        s = l.strip()
        if s == "\\{../shared/standard_wave_input.conf.inc}":
            out.append(EXT_SOURCE)
        elif s == "\\{../shared/standard_data_output.conf.inc}":
            out.append(EXT_SINK)
        elif s.startswith("\\{") and s.endswith("}"):
            out.append("\\{" + os.path.join(base, s[2:-1]) + "}")
        else:
            out.append(l)
    p = os.path.join(td, "is09_smileapi.conf")
    open(p, "w").write("\n".join(out) + "\n")
    return p


def mfcc_conf(td):
    txt = open(os.path.join(ROOT, "tests", "conf", "mfcc_smileapi.conf")).read().replace("blocksize_sec=0.1", "blocksize_sec=1.0\nbuffersize_sec=5.0")
    p = os.path.join(td, "mfcc_smileapi.conf")
    open(p, "w").write(txt)
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--block", type=int, default=16000, help="samples per smile_extaudiosource_write_data call")
    args = ap.parse_args()
    from oracle import lldo
    from opensmile_amd import synth
    lib = os.path.join(lldo.REF_DIR, "libSMILEapi.so")
    plugdir = os.path.join(ROOT, "opensmile_amd", "plugin")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), lldo.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        pcm = synth.utterance(5, int(args.seconds * 16000))
        pin = os.path.join(td, "pcm.npy")
        np.save(pin, pcm)
        for name, conf in (("MFCC12_0_D_A chain", mfcc_conf(td)), ("IS09_emotion lld", is09_conf(lldo.REF_DIR, td))):
            res = {}
            for mode, cwd, envx in (("plain_library", td, {}), ("plugin_block_per_tick", plugdir, {})):
                e = dict(env)
                e.update(envx)
                pout = os.path.join(td, mode + ".npy")
                best = None
                for _ in range(2):
                    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "smileapi_run.py"), lib, conf, pin, pout,
                                        "--block", str(args.block), "--timing"], cwd=cwd, env=e, capture_output=True, text=True, timeout=1800)
                    if r.returncode != 0:
                        print(json.dumps({"graph": name, "mode": mode, "error": (r.stdout + r.stderr)[-400:]}), flush=True)
                        best = None
                        break
                    t = json.loads(r.stdout.strip().split("\n")[-1])
                    if best is None or t["feed_to_end_s"] < best["feed_to_end_s"]:
                        best = t
                if best is None:
                    continue
                res[mode] = (best, np.load(pout))
                print(json.dumps({"graph": name, "mode": mode, "signal_seconds": args.seconds, "block_samples": args.block, "vectors": best["vectors"],
                                  "feed_to_end_s": round(best["feed_to_end_s"], 4), "init_s": round(best["init_s"], 4),
                                  "frames_per_s": round(best["vectors"] / best["feed_to_end_s"], 1)}), flush=True)
            if len(res) == 2:
                a, b = res["plain_library"][1], res["plugin_block_per_tick"][1]
                same_shape = a.shape == b.shape
                rec = {"graph": name, "same_shape": same_shape}
                if same_shape and a.size:
                    rec["bit_identical"] = bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
                    scale = np.maximum(np.abs(a).max(axis=1, keepdims=True), 1e-30)
                    rec["frame_scaled_err"] = float((np.abs(a.astype(np.float64) - b) / scale).max())
                    rec["speedup"] = round(res["plain_library"][0]["feed_to_end_s"] / res["plugin_block_per_tick"][0]["feed_to_end_s"], 3)
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
