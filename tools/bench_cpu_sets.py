#!/usr/bin/env python3
"""CPU reference throughput of the other feature sets (not the bench line): the real SMILExtract binary of oracle/_ref,
one process per 10 s file on all host cores (xargs -P), output to /dev/shm -- the protocol of bench.py's cpu_baseline with
another config and output option. Prints one JSON object per set."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SETS = {  # name: (config, output option, frames per file[, file length in samples])
    "is09": ("is09-13/IS09_emotion.conf", "-lldhtkoutput", 998),
    "compare": ("compare16/ComParE_2016.conf", "-lldhtkoutput", 995),
    "plp": ("plp/PLP_0_D_A.conf", "-O", 998),
    "mfcc_e_z": ("mfcc/MFCC12_E_D_A_Z.conf", "-O", 998),
    # BASELINE config 5: eGeMAPSv02 on 3 s utterances, functionals (88 values) as the output; 299 frames of the 20 ms framer
    "egemaps": ("egemaps/v02/eGeMAPSv02.conf", "-htkoutput", 299, 48000),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", default="is09,compare")
    ap.add_argument("--seconds", type=float, default=12.0)
    args = ap.parse_args()
    from oracle import lldo
    from opensmile_amd import synth
    exe = os.path.join(lldo.REF_DIR, "SMILExtract")
    if not os.path.exists(exe):
        sys.exit("oracle/_ref/SMILExtract not built")
    cores = os.cpu_count() or 1
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as td:
        n_unique = 8
        for name in args.sets.split(","):
            conf_rel, opt, frames = SETS[name][:3]
            n_samp = SETS[name][3] if len(SETS[name]) > 3 else 160000
            for i in range(n_unique):
                lldo.write_wav(os.path.join(td, f"u{i}.wav"), synth.utterance(2 + i, n_samp))
            conf = os.path.join(lldo.REF_DIR, "config", conf_rel)
            t0 = time.perf_counter()
            n_cal = 4
            for i in range(n_cal):
                subprocess.run([exe, "-C", conf, "-I", os.path.join(td, f"u{i}.wav"), opt, os.path.join(td, "cal.htk"), "-l", "0"],
                               cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            per_file = (time.perf_counter() - t0) / n_cal
            n_files = int(max(cores * 2, min(14000, args.seconds / per_file * cores)))
            jobs = "\n".join(f"{i % n_unique} {i % (4 * cores)}" for i in range(n_files))
            cmd = f"xargs -P {cores} -L 1 sh -c '{exe} -C {conf} -I {td}/u$0.wav {opt} {td}/o$1.htk -l 0 >/dev/null 2>&1'"
            t0 = time.perf_counter()
            subprocess.run(cmd, shell=True, input=jobs.encode(), cwd=td, check=True)
            dt = time.perf_counter() - t0
            print(json.dumps({"set": name, "config": conf_rel, "cores": cores, "files": n_files, "file_seconds": n_samp / 16000.0, "wall_s": dt,
                              "frames_per_s": n_files * frames / dt, "one_core_frames_per_s": frames / per_file}), flush=True)


if __name__ == "__main__":
    main()
