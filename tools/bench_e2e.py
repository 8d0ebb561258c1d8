#!/usr/bin/env python3
"""End to end, files in -> files out, beside the reference on the same list:

    N x 10 s (3 s for eGeMAPS) 16-bit mono WAV files on /dev/shm
      -> opensmile_amd/smilextract_hip --set <set> -filelist ... (one process, one GPU; HTK files per input / one ARFF)
      -> xargs -P $(nproc) oracle/_ref/SMILExtract -C <conf> -I f ... (one process per file, all host cores)

Wall clock of both, files per second, and -- for the sets whose chain is the reference's bit for bit -- a byte comparison
of the outputs. One JSON object per set on stdout (profiles/r03_e2e.jsonl)."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SETS = {
    # name: (smilextract_hip --set, reference conf, seconds per file, hip output options, reference output option, per-file ext)
    "mfcc12_0_d_a": ("mfcc12_0_d_a", "mfcc/MFCC12_0_D_A.conf", 10.0, ["-O", "1"], "-O", ".htk"),
    "is09_emotion": ("is09_emotion", "is09-13/IS09_emotion.conf", 10.0, ["-lldhtkoutput", "1"], "-lldhtkoutput", ".lld.htk"),
    "egemapsv02": ("egemapsv02", "egemaps/v02/eGeMAPSv02.conf", 3.0, ["-lldhtkoutput", "1"], "-lldhtkoutput", ".lld.htk"),
    "compare16": ("compare16_lld", "compare16/ComParE_2016.conf", 10.0, ["-lldhtkoutput", "1"], "-lldhtkoutput", ".lld.htk"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=1000)
    ap.add_argument("--sets", default="mfcc12_0_d_a,egemapsv02")
    ap.add_argument("--ref-files", type=int, default=0, help="files the reference runs on (0 = all); its rate is per file anyway")
    ap.add_argument("--dir", default="/dev/shm/smilehip_e2e")
    ap.add_argument("--reps", type=int, default=3, help="timed runs of smilextract_hip (the fastest is reported)")
    args = ap.parse_args()
    from opensmile_amd import synth
    from oracle import lldo
    exe = os.path.join(lldo.REF_DIR, "SMILExtract")
    hip = os.path.join(ROOT, "opensmile_amd", "smilextract_hip")
    cores = os.cpu_count() or 1
    for name in args.sets.split(","):
        hset, conf_rel, secs, hip_out, ref_opt, ext = SETS[name]
        d = args.dir
        # the files live in memory (/dev/shm): inputs + both outputs must fit with room to spare, or the box dies under the run
        need = args.files * (secs * 32000 + 2 * secs * 100 * 4 * 140) * 1.3
        try:
            avail = {l.split(":")[0]: int(l.split()[1]) * 1024 for l in open("/proc/meminfo")}["MemAvailable"]
            shm = shutil.disk_usage(os.path.dirname(d.rstrip("/")) or "/dev/shm").free
        except Exception:
            avail = shm = 0
        if need > 0.5 * min(avail, shm):
            print(json.dumps({"set": name, "files": args.files, "skipped": f"needs {need / 1e9:.1f} GB of /dev/shm, {min(avail, shm) / 1e9:.1f} GB available"}), flush=True)
            continue
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d + "/in"); os.makedirs(d + "/hip"); os.makedirs(d + "/ref")
        n_samp = int(secs * 16000)
        uniq = [synth.utterance(2 + i, n_samp) for i in range(32)]
        paths = []
        for i in range(args.files):
            p = f"{d}/in/u{i:05d}.wav"
            lldo.write_wav(p, uniq[i % 32])
            paths.append(p)
        lst = d + "/list.txt"
        open(lst, "w").write("\n".join(paths) + "\n")
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "opensmile_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
                   SMILEHIP_TIMING="1")
        cmd_hip = [hip, "--set", hset, "-filelist", lst, "-outdir", d + "/hip"] + hip_out
        subprocess.run(cmd_hip, env=env, capture_output=True, text=True)          # (first run: the files' pages, the code objects)
        runs = []
        for rep in range(args.reps):
            t0 = time.perf_counter()
            r = subprocess.run(cmd_hip, env=env, capture_output=True, text=True)
            runs.append((time.perf_counter() - t0, r))
        t_hip, r = min(runs, key=lambda x: x[0])
        if r.returncode != 0:
            print(json.dumps({"set": name, "error": r.stderr[-400:]}), flush=True)
            continue
        stages = [l for l in r.stderr.splitlines() if "timing:" in l]
        import re
        m = re.search(r"since the first ingest ([0-9.]+) s", stages[-1]) if stages else None
        t_net = float(m.group(1)) if m else None
        # --serve: the same list as the second list of a running server (process, context and plans already there)
        t_serve = None
        try:
            os.makedirs(d + "/hips", exist_ok=True)
            line = f"-filelist {lst} -outdir {d}/hips\n"
            r2 = subprocess.run(cmd_hip[:3] + hip_out + ["--serve"], input=line + line + "quit\n", env=env, capture_output=True, text=True)
            dl = [l.split() for l in r2.stdout.splitlines() if l.startswith("done ")]
            if r2.returncode == 0 and len(dl) == 2 and dl[1][1] == "0":
                t_serve = float(dl[1][2])
        except Exception:
            t_serve = None
        # A/B on the same list: the device stage of round 5 (every chunk's copies and kernels on the null stream, the host waiting for
        # each: SMILEHIP_E2E_SERIAL=1) and the route of round 3 (pageable buffers, serial stages; not at the large file counts)
        t_r5 = None
        for _ in range(2):
            t0 = time.perf_counter()
            subprocess.run(cmd_hip, env=dict(env, SMILEHIP_E2E_SERIAL="1"), capture_output=True, text=True)
            t_r5 = min(t_r5 or 1e9, time.perf_counter() - t0)
        t_old = None
        if args.files <= 10000:
            t0 = time.perf_counter()
            subprocess.run(cmd_hip, env=dict(env, SMILEHIP_NO_PINNED="1"), capture_output=True, text=True)
            t_old = time.perf_counter() - t0
        # the reference-order kernel (SMILEHIP_FORCE_GENERIC=1: the binary's bits) for the byte comparison of the HTK files
        generic_same = generic_cmp = None
        if name == "mfcc12_0_d_a":
            os.makedirs(d + "/hipg", exist_ok=True)
            subprocess.run([hip, "--set", hset, "-filelist", lst, "-outdir", d + "/hipg"] + hip_out, env=dict(env, SMILEHIP_FORCE_GENERIC="1"),
                           capture_output=True, text=True)
        n_ref = args.ref_files or args.files
        conf = os.path.join(lldo.REF_DIR, "config", *conf_rel.split("/"))
        jobs = "\n".join(f"u{i:05d}" for i in range(n_ref))
        cmd = f"xargs -P {cores} -L 1 sh -c '{exe} -C {conf} -I {d}/in/$0.wav {ref_opt} {d}/ref/$0{ext} -l 0 >/dev/null 2>&1'"
        t0 = time.perf_counter()
        subprocess.run(cmd, shell=True, input=jobs.encode(), cwd=d, check=True)
        t_ref = time.perf_counter() - t0
        same = diff = 0
        for i in range(0, n_ref, max(1, n_ref // 64)):
            a, b = f"{d}/hip/u{i:05d}{ext}", f"{d}/ref/u{i:05d}{ext}"
            if os.path.exists(a) and os.path.exists(b):
                if open(a, "rb").read() == open(b, "rb").read():
                    same += 1
                else:
                    diff += 1
        if name == "mfcc12_0_d_a":
            generic_same = generic_cmp = 0
            for i in range(0, n_ref, max(1, n_ref // 64)):
                a, b = f"{d}/hipg/u{i:05d}{ext}", f"{d}/ref/u{i:05d}{ext}"
                if os.path.exists(a) and os.path.exists(b):
                    generic_cmp += 1
                    generic_same += int(open(a, "rb").read() == open(b, "rb").read())
        frames_per_file = {10.0: 998, 3.0: 298}.get(secs)
        print(json.dumps({"set": name, "files": args.files, "seconds_per_file": secs, "smilextract_hip_wall_s": t_hip,
                          "frames_per_s_gross": args.files * frames_per_file / t_hip if frames_per_file else None,
                          "frames_per_s_net_of_startup": args.files * frames_per_file / t_net if (frames_per_file and t_net) else None,
                          "net_s_first_ingest_to_last_sink": t_net, "stages": stages[-1] if stages else None,
                          "serve_second_list_wall_s": t_serve,
                          "frames_per_s_served": args.files * frames_per_file / t_serve if (frames_per_file and t_serve) else None,
                          "round5_serial_device_stage_wall_s": t_r5, "round3_route_wall_s": t_old,
                          "force_generic_outputs_compared": generic_cmp, "force_generic_outputs_byte_identical": generic_same,
                          "smilextract_hip_files_per_s": args.files / t_hip, "smilextract_hip_audio_s_per_s": args.files * secs / t_hip,
                          "reference_files": n_ref, "reference_cores": cores, "reference_wall_s": t_ref,
                          "reference_files_per_s": n_ref / t_ref, "speedup_files_per_s": (args.files / t_hip) / (n_ref / t_ref),
                          "outputs_compared": same + diff, "outputs_byte_identical": same,
                          "note": "smilextract_hip: one process, one GPU, start-up included; reference: one SMILExtract process per file on all host cores"}),
              flush=True)
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
