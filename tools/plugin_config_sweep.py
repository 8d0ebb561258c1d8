"""Every configuration file the reference ships (config/**/*.conf), unmodified, inside the unmodified binary: once plain and once with
the plugin (every override active, CPU fall-through NOT allowed). For each file that the plain binary can run on a wave file with one
of the standard output options: does the plugin run refuse it (and why), or complete -- and are the output files byte-identical?
    python tools/plugin_config_sweep.py --probe   # CPU only: find, per file, an output option the plain binary produces a file with
    python tools/plugin_config_sweep.py           # GPU: plain vs plugin, one JSON line per file (stdout)"""
import argparse
import glob
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import lldo  # noqa: E402  (test infrastructure: wave writer and the reference binary's location)
from opensmile_amd import synth  # noqa: E402

PLUGDIR = os.path.join(ROOT, "opensmile_amd", "plugin")
PROBE = os.path.join(ROOT, "tools", "plugin_config_sweep_probe.json")
OPTS = [["-O"], ["-csvoutput"], ["-lldcsvoutput"], ["-output"], ["-arffout"], ["-csv"], ["-htkoutput"]]


def run(conf, wav, opt, out, env_extra, cwd):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "opensmile_amd"), lldo.REF_DIR, env.get("LD_LIBRARY_PATH", "")])
    env.setdefault("SMILEHIP_PLUGIN_FUSE", "0")          # (this tool checks the per-component operators; fused mode is the default since round 5)
    env.update(env_extra)
    try:
        r = subprocess.run([os.path.join(lldo.REF_DIR, "SMILExtract"), "-C", conf, "-I", wav] + opt + [out, "-l", "1", "-nologfile"],
                           cwd=cwd, env=env, capture_output=True, text=True, errors="replace", timeout=120)
        return r.returncode, r.stderr
    except subprocess.TimeoutExpired:
        return -999, "timeout"


def digest(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()[:16] if os.path.exists(p) and os.path.getsize(p) > 0 else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--probe", action="store_true")
    ap.add_argument("--only", default="", help="substring of the files to run")
    a = ap.parse_args()
    confs = sorted(glob.glob(os.path.join(lldo.REF_DIR, "config", "**", "*.conf"), recursive=True))
    pcm = synth.utterance(71, 24000)
    with tempfile.TemporaryDirectory() as td:
        wav = os.path.join(td, "in.wav")
        lldo.write_wav(wav, pcm, 16000)
        if a.probe:
            found = {}
            for c in confs:
                rel = os.path.relpath(c, os.path.join(lldo.REF_DIR, "config"))
                for opt in OPTS:
                    out = os.path.join(td, "o.bin")
                    if os.path.exists(out):
                        os.remove(out)
                    rc, err = run(c, wav, opt, out, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, td)
                    if rc == 0 and digest(out):
                        found[rel] = opt
                        break
                print(rel, found.get(rel), file=sys.stderr)
            json.dump(found, open(PROBE, "w"), indent=1, sort_keys=True)
            return
        found = json.load(open(PROBE))
        for rel, opt in sorted(found.items()):
            if a.only and a.only not in rel:
                continue
            c = os.path.join(lldo.REF_DIR, "config", rel)
            o1, o2 = os.path.join(td, "plain.bin"), os.path.join(td, "plugin.bin")
            for o in (o1, o2):
                if os.path.exists(o):
                    os.remove(o)
            rc1, _ = run(c, wav, opt, o1, {"SMILEHIP_PLUGIN_COMPONENTS": "none"}, PLUGDIR)
            trace = os.path.join(td, "trace.txt")
            if os.path.exists(trace):
                os.remove(trace)
            rc2, err2 = run(c, wav, opt, o2, {"SMILEHIP_PLUGIN_TRACE": trace}, PLUGDIR)
            tr = dict(l.split() for l in open(trace).read().split("\n") if l.strip()) if os.path.exists(trace) else {}
            gpu = {k: int(v) for k, v in tr.items() if int(v) and not k.endswith(".cpu") and not k.startswith("fused.")}
            refused = [l.split("libsmilehip plugin: ")[-1][:160] for l in err2.split("\n") if "libsmilehip plugin:" in l and "(ERR)" in l]
            d1, d2 = digest(o1), digest(o2)
            print(json.dumps({"conf": rel, "option": opt[0], "plain_rc": rc1, "plugin_rc": rc2, "identical": bool(d1 and d1 == d2),
                              "frames_on_gpu": sum(gpu.values()), "components_on_gpu": sorted(gpu), "refused": refused[:1]}), flush=True)


if __name__ == "__main__":
    main()
