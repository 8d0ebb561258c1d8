#!/usr/bin/env python3
"""Generate tests/golden/files/: output FILES of the REAL reference binary (HTK, CSV, ARFF)
for two short synthetic utterances, used by tests/test_host_io.py to pin the writers of
opensmile_amd/host byte for byte and the smilextract_hip front end end to end.

Run from the repo root in a container that has /root/reference:
    python tests/golden/make_golden_files.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import lldo  # noqa: E402
from opensmile_amd import synth  # noqa: E402


def main():
    lldo.build()
    assert lldo.have_ref(), "oracle/_ref/SMILExtract missing (needs /root/reference)"
    exe = os.path.join(lldo.REF_DIR, "SMILExtract")
    out = os.path.join(ROOT, "tests", "golden", "files")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        for u, n in ((2, 8000), (3, 4000)):
            lldo.write_wav(os.path.join(td, f"u{u}.wav"), synth.utterance(u, n))
            shutil.copy(os.path.join(td, f"u{u}.wav"), os.path.join(out, f"u{u}_{n}.wav"))
        # MFCC12_0_D_A: LLD level as HTK + CSV
        conf = os.path.join(lldo.REF_DIR, "config", "mfcc", "MFCC12_0_D_A.conf")
        subprocess.run([exe, "-C", conf, "-I", os.path.join(td, "u2.wav"), "-O", os.path.join(td, "m.htk"),
                        "-csvoutput", os.path.join(td, "m.csv"), "-instname", "utt two", "-l", "0"], check=True, cwd=td)
        shutil.copy(os.path.join(td, "m.htk"), os.path.join(out, "mfcc_u2_8000.htk"))
        shutil.copy(os.path.join(td, "m.csv"), os.path.join(out, "mfcc_u2_8000.csv"))
        # PLP_0_D_A: LLD level as HTK + CSV
        conf = os.path.join(lldo.REF_DIR, "config", "plp", "PLP_0_D_A.conf")
        subprocess.run([exe, "-C", conf, "-I", os.path.join(td, "u2.wav"), "-O", os.path.join(td, "p.htk"),
                        "-csvoutput", os.path.join(td, "p.csv"), "-l", "0"], check=True, cwd=td)
        shutil.copy(os.path.join(td, "p.htk"), os.path.join(out, "plp_u2_8000.htk"))
        shutil.copy(os.path.join(td, "p.csv"), os.path.join(out, "plp_u2_8000.csv"))
        # MFCC12_E_D_A (CSV + HTK through the standard output section) and MFCC12_E_D_A_Z (its own cHtkSink, parmKind 2886)
        conf = os.path.join(lldo.REF_DIR, "config", "mfcc", "MFCC12_E_D_A.conf")
        subprocess.run([exe, "-C", conf, "-I", os.path.join(td, "u2.wav"), "-O", os.path.join(td, "e.htk"),
                        "-csvoutput", os.path.join(td, "e.csv"), "-l", "0"], check=True, cwd=td)
        shutil.copy(os.path.join(td, "e.htk"), os.path.join(out, "mfcc_e_u2_8000.htk"))
        shutil.copy(os.path.join(td, "e.csv"), os.path.join(out, "mfcc_e_u2_8000.csv"))
        conf = os.path.join(lldo.REF_DIR, "config", "mfcc", "MFCC12_E_D_A_Z.conf")
        subprocess.run([exe, "-C", conf, "-I", os.path.join(td, "u2.wav"), "-O", os.path.join(td, "ez.htk"), "-l", "0"],
                       check=True, cwd=td)
        shutil.copy(os.path.join(td, "ez.htk"), os.path.join(out, "mfcc_e_z_u2_8000.htk"))
        # ComParE_2016: the 130-column LLD level as CSV + HTK
        conf = os.path.join(lldo.REF_DIR, "config", "compare16", "ComParE_2016.conf")
        subprocess.run([exe, "-C", conf, "-I", os.path.join(td, "u3.wav"), "-lldcsvoutput", os.path.join(td, "c.csv"),
                        "-lldhtkoutput", os.path.join(td, "c.htk"), "-instname", "u3", "-l", "0"], check=True, cwd=td)
        shutil.copy(os.path.join(td, "c.htk"), os.path.join(out, "compare16_lld_u3.htk"))
        shutil.copy(os.path.join(td, "c.csv"), os.path.join(out, "compare16_lld_u3.csv"))
        # ... and its functionals level (6373 values) as ARFF + HTK
        subprocess.run([exe, "-C", conf, "-I", os.path.join(td, "u3.wav"), "-O", os.path.join(td, "cf.arff"),
                        "-htkoutput", os.path.join(td, "cf.htk"), "-instname", "u3", "-l", "0"], check=True, cwd=td)
        shutil.copy(os.path.join(td, "cf.arff"), os.path.join(out, "compare16_func_u3.arff"))
        shutil.copy(os.path.join(td, "cf.htk"), os.path.join(out, "compare16_func_u3.htk"))
        # IS09_emotion: two files appended into one ARFF / functionals CSV; LLD CSV + HTK and the
        # functionals HTK of the second one (instance names exercise the ARFF escaping)
        conf = os.path.join(lldo.REF_DIR, "config", "is09-13", "IS09_emotion.conf")
        for u, name in ((2, "a.wav"), (3, "b'x")):
            subprocess.run([exe, "-C", conf, "-I", os.path.join(td, f"u{u}.wav"), "-O", os.path.join(td, "f.arff"),
                            "-csvoutput", os.path.join(td, "f.csv"), "-htkoutput", os.path.join(td, f"f{u}.htk"),
                            "-lldcsvoutput", os.path.join(td, f"l{u}.csv"), "-lldhtkoutput", os.path.join(td, f"l{u}.htk"),
                            "-instname", name, "-l", "0"], check=True, cwd=td)
        for f, g in (("f.arff", "is09_func.arff"), ("f.csv", "is09_func.csv"), ("f3.htk", "is09_func_u3.htk"),
                     ("l3.csv", "is09_lld_u3.csv"), ("l3.htk", "is09_lld_u3.htk")):
            shutil.copy(os.path.join(td, f), os.path.join(out, g))
        # eGeMAPSv02: the 25-column LLD level as CSV + HTK, the 88 functionals as ARFF + CSV + HTK
        conf = os.path.join(lldo.REF_DIR, "config", "egemaps", "v02", "eGeMAPSv02.conf")
        subprocess.run([exe, "-C", conf, "-I", os.path.join(td, "u3.wav"), "-lldcsvoutput", os.path.join(td, "g.csv"),
                        "-lldhtkoutput", os.path.join(td, "g.htk"), "-O", os.path.join(td, "gf.arff"), "-csvoutput",
                        os.path.join(td, "gf.csv"), "-htkoutput", os.path.join(td, "gf.htk"), "-instname", "u3", "-l", "0"],
                       check=True, cwd=td)
        for a, b in (("g.csv", "egemaps_lld_u3.csv"), ("g.htk", "egemaps_lld_u3.htk"), ("gf.arff", "egemaps_func_u3.arff"),
                     ("gf.csv", "egemaps_func_u3.csv"), ("gf.htk", "egemaps_func_u3.htk")):
            shutil.copy(os.path.join(td, a), os.path.join(out, b))

    for f in sorted(os.listdir(out)):
        print(f, os.path.getsize(os.path.join(out, f)))


if __name__ == "__main__":
    main()
