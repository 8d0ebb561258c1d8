"""GeMAPSv01b.conf and eGeMAPSv01b.conf as column subsets of eGeMAPSv02.conf, held against the REAL binary on the host: for
every test input, what the binary writes for the smaller sets is -- name for name, row for row, bit for bit -- the selected
columns of what it writes for eGeMAPSv02 (the sub-graphs are included by the v02 file: same instances, same levels), with the
selection opensmile_amd/host's egemaps_subset_columns() encodes. The fused path computes the v02 levels
(tests/test_gpu_egemaps.py holds it against the same binary); `smilextract_hip --set gemapsv01b | egemapsv01b` writes these
columns of them."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
EXE = os.path.join(REF, "SMILExtract")
G = os.path.join(ROOT, "tests", "golden", "files")

needs_ref = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/SMILExtract not built")


def run_ref(conf, wav, td, tag):
    from test_host_io import read_htk
    lld, func, csv, fcsv = (os.path.join(td, tag + e) for e in (".lld.htk", ".func.htk", ".lld.csv", ".func.csv"))
    env = dict(os.environ, SMILEHIP_PLUGIN_COMPONENTS="none", LD_LIBRARY_PATH=REF)
    subprocess.run([EXE, "-C", os.path.join(REF, "config", conf), "-I", wav, "-lldhtkoutput", lld, "-htkoutput", func, "-lldcsvoutput", csv,
                    "-csvoutput", fcsv, "-instname", "x", "-l", "0"], check=True, env=env, cwd=td)
    names = open(csv).readline().strip().split(";")[2:]
    fnames = open(fcsv).readline().strip().split(";")[2:]
    return read_htk(lld)[1], read_htk(func)[1], names, fnames


@needs_ref
def test_smaller_gemaps_sets_are_column_subsets_of_v02(tmp_path):
    from test_host_io import build_hostlib
    L = build_hostlib()
    L.shim_egemaps_subset.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.shim_egemaps_names.argtypes = [C.c_int, C.c_char_p, C.c_int]
    buf = C.create_string_buffer(1 << 16)
    own = {}
    for func in (0, 1):
        n = L.shim_egemaps_names(func, buf, len(buf))
        own[func] = buf.value.decode().split(";")
        assert n == len(own[func]) == (88 if func else 25)

    def cols(setname, func):
        out = (C.c_int * 128)()
        n = L.shim_egemaps_subset(setname.encode(), func, out, 128)
        return list(out[:n])

    assert cols("egemapsv02", 0) == [] and len(cols("gemapsv01b", 0)) == 18 and len(cols("gemapsv01b", 1)) == 62
    assert len(cols("egemapsv01b", 0)) == 23 and cols("egemapsv01b", 1) == list(range(88))
    from opensmile_amd import synth
    from oracle import lldo
    wavs = [os.path.join(G, "u3_4000.wav")]
    for i, n in enumerate((48000, 9000, 1200)):         # 3 s, short, shorter than a 60 ms frame + one 20 ms frame only
        w = str(tmp_path / f"s{i}.wav")
        lldo.write_wav(w, synth.utterance(30 + i, n))
        wavs.append(w)
    for w in wavs:
        x2, f2, n2, fn2 = run_ref("egemaps/v02/eGeMAPSv02.conf", w, str(tmp_path), "v02")
        assert n2 == own[0] and (fn2 == own[1] or f2.shape[0] == 0)
        for setname, conf in (("gemapsv01b", "gemaps/v01b/GeMAPSv01b.conf"), ("egemapsv01b", "egemaps/v01b/eGeMAPSv01b.conf")):
            x, f, n, fn = run_ref(conf, w, str(tmp_path), setname)
            cl, cf = cols(setname, 0), cols(setname, 1)
            assert n == [n2[c] for c in cl], setname
            assert x.shape == (x2.shape[0], len(cl)) and np.array_equal(x.view(np.uint32), x2[:, cl].view(np.uint32)), (setname, w)
            assert f.shape[0] == f2.shape[0]
            if f.shape[0]:
                assert fn == [fn2[c] for c in cf]
                assert np.array_equal(f.view(np.uint32), f2[:, cf].view(np.uint32)), (setname, w)


@needs_ref
def test_v01a_sets_equal_the_oracle_in_v01a_mode(tmp_path):
    """GeMAPSv01a.conf / eGeMAPSv01a.conf: the v01b sub-graphs with three option values of openSMILE 2.2 (zeroPadSymmetric = 0 in both
    cTransformFFT instances, useBrokenJitterThresh = 1, cFormantLpc maxF = 5500). The oracle's eGeMAPS chain with those values
    (lldo_gemaps_set_v01a), cut down to the sets' columns, is the binary's output bit for bit -- LLD level and functionals."""
    from test_host_io import build_hostlib
    from opensmile_amd import synth
    from oracle import lldo
    L = build_hostlib()
    L.shim_egemaps_subset.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_int), C.c_int]

    def cols(setname, func):
        out = (C.c_int * 128)()
        n = L.shim_egemaps_subset(setname.encode(), func, out, 128)
        return list(out[:n])

    assert cols("gemapsv01a", 0) == cols("gemapsv01b", 0) and cols("egemapsv01a", 1) == list(range(88))
    lib = lldo.lib()
    differs = 0
    for i, n in enumerate((48000, 30000, 9000)):
        pcm = synth.utterance(60 + i, n)
        w = str(tmp_path / f"a{i}.wav")
        lldo.write_wav(w, pcm)
        lib.lldo_gemaps_set_v01a(1)
        try:
            lld, fn = lldo.egemaps_lld_chain(pcm), lldo.egemaps_func(pcm)
        finally:
            lib.lldo_gemaps_set_v01a(0)
        differs += int(not np.array_equal(lld, lldo.egemaps_lld_chain(pcm)))
        for setname, conf in (("gemapsv01a", "gemaps/v01a/GeMAPSv01a.conf"), ("egemapsv01a", "egemaps/v01a/eGeMAPSv01a.conf")):
            x, f, names, fnames = run_ref(conf, w, str(tmp_path), setname)
            cl, cf = cols(setname, 0), cols(setname, 1)
            assert x.shape == (lld.shape[0], len(cl)), (setname, x.shape, lld.shape)
            assert np.array_equal(x.view(np.uint32), np.ascontiguousarray(lld[:, cl]).view(np.uint32)), (setname, i)
            assert f.shape[0] == fn.shape[0]
            if f.shape[0]:
                assert np.array_equal(f.view(np.uint32), np.ascontiguousarray(fn[:, cf]).view(np.uint32)), (setname, i)
    assert differs >= 2            # the option values matter: the v02 chain gives other numbers
