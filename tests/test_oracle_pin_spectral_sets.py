"""oracle/lld_oracle_compare.c::lldo_spectral_general -- cSpectral for any number of bands, with maxPos / minPos and every
descriptor optional -- pinned against the real binary on the cSpectral instances of the shipped files the plugin still refuses for
this component (avec2011, emo_large, MediaEval): the instance's own input level in, its own output level out, bit for bit. The
groundwork for the general GPU operator."""
import os
import subprocess

import numpy as np
import pytest

from oracle import lldo

pytestmark = pytest.mark.skipif(not lldo.have_ref(), reason="oracle/_ref not built")

CASES = {
    "avec11-14/avec2011.conf": ("fftmagH25", [(250, 650), (1000, 4000)],
                                dict(flux=1, entropy=1, variance=1, skewness=1, kurtosis=1, sharpness=1, harmonicity=1)),
    "avec11-14/avec2013.conf": ("fftmagH25", [(250, 650), (1000, 4000)],
                                dict(flux=1, entropy=1, variance=1, skewness=1, kurtosis=1, sharpness=1, harmonicity=1, flatness=1)),
    "misc/emo_large.conf": ("fftmag", [(0, 250), (0, 650), (250, 650), (1000, 4000)], dict(flux=1, centroid=1, max_pos=1, min_pos=1)),
    "mediaeval12/MediaEval_Audio_IS12based_subwin2.conf": ("fftmagH25", [(40, 150), (250, 650), (1000, 4000), (5000, 15000)],
                                                           dict(flux=1, centroid=1, entropy=1, variance=1, skewness=1, kurtosis=1, slope=1,
                                                                harmonicity=1, sharpness=1)),
}


@pytest.mark.parametrize("conf", sorted(CASES))
def test_spectral_descriptor_sets_bit_exact(conf, tmp_path):
    from opensmile_amd import synth
    rd, bands, flags = CASES[conf]
    td = str(tmp_path)
    for u, n in ((9, 32000), (3, 12000)):
        wav = os.path.join(td, "in.wav")
        lldo.write_wav(wav, synth.utterance(u, n), 16000)
        c = os.path.join(td, "t.conf")
        open(c, "w").write("\\{%s}\n[componentInstances:cComponentManager]\ninstance[tap_in].type=cHtkSink\ninstance[tap_out].type=cHtkSink\n"
                           "[tap_in:cHtkSink]\nreader.dmLevel=%s\nfilename=%s/tap_in.htk\n[tap_out:cHtkSink]\nreader.dmLevel=spectral\n"
                           "filename=%s/tap_out.htk\n" % (os.path.join(lldo.REF_DIR, "config", conf), rd, td, td))
        subprocess.run([os.path.join(lldo.REF_DIR, "SMILExtract"), "-C", c, "-I", wav, "-O", os.path.join(td, "o.bin"), "-l", "0"], cwd=td,
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        mag = lldo.read_htk(os.path.join(td, "tap_in.htk"))[0]
        ref = lldo.read_htk(os.path.join(td, "tap_out.htk"))[0]
        got = lldo.spectral_general_rows(mag, (mag.shape[1] - 1) * 2 / 16000.0, bands, **flags)
        assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (conf, u, n)


# Round 6: the options of the linear-spectrum branch no shipped file uses (specDiff, specPosDiff, fluxCentroid, fluxAtFluxCentroid,
# standardDeviation, slopes[]): a second cSpectral instance behind avec2011's magnitude level, its own output level tapped.
EXTRA = {
    "all": ([(250, 650), (1000, 4000)], [(0, 500), (500, 1500), (1500, 8000)],
            dict(spec_diff=1, spec_pos_diff=1, flux=1, flux_centroid=1, flux_at_flux_centroid=1, centroid=1, standard_deviation=1, variance=1,
                 skewness=1)),
    "noflux": ([(0, 250)], [(300, 3400)], dict(spec_diff=1, flux_centroid=1, standard_deviation=1)),
    "posdiff_only": ([], [], dict(spec_pos_diff=1, flux_at_flux_centroid=1, kurtosis=1, entropy=1)),
}
CONF_NAMES = {"spec_diff": "specDiff", "spec_pos_diff": "specPosDiff", "flux": "flux", "flux_centroid": "fluxCentroid",
              "flux_at_flux_centroid": "fluxAtFluxCentroid", "centroid": "centroid", "standard_deviation": "standardDeviation",
              "variance": "variance", "skewness": "skewness", "kurtosis": "kurtosis", "entropy": "entropy", "max_pos": "maxPos",
              "min_pos": "minPos", "slope": "slope", "sharpness": "sharpness", "harmonicity": "harmonicity", "flatness": "flatness"}


def spectral_section(name, reader, writer, bands, slopes, flags, rolloff=(0.25, 0.5, 0.75, 0.9)):
    """a cSpectral section with exactly these options on (the component's defaults switch flux, centroid, maxPos, minPos on)"""
    lines = [f"[{name}:cSpectral]", f"reader.dmLevel={reader}", f"writer.dmLevel={writer}", "copyInputName=1", "processArrayFields=1",
             "squareInput=1"]
    lines += [f"bands[{i}]={a}-{b}" for i, (a, b) in enumerate(bands)]
    lines += [f"slopes[{i}]={a}-{b}" for i, (a, b) in enumerate(slopes)]
    lines += [f"rollOff[{i}]={r}" for i, r in enumerate(rolloff)]
    for key, conf in CONF_NAMES.items():
        lines.append(f"{conf}={int(bool(flags.get(key, 0)))}")
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("case", sorted(EXTRA))
def test_spectral_round6_options_bit_exact(case, tmp_path):
    from opensmile_amd import synth
    bands, slopes, flags = EXTRA[case]
    td = str(tmp_path)
    for u, n in ((9, 32000), (3, 12000)):
        wav = os.path.join(td, "in.wav")
        lldo.write_wav(wav, synth.utterance(u, n), 16000)
        c = os.path.join(td, "t.conf")
        open(c, "w").write("\\{%s}\n[componentInstances:cComponentManager]\ninstance[spec2].type=cSpectral\ninstance[tap_in].type=cHtkSink\n"
                           "instance[tap_out].type=cHtkSink\n%s[tap_in:cHtkSink]\nreader.dmLevel=fftmagH25\nfilename=%s/tap_in.htk\n"
                           "[tap_out:cHtkSink]\nreader.dmLevel=spectral2\nfilename=%s/tap_out.htk\n"
                           % (os.path.join(lldo.REF_DIR, "config", "avec11-14/avec2011.conf"),
                              spectral_section("spec2", "fftmagH25", "spectral2", bands, slopes, flags), td, td))
        subprocess.run([os.path.join(lldo.REF_DIR, "SMILExtract"), "-C", c, "-I", wav, "-O", os.path.join(td, "o.bin"), "-l", "0"], cwd=td,
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        mag = lldo.read_htk(os.path.join(td, "tap_in.htk"))[0]
        ref = lldo.read_htk(os.path.join(td, "tap_out.htk"))[0]
        got = lldo.spectral_general_rows(mag, (mag.shape[1] - 1) * 2 / 16000.0, bands, slopes=slopes, **flags)
        assert got.shape == ref.shape, (got.shape, ref.shape)
        bad = np.argwhere(got.view(np.uint32) != ref.view(np.uint32))
        assert bad.size == 0, (case, u, n, bad[:8].tolist(), got[bad[0][0]].tolist(), ref[bad[0][0]].tolist())
