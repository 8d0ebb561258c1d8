import os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import lldo
import test_gpu_plugin as T
g = np.load("tests/golden/mfcc12_0_d_a_synth.npz")
ref = g["out_u3_16000"]
for comps in ["cVectorPreemphasis", "cWindower", "cFFTmagphase", "cMelspec", "cMfcc", "cTransformFFT"]:
    y, tr = T._run(lldo, g["pcm_u3_16000"], {"SMILEHIP_PLUGIN_COMPONENTS": comps})
    d = np.abs(y - ref)
    print(comps, "bitexact" if np.array_equal(y, ref) else f"max abs {d.max():.3e} n_diff {(y != ref).sum()} of {y.size}", tr.get(comps))
