"""log_d (opensmile_amd/csrc/lld_device.hpp), the table + polynomial logarithm of the frame kernels, measured on the host: the
same table and the same operations in C (tests/helpers/log_d_host.c, fma() where the kernel calls fma) against numpy's
long-double log. The GPU test (tests/test_gpu_fft.py) measures the device build itself; this one pins the ALGORITHM's accuracy
where no GPU is available, on many more arguments."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_log_d_algorithm_accuracy_on_the_host(tmp_path):
    so = str(tmp_path / "liblogd.so")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "opensmile_amd", "csrc"),
                    os.path.join(ROOT, "tests", "helpers", "log_d_host.c"), "-o", so, "-lm"], check=True)
    lib = C.CDLL(so)
    lib.log_d_host_array.restype = None
    lib.log_d_host_array.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
    rng = np.random.default_rng(3)
    parts = [np.exp(rng.uniform(-700, 700, 2000000)), 1.0 + rng.uniform(-0.3125, 0.375, 2000000), 1.0 + rng.uniform(-1e-3, 1e-3, 500000),
             1.0 + np.exp(rng.uniform(-40, 0, 500000)), rng.uniform(0.0, 2.0, 1000000).astype(np.float32).astype(np.float64) + 1.0,
             # every sub-interval boundary of the table and its neighbours
             np.concatenate([[0.6875 + i * 2.0 ** -8 for i in range(81)], [1.0 + i * 2.0 ** -7 for i in range(49)]]),
             np.array([1.0, 2.0, 0.5, np.nextafter(1.0, 0), np.nextafter(1.0, 2), 2.2250738585072014e-308, 1.7976931348623157e308])]
    x = np.concatenate(parts)
    x = np.concatenate([x, np.nextafter(parts[5], 0), np.nextafter(parts[5], 4)])
    y = np.empty_like(x)
    lib.log_d_host_array(x.ctypes.data, y.ctypes.data, len(x))
    ref = np.log(x.astype(np.longdouble))
    ulp = np.spacing(np.abs(ref.astype(np.float64))).astype(np.longdouble)
    err = np.abs(y.astype(np.longdouble) - ref) / ulp
    err[ref == 0] = np.abs(y[ref == 0])
    # <= 1 ulp except where k ln2 + log c lands just above a binade boundary and the result just below it
    assert err.max() <= 1.5, (float(err.max()), x[np.argmax(err)])
    assert (err > 1.0).mean() <= 1e-4 and err.mean() <= 0.3
    # rounded to float -- how the kernels use it -- it is the correctly rounded float logarithm but for near-ties
    assert (y.astype(np.float32) != ref.astype(np.float32)).mean() <= 1e-6
    # and it agrees with the platform's libm (what the reference calls) to the same degree
    assert (y.astype(np.float32) != np.log(x).astype(np.float32)).mean() <= 1e-6
