"""opensmile_amd/csrc/glibc_float.hpp (logf / expf / log10f in glibc 2.35's operation order, the functions the device calls
where the reference calls the C library's float functions) compiled for the host and swept against the REAL libm.
The default run takes every 61st float plus the complete binades around 1 (seconds); GLIBC_FLOAT_FULL=1 sweeps all 2^32
arguments of each function (~100 s; result of the round-3 run: profiles/r03_glibc_float_sweep.json, 0 mismatches)."""
import ctypes as C
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    src = os.path.join(ROOT, "tests", "helpers", "glibc_float_check.cpp")
    so = os.path.join(ROOT, "tests", "helpers", "_glibc_float_check.so")
    hdr = os.path.join(ROOT, "opensmile_amd", "csrc", "glibc_float.hpp")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fno-builtin-logf",
                        "-fno-builtin-expf", "-fno-builtin-log10f", "-fno-builtin-atanf", "-fno-builtin-atan2f", "-fno-builtin-acosf", "-o", so, src, "-lm"], check=True)
    L = C.CDLL(so)
    L.glibc_float_sweep.restype = C.c_longlong
    L.glibc_float_sweep.argtypes = [C.c_int, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_uint)]
    return L


@pytest.mark.parametrize("which,name", [(0, "logf"), (1, "expf"), (2, "log10f"), (3, "atanf"), (4, "acosf")])
def test_bits_equal_libm(lib, which, name):
    if "fma" not in open("/proc/cpuinfo").read():
        pytest.skip("CPU without FMA: the dynamic linker selects glibc's non-FMA build of logf / expf")
    fb = C.c_uint(0)
    if os.environ.get("GLIBC_FLOAT_FULL") == "1":
        ranges = [(0, 1 << 32, 1)]
    else:
        ranges = [(0, 1 << 32, 61), (0x3e800000, 0x40800000, 1), (0xbe800000, 0xc0800000, 1), (0, 0x01000000, 7)]
    for lo, hi, step in ranges:
        bad = lib.glibc_float_sweep(which, lo, hi, step, C.byref(fb))
        assert bad == 0, f"{name}: {bad} arguments differ from libm in [{lo:#x}, {hi:#x}) step {step}, first {fb.value:#010x}"


def test_atan2f_pairs_equal_libm(lib):
    """glibc_atan2f (fdlibm's float atan2 over glibc_atanf) on pseudo-random pairs, a third with close exponents, special values
    mixed in (GLIBC_FLOAT_FULL=1: 3e8 pairs)."""
    lib.glibc_atan2f_pairs.restype = C.c_longlong
    lib.glibc_atan2f_pairs.argtypes = [C.c_ulonglong, C.c_ulonglong, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    by, bx = C.c_uint(0), C.c_uint(0)
    n = 300_000_000 if os.environ.get("GLIBC_FLOAT_FULL") == "1" else 20_000_000
    bad = lib.glibc_atan2f_pairs(n, 12345, C.byref(by), C.byref(bx))
    assert bad == 0, f"atan2f: {bad} of {n} pairs differ from libm, first y = {by.value:#010x}, x = {bx.value:#010x}"
