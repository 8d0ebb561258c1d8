"""The padded orders of the fused half-length transform (opensmile_amd/csrc/lld_fft.hpp), restated on the host: that the passes
with these index formulas compute the DFT, and that every b64 LDS instruction of every pass is bank-conflict-free (the 32 lanes
of a half-wave touch 32 different (re, im) slots modulo the 32 bank pairs) -- the claim DESIGN.md makes. The device code itself
is held against the in-place radix-2 transform bit for bit by tests/test_gpu_fft.py."""
import numpy as np


def brev(x, bits):
    return int(format(x, f"0{bits}b")[::-1], 2)


def passes(logm):
    """[(read index fn(lane, m) or None for the loader, write index fn(lane, m), points per lane, h0)]"""
    if logm == 9:
        return [(None, lambda l, k: 9 * l + k, 8, 1),
                (lambda l, m: (l & 7) + 72 * (l >> 3) + 9 * m, lambda l, m: (l & 7) + 72 * (l >> 3) + 8 * m, 8, 8),
                (lambda l, m: l + 72 * m, lambda l, m: l + 72 * m, 8, 64)]
    return [(None, lambda l, k: 5 * l + k, 4, 1),
            (lambda l, m: (l & 3) + 20 * (l >> 2) + 5 * m, lambda l, m: (l & 3) + 20 * (l >> 2) + 4 * m, 4, 4),
            (lambda l, m: (l & 15) + 80 * (l >> 4) + 20 * m, lambda l, m: (l & 15) + 80 * (l >> 4) + 16 * m, 4, 16),
            (lambda l, m: l + 80 * m, lambda l, m: l + 80 * m, 4, 64)]


def element_of(logm, h0, pts, lane, m):
    """which element (index in the bit-reversed order) local point m of `lane` is in the pass with first half h0"""
    if h0 == 1:
        return pts * lane + m
    jb, hi = lane & (h0 - 1), lane // h0
    return jb + h0 * m + hi * pts * h0


def run(logm, x):
    M = 1 << logm
    kz = 576 if logm == 9 else 320
    z = np.zeros(kz, np.complex128)
    tw = np.exp(-2j * np.pi * np.arange(M // 2) / M)
    for rd, wr, pts, h0 in passes(logm):
        vals = {}
        for lane in range(64):
            if rd is None:
                rg = brev(lane, 6)
                v = [None] * pts
                for k in range(pts):
                    v[brev(k, 3 if pts == 8 else 2)] = x[rg + 64 * k]
            else:
                v = [z[rd(lane, m)] for m in range(pts)]
            # log2(pts) radix-2 stages on local points: halves h0, 2 h0, ...
            base = element_of(logm, h0, pts, lane, 0)
            half = 1
            while half < pts:
                for m in range(pts):
                    if m & half:
                        continue
                    e0 = base + h0 * m
                    j = e0 & (h0 * half - 1)
                    w = tw[j * (M // (2 * h0 * half))]
                    a, b = v[m], v[m + half] * w
                    v[m], v[m + half] = a + b, a - b
                half *= 2
            vals[lane] = v
        for lane in range(64):                         # all reads of a pass precede its writes
            for m in range(pts):
                z[wr(lane, m)] = vals[lane][m]
    pad = 8 if logm == 9 else 16
    return np.array([z[e + (e >> 6) * pad] for e in range(M)])


def test_the_documented_orders_compute_the_dft():
    rng = np.random.default_rng(0)
    for logm in (8, 9):
        M = 1 << logm
        x = rng.standard_normal(M) + 1j * rng.standard_normal(M)
        assert np.abs(run(logm, x) - np.fft.fft(x)).max() < 1e-10 * M


def test_every_lds_access_of_every_pass_is_conflict_free():
    for logm in (8, 9):
        for rd, wr, pts, h0 in passes(logm):
            for fn in (rd, wr):
                if fn is None:
                    continue
                for m in range(pts):                   # one b64 instruction: fixed m, 64 lanes, two half-waves
                    for half in (range(0, 32), range(32, 64)):
                        slots = {fn(lane, m) % 32 for lane in half}
                        assert len(slots) == 32, (logm, h0, m)
                # and nothing leaves the buffer
                assert max(fn(lane, m) for lane in range(64) for m in range(pts)) < (576 if logm == 9 else 320)
