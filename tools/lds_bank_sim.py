#!/usr/bin/env python3
"""LDS bank-conflict model of the fast MFCC kernel's wave-level access patterns
(rules from /opt/skills/guides/MI355X_MICROARCH.md, LDS section). Prints the extra
(conflict) cycles per DS instruction class for one pass of one wave."""
import numpy as np

H, MP, U = 160, 13, 6
shared = 512 + MP * 32 + 512 + U * 144 + 448 + 64
stage_alloc = 896
wave_floats = stage_alloc + 4 + 4 * 272


def groups(kind):
    if kind in ("r32", "w32", "r64"):
        return [list(range(0, 32)), list(range(32, 64))]
    if kind in ("w64", "r2_64"):
        return [list(range(16 * i, 16 * i + 16)) for i in range(4)]
    if kind == "r128":
        return [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
                [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
                [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]
    raise ValueError(kind)


def cost(kind, addr_bytes, active=None):
    """extra cycles beyond the conflict-free cost for one wave instruction"""
    nb = {"r32": 32, "w32": 32, "w64": 32, "r2_64": 32, "r64": 64, "r128": 64}[kind]
    width = {"r32": 1, "w32": 1, "w64": 2, "r2_64": 2, "r64": 2, "r128": 4}[kind]
    extra = 0
    for g in groups(kind):
        per_bank = {}
        for l in g:
            if active is not None and not active[l]:
                continue
            a = int(addr_bytes[l])
            for k in range(width):
                bank = ((a // 4) + k) % nb
                per_bank.setdefault(bank, set()).add((a // 4 + k))
        worst = max((len(v) for v in per_bank.values()), default=1)
        extra += worst - 1
    return extra


lane = np.arange(64)
g, j = lane >> 4, lane & 15
wbase = shared  # wave 0
stage = wbase
gb = wbase + stage_alloc + 4 + g * 272
pb = gb + ((g & 1) << 2)
ps = stage + g * 128
tot = {}


def add(name, c):
    tot[name] = tot.get(name, 0) + c


for r in range(7):
    add("stage write b64", cost("w64", 4 * (stage + 2 * (lane + 64 * r))))
for m in range(MP):
    add("frame read b64", cost("r64", 4 * (stage + g * H + 2 * j + 32 * m)))
    add("window read b64", cost("r64", 4 * (512 + 2 * (m * 16 + j))))
for k1 in range(1, 16):
    add("tw256 read b64", cost("r64", 4 * (512 + MP * 32 + 2 * (k1 * 16 + j))))
for k1 in range(16):
    add("TB write b32", 2 * cost("w32", 4 * (gb + k1 * 17 + j)))   # re and im
for jj in range(16):
    add("TB read b32", 2 * cost("r32", 4 * (gb + j * 17 + jj)))
for k2 in range(8):
    add("ZX write b64", cost("w64", 4 * (gb + 2 * (k2 * 16 + j))))
pj, zrow = (16 - j) & 15, np.where(j == 0, 16, 0)
for q in range(8):
    add("ZX read b64", cost("r64", 4 * (gb + 2 * ((7 - q) * 16 + pj + zrow))))
    add("tw512 read b64", cost("r64", 4 * (2 * (j + 16 * q))))
    add("PB write b32", cost("w32", 4 * (pb + j + 16 * q)) + cost("w32", 4 * (pb + 256 - j - 16 * q)))
# mel units: octets per unit from the real bank would need the table; model consecutive octets
rng = np.random.default_rng(0)
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
try:
    from opensmile_amd import capi
    p = capi.Plan(None)
    chan = p.mel_chanmap()
    # rebuild units as fast512_build_host does
    nb = 26
    units = []
    for b in range(nb):
        rise = np.where(chan == b - 1)[0]
        fall = np.where(chan == b)[0]
        bins = np.concatenate([rise, fall]) if b > 0 else fall
        lo, hi = bins.min(), bins.max() + 1
        for o in range(lo // 8, (hi - 1) // 8 + 1):
            units.append(o)
    print("mel units:", len(units))
    for i in range((len(units) + 15) // 16):
        octs = np.array([units[i * 16 + jj] if i * 16 + jj < len(units) else 0 for jj in range(16)])
        a = 4 * (pb) + 32 * octs[j]
        add("mel p read b128", cost("r128", a) + cost("r128", a + 16))
        add("mel w read b128", 2 * cost("r128", 4 * (512 + MP * 32 + 512 + 4 * (i * 16 + j))))
        add("PS write b32", cost("w32", 4 * (ps + (i * 16 + j))))
except Exception as e:  # pragma: no cover
    print("mel model skipped:", e)
for q in range(7):
    act = j < 13
    add("lmel read b128", cost("r128", 4 * (ps + 96 + 4 * q + 0 * j), act))
    add("dct read b128", cost("r128", 4 * (shared - 64 - 448 + j * 28 + 4 * q), act))
for name, c in tot.items():
    print(f"{name:20s} extra cycles {c}")
print("total extra", sum(tot.values()))
