#!/usr/bin/env python3
"""usage: tools/dev/serial_loads.py <kernel.s> ...  -> per kernel: vector-memory loads (global / flat / buffer) and how many of them are
waited for alone -- an `s_waitcnt vmcnt(0)` within a few instructions behind the load, with no other load in between: the signature of a
load inside a lane-conditional branch (the compiler waits before the branch ends). CPU only (assembly from `hipcc -S`)."""
import re
import sys

for path in sys.argv[1:]:
    L = open(path).read().split("\n")
    i = 0
    while i < len(L):
        m = re.match(r"^(_ZN8smilehip[A-Za-z0-9_]+):", L[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        j = i + 1
        loads = alone = 0
        while j < len(L) and "s_endpgm" not in L[j]:
            t = L[j].strip()
            if re.match(r"(global|flat|buffer)_load", t) and "lds" not in t.split()[0]:
                loads += 1
                for k in range(j + 1, min(j + 6, len(L))):
                    u = L[k].strip()
                    if re.match(r"(global|flat|buffer)_load", u):
                        break
                    if re.search(r"s_waitcnt vmcnt\(0\)", u):
                        alone += 1
                        break
            j += 1
        if alone >= 4:
            print("%-70s loads %4d  waited for alone %4d" % (name[12:82], loads, alone))
        i = j
