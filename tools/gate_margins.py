#!/usr/bin/env python3
"""usage: gate_margins.py <gate log (SMILEHIP_GATE_LOG of a `pytest -m gpu` run)> <out.json>: per gate and measured quantity the
number of records and the extremes -- what the parity gates of tests/ are held against (they sit at about twice `max`)."""
import json
import sys

agg = {}
for line in open(sys.argv[1]):
    d = json.loads(line)
    g = agg.setdefault(d.pop("gate"), {})
    for k, v in d.items():
        if isinstance(v, str):
            continue
        e = g.setdefault(k, {"n": 0, "max": v, "min": v})
        e["n"] += 1
        e["max"] = max(e["max"], v)
        e["min"] = min(e["min"], v)
json.dump(agg, open(sys.argv[2], "w"), indent=1)
print(len(agg), "gates")
