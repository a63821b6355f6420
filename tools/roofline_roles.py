#!/usr/bin/env python3
"""Sum of the roofline leg's per-launch times by role, for A/B of plans:  python bench.py --roofline-only | python tools/roofline_roles.py"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])["roofline"]
tot = {}
for l in d["per_launch"]:
    tot[l["role"]] = tot.get(l["role"], 0) + l["ms"]
print({k: round(v, 3) for k, v in tot.items()}, "achieved", d["achieved"])
if len(sys.argv) > 1:
    for l in d["per_launch"]:
        if l["role"] == sys.argv[1]:
            print(f"   {l['cin']:5d}->{l['cout']:4d}@{l['hw']:2d} {l['ms'] * 1e3:7.1f} us {l['tflops']:6.1f} TF")
