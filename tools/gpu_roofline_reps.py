#!/usr/bin/env python3
"""The roofline leg of bench.py with a different number of back-to-back repetitions per launch (default 5):
does a layer's time depend on how long the same kernel has been running?  python tools/gpu_roofline_reps.py 5 40"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
torch.zeros(1, device="cuda")
for reps in [int(a) for a in sys.argv[1:]] or [5, 40]:
    rows, flop, ms, n = bench.conv_roofline(256, reps=reps)
    print(f"reps {reps}: {flop / ms / 1e9:.1f} TFLOP/s")
    for i in range(0, len(rows), 3):
        r = rows[i]
        print(f"   {r['cin']:5d}->{r['cout']:4d}@{r['hw']:2d}", " ".join(f"{x['role']} {x['ms'] * 1e3:7.1f}" for x in rows[i:i + 3]), flush=True)
