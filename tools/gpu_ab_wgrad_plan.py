import sys, os
sys.path.insert(0, '/root/repo')
import torch, bench
from tiny_diffusion_amd._lib import lib
for plan in (1, 3, 1, 3):
    lib.tdx_tune_set(b"wgrad_plan", plan)
    rows, flop, ms, nl = bench.conv_roofline(256, reps=10)
    w = [r for r in rows if r["role"] == "wgrad"]
    print(f"wgrad_plan {plan}: leg {flop / ms / 1e9:.1f} TF; wgrad ms " + " ".join(f"{r['ms']:.3f}" for r in w), flush=True)
