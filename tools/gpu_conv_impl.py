#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from tiny_diffusion_amd._lib import lib
for rep in range(2):
    for impl in (1, 2):
        lib.tdx_tune_set(b"conv_impl", impl)
        rows, flop, ms, nl = bench.conv_roofline(256, reps=8)
        fd = [r for r in rows if r["role"] != "wgrad"]
        print(f"impl {impl}: fwd+dgrad {sum(r['ms'] for r in fd):7.3f} ms | " + " ".join(f"{r['tflops']:.0f}" for r in fd))
