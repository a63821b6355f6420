#!/usr/bin/env python3
"""Timing experiments on the conv main loop (tdx_tune_set "conv_dbg")."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from tiny_diffusion_amd._lib import lib
cases = [(int(x), f"dbg={x}") for x in sys.argv[1:]] or [(0, "full"), (8, "stagger 2048 cyc"), (16, "stagger 4096 cyc"), (24, "stagger 6144 cyc")]
for rep in range(2):
    for dbg, name in cases:
        lib.tdx_tune_set(b"conv_dbg", dbg)
        rows, flop, ms, nl = bench.conv_roofline(256, reps=8)
        fd = [r for r in rows if r["role"] != "wgrad"]
        print(f"{name:22s}: fwd+dgrad {sum(r['ms'] for r in fd):7.3f} ms | " + " ".join(f"{r['tflops']:.0f}" for r in fd))
lib.tdx_tune_set(b"conv_dbg", 0)
