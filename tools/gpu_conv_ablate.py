#!/usr/bin/env python3
"""Timing-only ablations of the conv main loop (results are wrong by construction)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from tiny_diffusion_amd._lib import lib
for dbg, name in ((0, "full"), (1, "no barrier"), (2, "no LDS stores"), (4, "no global loads"), (6, "no loads, no stores"), (7, "MFMA+ds_read only")):
    lib.tdx_tune_set(b"conv_dbg", dbg)
    rows, flop, ms, nl = bench.conv_roofline(256, reps=8)
    fd = [r for r in rows if r["role"] != "wgrad"]
    print(f"{name:22s}: fwd+dgrad {sum(r['ms'] for r in fd):7.3f} ms | " + " ".join(f"{r['tflops']:.0f}" for r in fd))
lib.tdx_tune_set(b"conv_dbg", 0)
