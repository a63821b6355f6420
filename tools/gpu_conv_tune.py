#!/usr/bin/env python3
"""Per-layer conv kernel timing under forced tile configurations (tuning aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from tiny_diffusion_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for impl in (0, 1):
    lib.tdx_tune_set(b"conv_impl", impl)
    lib.tdx_tune_set(b"conv_tile", 0)
    rows, flop, ms, nl = bench.conv_roofline(B, reps=8)
    fd = [r for r in rows if r["role"] != "wgrad"]
    print(f"conv_impl {impl}: fwd+dgrad {sum(r['ms'] for r in fd):.3f} ms | " + " ".join(f"{r['tflops']:.0f}" for r in fd))
if len(sys.argv) > 2:
    lib.tdx_tune_set(b"conv_impl", int(sys.argv[2]))
res = {}
for tile, name in ((0, "auto"), (1, "128x128"), (2, "128x64"), (3, "64x64")):
    lib.tdx_tune_set(b"conv_tile", tile)
    rows, flop, ms, nl = bench.conv_roofline(B, reps=5)
    res[name] = rows
    print(f"tile {name}: total {ms:.3f} ms  {flop / ms / 1e9:.1f} TF")
lib.tdx_tune_set(b"conv_tile", 0)
print(f"{'layer':28s}" + "".join(f"{n:>10s}" for n in res))
for i, r in enumerate(res["auto"]):
    if r["role"] == "wgrad":
        continue
    print(f"{r['cin']:5d}->{r['cout']:4d}@{r['hw']:2d} {r['role']:6s}      " + "".join(f"{res[n][i]['tflops']:10.1f}" for n in res))
for tgt in (512, 1024, 2048, 4096):
    lib.tdx_tune_set(b"wgrad_target", tgt)
    rows, flop, ms, nl = bench.conv_roofline(B, reps=5)
    w = [r for r in rows if r["role"] == "wgrad"]
    print(f"wgrad_target {tgt}: " + " ".join(f"{r['tflops']:.0f}" for r in w), " sum ms", round(sum(r["ms"] for r in w), 3))
