#!/usr/bin/env python3
"""Reverse-process kernels of the MNIST UNet at a small batch, for a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/samp -- python3 tools/gpu_sample_trace.py [n] [T]

Runs sample(n, T) exactly as bench.py's `sample` leg does (graph replay, in-kernel Philox noise, device-side step
counter).  tools/sample_timeline.py turns the trace into one step's timeline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd.diffusion import ForwardProcess, NoiseModel, sample

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
torch.manual_seed(0)
model = NoiseModel().to("cuda").eval()
fp = ForwardProcess(num_timesteps=T)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    x = sample(model, fp, "cuda", n_samples=n, use_graph=True, philox_seed=7)
    torch.cuda.synchronize()
    print(f"n={n} T={T}: {(time.perf_counter() - t0) * 1e3 / T:.4f} ms/step", flush=True)
assert torch.isfinite(x).all()
