#!/usr/bin/env python3
"""Training steps of the MNIST UNet alone (for rocprofv3 traces and A/B timing):

    python3 tools/gpu_mnist_train.py [B=256] [f32|bf16] [steps=20] [cond=0|1]

The step bench.py's headline (f32) and `mnist_bf16` legs time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd.diffusion import ForwardProcess, NoiseModel
from tiny_diffusion_amd.train import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dtype = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
torch.manual_seed(0)
m = NoiseModel().cuda().train().set_compute_dtype(dtype)
ts = TrainStep(m, ForwardProcess(), lr=1e-3, philox_seed=1234)
x0 = torch.rand(B, 1, 28, 28, device="cuda") * 2 - 1
for _ in range(6):
    ts.step(x0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = ts.step(x0)
torch.cuda.synchronize()
print(f"MNIST B={B} {dtype}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step, loss {loss.item():.4f}", flush=True)
