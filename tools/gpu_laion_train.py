#!/usr/bin/env python3
"""Training steps of the LAION-shaped UNet alone (for rocprofv3 traces and A/B timing):

    python3 tools/gpu_laion_train.py [B=256] [hw=32] [f32|bf16] [steps=20]

The same step bench.py's `laion_unet` leg times (q_sample + fwd + MSE + bwd + clip + Adam, cosine LR)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd.conditional_diffusion_laion import ForwardProcess, NoiseModel
from tiny_diffusion_amd.train import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hw = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dtype = torch.bfloat16 if len(sys.argv) > 3 and sys.argv[3] == "bf16" else torch.float32
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
torch.manual_seed(0)
model = NoiseModel(time_dim=768).cuda().train().set_compute_dtype(dtype)
ts = TrainStep(model, ForwardProcess(), lr=1e-4, philox_seed=99, max_grad_norm=10.0, cosine_T_max=1000, cosine_eta_min=1e-6)
x0 = torch.randn(B, 4, hw, hw, device="cuda") * 0.8
cond = torch.randn(B, 768, device="cuda")
for _ in range(6):
    ts.step(x0, cond)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = ts.step(x0, cond)
torch.cuda.synchronize()
print(f"LAION B={B} hw={hw} {dtype}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step, loss {loss.item():.4f}", flush=True)
