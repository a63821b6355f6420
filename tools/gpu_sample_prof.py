#!/usr/bin/env python3
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd.diffusion import ForwardProcess, NoiseModel, sample
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 200
torch.manual_seed(0)
m = NoiseModel().cuda().eval()
fp = ForwardProcess(num_timesteps=T)
sample(m, ForwardProcess(num_timesteps=3), "cuda", n_samples=n, use_graph=True, philox_seed=1)
torch.cuda.synchronize(); t0 = time.perf_counter()
x = sample(m, fp, "cuda", n_samples=n, use_graph=True, philox_seed=1)
torch.cuda.synchronize(); print("n", n, "T", T, "ms/step", (time.perf_counter() - t0) / T * 1e3)
