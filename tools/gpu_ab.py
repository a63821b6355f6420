#!/usr/bin/env python3
"""A/B of tuning knobs in ONE process (interleaved rounds): training ms/step and
sampling ms/step.  usage: gpu_ab.py "conv_impl=1" "conv_impl=2" ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd._lib import lib
from tiny_diffusion_amd.diffusion import ForwardProcess, NoiseModel, sample
from tiny_diffusion_amd.train import TrainStep

def apply(spec):
    for item in filter(None, spec.split(",")):
        k, v = item.split("=")
        assert lib.tdx_tune_set(k.encode(), int(v)) == 0, k

specs = sys.argv[1:] or ["conv_impl=1", "conv_impl=2"]
torch.manual_seed(0)
m = NoiseModel().cuda().train(); fp = ForwardProcess()
ts = TrainStep(m, fp, philox_seed=1)
x0 = torch.rand(256, 1, 28, 28, device="cuda") * 2 - 1
res = {s: {"train": [], "s16": [], "s64": []} for s in specs}
for rnd in range(3):
    for s in specs:
        apply(s)
        m.train()
        for _ in range(5): ts.step(x0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): ts.step(x0)
        torch.cuda.synchronize(); res[s]["train"].append((time.perf_counter() - t0) / 20 * 1e3)
        for n, key in ((16, "s16"), (64, "s64")):
            fpT = ForwardProcess(num_timesteps=150)
            sample(m, ForwardProcess(num_timesteps=3), "cuda", n_samples=n, use_graph=True, philox_seed=1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sample(m, fpT, "cuda", n_samples=n, use_graph=True, philox_seed=1)
            torch.cuda.synchronize(); res[s][key].append((time.perf_counter() - t0) / 150 * 1e3)
for s in specs:
    r = res[s]
    print(f"{s:28s} train ms/step {min(r['train']):.3f} (all {[round(v,3) for v in r['train']]})  "
          f"sample n16 {min(r['s16']):.3f} n64 {min(r['s64']):.3f} ms/step")
