#!/usr/bin/env python3
"""Why does the 2-rank gradient test miss the single-GPU gate?  Same shard configuration as
tests/test_gpu_ddp.py, ONE process, no collective: per-parameter error of (a) the module path and
(b) TrainStep against the fp64 oracle with the GPU's pool routing."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from oracle import ref_cpu as R
from oracle.weights import make_state_dict
from parity_helpers import gpu_pool_routing, grad_precision_failures
from tiny_diffusion_amd.conditional_diffusion import ForwardProcess, NoiseModel
from tiny_diffusion_amd.train import TrainStep


def shard(rank, B=8):
    g = torch.Generator().manual_seed(500 + rank)
    x0 = torch.rand(B, 1, 28, 28, generator=g) * 2 - 1
    noise = torch.randn(B, 1, 28, 28, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    return x0, noise, t, y


B = 8
sd = make_state_dict(6, True)
for rank in (0, 1):
    x0, noise, t, y = shard(rank, B)
    x_t = R.q_sample(R.Schedule(), x0, t, noise)
    for path in ("module", "trainstep"):
        m = NoiseModel(); m.load_state_dict(sd); m = m.cuda().train()
        if path == "module":
            eps = m(x_t.cuda(), t.cuda(), y.cuda())
            F.mse_loss(eps, noise.cuda()).backward()
            got = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
        else:
            ts = TrainStep(m, ForwardProcess(), lr=1e-3)
            ts.step(x0.cuda(), y.cuda(), t=t.cuda(), noise=noise.cuda())
            torch.cuda.synchronize()
            got = {k: v.detach().cpu().clone() for k, v in ts.grad_views.items()}
        cpu_args = (sd, x_t, t, noise, y)
        pidx = gpu_pool_routing(m, B, cpu_args)
        _, _, g32, _ = R.train_step_grads(*cpu_args, pool_idx=pidx)
        _, _, g64, _ = R.train_step_grads(*cpu_args, dtype=torch.float64, pool_idx=pidx)
        bad = grad_precision_failures(got, g32, g64, True)
        worst = sorted(((got[k].double() - g64[k]).norm().item() / max(g64[k].norm().item(), 1e-30), k) for k in g64)[-5:]
        print(f"rank-shard {rank} path {path}: {len(bad)} over the gate; worst {worst}", flush=True)
