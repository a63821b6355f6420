#!/usr/bin/env python3
"""Print per-parameter gradient errors (GPU vs fp64 oracle, fp32 oracle vs fp64)."""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R
from oracle.weights import make_state_dict

def run(cond, B, training, seed=2):
    if cond:
        from tiny_diffusion_amd.conditional_diffusion import NoiseModel
    else:
        from tiny_diffusion_amd.diffusion import NoiseModel
    sd = make_state_dict(seed, cond)
    g = torch.Generator().manual_seed(17 + B)
    x = torch.randn(B, 1, 28, 28, generator=g); noise = torch.randn(B, 1, 28, 28, generator=g)
    t = torch.randint(0, 1000, (B,), generator=g)
    y = torch.randint(0, 10, (B,), generator=g) if cond else None
    _, _, g32, _ = R.train_step_grads(sd, x, t, noise, y, training=training)
    _, e64, g64, _ = R.train_step_grads(sd, x, t, noise, y, training=training, dtype=torch.float64)
    m = NoiseModel(); m.load_state_dict(sd); m = m.cuda(); m.train(training)
    eps = m(x.cuda(), t.cuda(), y.cuda()) if cond else m(x.cuda(), t.cuda())
    print(f"case cond={cond} B={B} training={training}: eps relerr vs fp64 "
          f"{((eps.detach().cpu().double()-e64).norm()/e64.norm()).item():.2e}")
    F.mse_loss(eps, noise.cuda()).backward()
    for k, p in m.named_parameters():
        n64 = g64[k].norm().item()
        eg = (p.grad.cpu().double() - g64[k]).norm().item() / max(n64, 1e-30)
        ec = (g32[k].double() - g64[k]).norm().item() / max(n64, 1e-30)
        print(f"  {k:28s} gpu {eg:.2e}  cpu32 {ec:.2e}  |g| {n64:.2e}")

if __name__ == "__main__":
    run(True, 4, True)
    run(False, 5, False)
