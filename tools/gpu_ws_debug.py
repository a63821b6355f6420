#!/usr/bin/env python3
"""Which forward-saved workspace tensors change during backward? (none should)"""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.weights import make_state_dict
from tiny_diffusion_amd.conditional_diffusion import NoiseModel
cond, B, seed = True, 4, 2
sd = make_state_dict(seed, cond)
g = torch.Generator().manual_seed(17 + B)
x = torch.randn(B, 1, 28, 28, generator=g); noise = torch.randn(B, 1, 28, 28, generator=g)
t = torch.randint(0, 1000, (B,), generator=g); y = torch.randint(0, 10, (B,), generator=g)
m = NoiseModel(); m.load_state_dict(sd); m = m.cuda().train()
eps = m(x.cuda(), t.cuda(), y.cuda())
plan = list(m._plans.values())[0]
names = ["x0"] + [f"Y{i}" for i in range(13)] + [f"ss{i}" for i in range(13)] + ["e1p", "e2p", "e3p", "cat3", "cat2", "cat1", "d1a", "emb", "t1", "t2", "t3"]
torch.cuda.synchronize()
before = {n: plan.tensor(n).clone() for n in names}
F.mse_loss(eps, noise.cuda()).backward()
torch.cuda.synchronize()
for n in names:
    a, b = before[n], plan.tensor(n)
    nd = int((a != b).sum())
    if nd:
        idx = (a != b).nonzero().flatten()
        print(f"{n}: {nd} of {a.numel()} elements changed during backward; first idx {idx[:8].tolist()} last {idx[-3:].tolist()}")
print("done")
