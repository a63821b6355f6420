#!/usr/bin/env python3
"""Soak test of the round-2 placement of the time / class path's backward (projections behind their pixel
sums, the middle behind the third one, beside the weight-gradient GEMMs): every gradient must be
bit-identical to the one-shot placement after the last stage (knob time_proj_early=0), on fresh models and
plans, with an old plan being destroyed while the new one runs - the flow that once exposed a wrong dW1.

    python tools/gpu_time_early_soak.py [B=64] [iterations=100] [cond=1]"""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiny_diffusion_amd._lib as L

lib = L.lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cond = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
if cond:
    from tiny_diffusion_amd.conditional_diffusion import NoiseModel
else:
    from tiny_diffusion_amd.diffusion import NoiseModel


def fresh():
    torch.manual_seed(1234)
    return NoiseModel().cuda().train()


def grads(m, x, t, y):
    out = m(x, t, y) if cond else m(x, t)
    (out ** 2).mean().backward()


bad = 0
for it in range(iters):
    g = torch.Generator(device="cuda").manual_seed(it)
    x = torch.randn(B, 1, 28, 28, device="cuda", generator=g)
    t = torch.randint(0, 1000, (B,), device="cuda", generator=g)
    y = torch.randint(0, 10, (B,), device="cuda", generator=g)
    lib.tdx_tune_set(b"time_proj_early", 0)
    ref_m = fresh(); grads(ref_m, x, t, y); torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in ref_m.named_parameters()}
    del ref_m; gc.collect()
    lib.tdx_tune_set(b"time_proj_early", 1)
    a = fresh(); grads(a, x, t, y); torch.cuda.synchronize()
    b = fresh()
    grads(b, x, t, y)
    del a; gc.collect()
    torch.cuda.synchronize()
    wrong = [k for k, p in b.named_parameters() if not torch.equal(p.grad, ref[k])]
    bad += bool(wrong)
    if wrong or it % 20 == 0:
        print(f"iter {it}: {'ok' if not wrong else 'BAD ' + str(wrong)}", flush=True)
    del b; gc.collect()
print(f"done: {bad} bad of {iters} (B={B}, cond={cond})")
sys.exit(1 if bad else 0)
