#!/usr/bin/env python3
"""Regression check for the time-path placement knob: gradients with the time / class path enqueued
at backward stage 6 must be bit-identical to the default (last stage), on fresh models + plans and
with an old plan being destroyed while the new one runs (the flow that exposed a miscomputed dW1
with the first version of time_l1_bwd_kernel)."""
import gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tiny_diffusion_amd._lib as L
from tiny_diffusion_amd.diffusion import NoiseModel

lib = L.lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def fresh():
    torch.manual_seed(1234)
    return NoiseModel().cuda().train()


def grads(m, x, t):
    (m(x, t) ** 2).mean().backward()


bad = 0
for it in range(iters):
    g = torch.Generator(device="cuda").manual_seed(it)
    x = torch.randn(B, 1, 28, 28, device="cuda", generator=g)
    t = torch.randint(0, 1000, (B,), device="cuda", generator=g)
    lib.tdx_tune_set(b"time_stage_diag", 14)
    ref_m = fresh(); grads(ref_m, x, t); torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in ref_m.named_parameters()}
    del ref_m; gc.collect()
    lib.tdx_tune_set(b"time_stage_diag", 6)
    a = fresh(); grads(a, x, t); torch.cuda.synchronize()
    b = fresh()
    grads(b, x, t)
    del a; gc.collect()
    torch.cuda.synchronize()
    wrong = [k for k, p in b.named_parameters() if not torch.equal(p.grad, ref[k])]
    bad += bool(wrong)
    print(f"iter {it}: {'ok' if not wrong else 'BAD ' + str(wrong)}", flush=True)
    del b; gc.collect()
lib.tdx_tune_set(b"time_stage_diag", 14)
print(f"done: {bad} bad of {iters}")
sys.exit(1 if bad else 0)
