#!/usr/bin/env python3
"""Isolate one conv/BN unit on REAL network data: oracle (fp64) intermediates for
enc2.3 -> run libtdx bn_relu_bwd / wgrad / dgrad on them and compare."""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R
from oracle.weights import make_state_dict
import tiny_diffusion_amd._lib as L
lib, check = L.lib, L.check

cond, B, seed = True, 4, 2
sd = make_state_dict(seed, cond)
g = torch.Generator().manual_seed(17 + B)
x = torch.randn(B, 1, 28, 28, generator=g); noise = torch.randn(B, 1, 28, 28, generator=g)
t = torch.randint(0, 1000, (B,), generator=g); y = torch.randint(0, 10, (B,), generator=g)

def run(dtype):
    params, buffers = R.split_state(sd)
    params = {k: v.to(dtype).requires_grad_(True) for k, v in params.items()}
    buffers = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in buffers.items()}
    # manual replay of enc1/enc2 to grab the pre-BN tensors of enc2.0 / enc2.3
    store = {}
    orig = F.conv2d
    def conv_hook(inp, w, b=None, padding=0):
        out = orig(inp, w, b, padding=padding)
        for k, v in params.items():
            if v is w:
                out.retain_grad(); inp.retain_grad() if inp.requires_grad else None
                store[k] = (inp, out)
        return out
    F.conv2d = conv_hook
    try:
        eps = R.unet_forward(params, buffers, x.to(dtype), t, y, training=True)
    finally:
        F.conv2d = orig
    loss = F.mse_loss(eps, noise.to(dtype)); loss.backward()
    return params, store

p64, s64 = run(torch.float64)
p32, s32 = run(torch.float32)
def nhwc(v): return v.permute(0, 2, 3, 1).contiguous()
def nchw(v): return v.permute(0, 3, 1, 2).contiguous()
def rel(a, b): return ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
st = torch.cuda.current_stream().cuda_stream

for unit, prev in (("enc2.3", "enc2.0"), ("enc3.3", "enc3.0"), ("enc1.3", "enc1.0")):
    a_in64, y64 = s64[unit + ".weight"]; _, yprev64 = s64[prev + ".weight"]
    a_in32, y32 = s32[unit + ".weight"]
    dy64 = y64.grad
    print(unit, "cpu32 dy vs fp64:", rel(y32.grad, dy64), " a_in:", rel(a_in32.detach(), a_in64.detach()))
    # dy is the gradient w.r.t. the pre-BN conv output.  Feed the fp64 dy (rounded) to wgrad/dgrad
    Bn, Cc, H, _ = y64.shape
    cin = a_in64.shape[1]
    dyd = nhwc(dy64.float()).cuda(); ain = nhwc(a_in64.detach().float()).cuda()
    w = p64[unit + ".weight"].detach().float().cuda()
    wf = torch.empty(w.numel(), device="cuda"); wd = torch.empty(w.numel(), device="cuda")
    check(lib.tdx_pack_conv3x3(w.data_ptr(), wf.data_ptr(), wd.data_ptr(), Cc, cin, st))
    splits = lib.tdx_conv3x3_wgrad_splits(Bn, H, H, cin, Cc)
    slabs = torch.empty(splits * w.numel(), device="cuda"); dw = torch.empty_like(w)
    check(lib.tdx_conv3x3_wgrad(ain.data_ptr(), dyd.data_ptr(), slabs.data_ptr(), Bn, H, H, cin, Cc, 0, None, None, st))
    check(lib.tdx_conv3x3_wgrad_reduce(slabs.data_ptr(), dw.data_ptr(), splits, Cc, cin, st))
    print("   wgrad(raw a_in, exact dy) vs fp64:", rel(dw, p64[unit + ".weight"].grad), " cpu32:", rel(p32[unit + ".weight"].grad, p64[unit + ".weight"].grad))
    gin = torch.empty(Bn, H, H, cin, device="cuda")
    check(lib.tdx_conv3x3_fwd(dyd.data_ptr(), wd.data_ptr(), None, gin.data_ptr(), Bn, H, H, Cc, cin, 0, None, None, None, None, None, st))
    print("   dgrad(exact dy) vs fp64:", rel(nchw(gin), a_in64.grad))
    # BN backward on real data: need g wrt activation = grad of relu output; reconstruct from autograd:
    # a = relu(bn(y)); use fp64 autograd to get g_a by re-running BN+ReLU alone
    yy = y64.detach().clone().requires_grad_(True)
    bnw = p64[unit[:-1] + str(int(unit[-1]) + 1) + ".weight"].detach(); bnb = p64[unit[:-1] + str(int(unit[-1]) + 1) + ".bias"].detach()
    mean = yy.mean((0, 2, 3)); var = (yy - mean.view(1, -1, 1, 1)).pow(2).mean((0, 2, 3))
    a = F.relu((yy - mean.view(1, -1, 1, 1)) * (torch.rsqrt(var + 1e-5) * bnw).view(1, -1, 1, 1) + bnb.view(1, -1, 1, 1))
    # least-squares-free: g_a is not directly available; instead check BN bwd consistency with a random g_a
    ga = torch.randn(a.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    a.backward(ga)
    rows = Bn * H * H
    yd = nhwc(y64.detach().float()).cuda()
    rstd = torch.rsqrt(var.detach() + 1e-5)
    sc = (bnw * rstd).float().cuda(); sh = (bnb - mean.detach() * bnw * rstd).float().cuda()
    mu = mean.detach().float().cuda(); rs = rstd.float().cuda()
    gd = nhwc(ga.float()).cuda()
    scr = torch.empty(lib.tdx_bn_relu_bwd_scratch_floats(rows, Cc), device="cuda")
    dg, db, dbias = (torch.empty(Cc, device="cuda") for _ in range(3))
    gam = bnw.float().cuda()
    check(lib.tdx_bn_relu_bwd(gd.data_ptr(), yd.data_ptr(), rows, Cc, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), rs.data_ptr(),
                              gam.data_ptr(), dg.data_ptr(), db.data_ptr(), dbias.data_ptr(), scr.data_ptr(), 1, st))
    print("   bn_relu_bwd(real y, random g) dy vs fp64:", rel(nchw(gd), yy.grad), " min var:", var.min().item(),
          " max |mean|/std:", (mean.abs() / var.sqrt()).max().item())
