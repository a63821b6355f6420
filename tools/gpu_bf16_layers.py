#!/usr/bin/env python3
"""Every bf16-mode GEMM launch of one training step in isolation (bf16 storage forms, HIP events): per unit forward,
input gradient and weight gradient (+ slab reduction), with both yardsticks - the bf16 matrix peak (2.5 PFLOP/s) and
the HBM time of the launch's algorithmic bytes at 8 TB/s.   usage: gpu_bf16_layers.py [mnist|laion32|laion64] [B]
Inputs are raw, as in the training step since it materialises relu(bn(Y)) (knob bf16_materialize); TDX_ONLOAD=1 applies
BN + ReLU while staging on the units marked in_bn (the step with bf16_materialize = 0).
Environment: TDX_CONV_DBG = ablation bits of the bf16 kernels (16 no MFMA, 32 no LDS staging, 64 no loads in the main
loop: results are wrong, only the times mean something), TDX_NO_REDUCE=1 leaves the slab reduction out of `wgrad`."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from tiny_diffusion_amd._lib import lib, check

which = sys.argv[1] if len(sys.argv) > 1 else "mnist"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if which == "mnist":
    units = [(ci, co, h, ib) for ci, co, h, ib in bench.UNITS]
else:
    s = 2 if which == "laion64" else 1
    inbn = [0, 1, 0, 1, 0, 1, 0, 0, 1, 0, 1, 0, 1]
    units = [(ci, co, h * s, ib) for (ci, co, h), ib in zip(bench.LAION_UNITS, inbn)]
dev = torch.device("cuda")
DBG = int(os.environ.get("TDX_CONV_DBG", "0"))
NO_REDUCE = os.environ.get("TDX_NO_REDUCE") == "1"
if DBG:
    check(lib.tdx_tune_set(b"conv_dbg", DBG))
if "TDX_THIN" in os.environ:        # 0 | 1 | 2: conv3x3_bf16_thin_kernel off | 64 output channels | + 64 input channels
    check(lib.tdx_tune_set(b"bf16_thin", int(os.environ["TDX_THIN"])))
if "TDX_WGRAD_SWZ" in os.environ:   # 0: the round-2 weight-gradient staging
    check(lib.tdx_tune_set(b"bf16_wgrad_swz", int(os.environ["TDX_WGRAD_SWZ"])))
st = torch.cuda.current_stream().cuda_stream
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for cin, cout, H, in_bn in units:
    M = B * H * H
    x = torch.randn(M * cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(M * cout, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
    wf = torch.empty(cout * 9 * cin, dtype=torch.bfloat16, device=dev)
    wd = torch.empty(cout * 9 * cin, dtype=torch.bfloat16, device=dev)
    check(lib.tdx_pack_conv3x3_bf16(w.data_ptr(), wf.data_ptr(), wd.data_ptr(), cout, cin, st))
    out = torch.empty(M * cout, dtype=torch.bfloat16, device=dev)
    gin = torch.empty(M * cin, dtype=torch.bfloat16, device=dev)
    stats = torch.empty(((M + 127) // 128) * 2 * cout, device=dev)
    splits = lib.tdx_conv3x3_wgrad_splits_bf16(B, H, H, cin, cout)
    slabs = torch.empty(splits * cout * 9 * cin, device=dev)
    dw = torch.empty(cout * cin * 9, device=dev)
    bias = torch.zeros(cout, device=dev)
    isc, ish = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.1
    in_bn = in_bn if os.environ.get("TDX_ONLOAD") == "1" else 0
    scp, shp = (isc.data_ptr(), ish.data_ptr()) if in_bn else (None, None)
    fl = 4 | (1 if in_bn else 0)

    def fwd():
        check(lib.tdx_conv3x3_fwd_bf16_io(x.data_ptr(), wf.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, H, cin, cout,
                                          fl, scp, shp, None, None, stats.data_ptr(), 1, st))

    def dgrad():
        check(lib.tdx_conv3x3_fwd_bf16_io(dy.data_ptr(), wd.data_ptr(), None, gin.data_ptr(), B, H, H, cout, cin, 0, None,
                                          None, None, None, None, 1, st))

    def wgrad():
        check(lib.tdx_conv3x3_wgrad_bf16_io(x.data_ptr(), dy.data_ptr(), slabs.data_ptr(), B, H, H, cin, cout,
                                            1 if in_bn else 0, scp, shp, 1, st))
        if not NO_REDUCE:
            check(lib.tdx_conv3x3_wgrad_reduce(slabs.data_ptr(), dw.data_ptr(), splits, cout, cin, st))

    flop = 2.0 * M * 9 * cin * cout
    line = f"{cin:5d}->{cout:4d} @{H:2d} in_bn={in_bn}:"
    for name, fn, byts in (("fwd", fwd, 2.0 * M * (cin + cout)), ("dgrad", dgrad, 2.0 * M * (cin + cout)),
                           ("wgrad", wgrad, 2.0 * M * (cin + cout) + 4.0 * 9 * cin * cout)):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        tot[name] += us
        line += f"  {name} {us:6.1f} us (mfma {flop / 2.5e15 * 1e6:5.1f}, hbm {byts / 8e12 * 1e6:5.1f})"
    print(line + (f"  splits {splits}"), flush=True)
print("sums, us:", {k: round(v, 1) for k, v in tot.items()}, "total", round(sum(tot.values()), 1))
