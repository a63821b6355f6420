#!/usr/bin/env python3
"""The convolutions of ONE reverse step at n samples (default 16), as sample() launches them (inference epilogue,
split-K where the planner asks for it), with the per-workgroup stamps of knob conv_stamp: launch span, and the
median prologue (first instruction -> main loop), main loop and epilogue of a workgroup.  Answers what a 16-47 us
launch of 9-36 K-tiles spends outside its MFMA loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd._lib import lib, check

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
stamps = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
lib.tdx_diag_set_buffer(stamps.data_ptr(), stamps.numel() * stamps.element_size())
UNITS = [(64, 128, 28), (128, 128, 28), (128, 256, 14), (256, 256, 14), (256, 512, 7), (512, 512, 7), (512, 512, 4),
         (1024, 256, 8), (256, 256, 8), (512, 128, 16), (128, 128, 16), (256, 64, 32), (64, 64, 32)]
tot_span = tot_ev = 0.0
for cin, cout, H in UNITS:
    M = n * H * H
    x = torch.randn(M * cin, device=dev)
    wf = torch.randn(cout * 9 * cin, device=dev) * 0.02
    y = torch.empty(M * cout, device=dev)
    bias = torch.zeros(cout, device=dev)
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev) * 0.1
    need = max(lib.tdx_conv3x3_splitk_scratch_floats(n, H, H, cin, cout), 1)
    scratch = torch.empty(need, device=dev)
    fn = lambda: check(lib.tdx_conv3x3_fwd_splitk(x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), n, H, H, cin,
                                                  cout, 2, None, None, sc.data_ptr(), sh.data_ptr(), scratch.data_ptr(),
                                                  need, st))
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ev_us = e0.elapsed_time(e1) / 20 * 1e3
    stamps.zero_()
    lib.tdx_tune_set(b"conv_stamp", 1)
    fn()
    torch.cuda.synchronize()
    lib.tdx_tune_set(b"conv_stamp", 0)
    s = stamps.cpu().view(-1, 8)
    s = s[s[:, 1] > 0].double()
    entry, l0, l1, end = s[:, 6], s[:, 2], s[:, 3], s[:, 4]
    span = float((end.max() - entry.min()) / 100)
    med = lambda v: float(v.median())
    flop = 2.0 * M * 9 * cin * cout
    tot_span += span; tot_ev += ev_us
    print(f"{cin:5d}->{cout:4d} @{H:2d}: launch+reduce {ev_us:6.1f} us by events ({flop / ev_us / 1e6:5.1f} TFLOP/s), conv span {span:5.1f} us, "
          f"{int(s.shape[0])} stamped wgs: prologue {med((l0 - entry) / 100):4.1f} us, loop {med((l1 - l0) / 100):5.1f}, "
          f"epilogue {med((end - l1) / 100):4.1f}; first entry -> first loop start {float((l0.min() - entry.min()) / 100):4.1f}", flush=True)
print(f"sum over the 13 convolutions: {tot_ev:.1f} us by events, {tot_span:.1f} us of conv spans")
lib.tdx_diag_set_buffer(None, 0)
