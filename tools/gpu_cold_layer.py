#!/usr/bin/env python3
"""The forward convolution 128 -> 128 at 28x28, B = 256, launched N times back to back after touching 1 GB of
other memory (so the first launch finds nothing of its operands in L2 / MALL).  Prints per-launch times (HIP
events); under `rocprofv3 --pmc ...` the per-dispatch counters of the same launches show what changes between
the cold first launches and the warm later ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd._lib import lib, check

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
B, H, cin, cout = 256, 28, 128, 128
M = B * H * H
x = torch.randn(M * cin, device=dev)
wf = torch.randn(cout * 9 * cin, device=dev) * 0.02
y = torch.empty(M * cout, device=dev)
bias = torch.zeros(cout, device=dev)
stat = torch.empty(lib.tdx_conv3x3_stat_tiles(B, H, H, cin, cout) * 2 * cout, device=dev)
junk = torch.empty(256 << 20, device=dev)   # 1 GiB of floats
fn = lambda: check(lib.tdx_conv3x3_fwd(x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, H, cin, cout, 4,
                                       None, None, None, None, stat.data_ptr(), st))
fn(); torch.cuda.synchronize()
for rnd in range(2):
    junk.fill_(float(rnd))          # evict
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    ev[0].record()
    for i in range(N):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    print("round", rnd, "us per launch:", [round(ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(N)], flush=True)
