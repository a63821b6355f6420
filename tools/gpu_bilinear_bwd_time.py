import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tiny_diffusion_amd._lib as L
lib = L.lib
B = 256
# (Hi, Ho, C, gcs, gcoff)
CASES = [(4, 8, 512, 1024, 0), (8, 16, 256, 512, 0), (16, 32, 128, 256, 0), (7, 8, 512, 1024, 512), (14, 16, 256, 512, 256),
         (28, 32, 128, 256, 128), (32, 28, 64, 64, 0)]
for Hi, Ho, C, gcs, off in CASES:
    g = torch.randn(B, Ho, Ho, gcs, device='cuda')
    gi = torch.empty(B, Hi, Hi, C, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    f = lambda: L.check(lib.tdx_bilinear_ac_bwd(g.data_ptr(), gi.data_ptr(), B, Hi, Hi, Ho, Ho, C, gcs, off, st))
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    mb = (B * Ho * Ho * C + B * Hi * Hi * C) * 4 / 1e6
    print(f"{Hi}->{Ho} C={C}: {us:.1f} us, {mb:.0f} MB, {mb / us / 1e3 * 1e3:.0f} GB/s", flush=True)
