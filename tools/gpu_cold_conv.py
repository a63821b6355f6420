#!/usr/bin/env python3
"""Hot vs cold caches for the long-K, few-pixel convolutions (the 4x4 bottleneck of the MNIST UNet at
B=256: 4096 pixels, K = 4608, 9.4 MB of weights): the roofline leg of bench.py re-runs a launch on the same
buffers (weights in L2 / Infinity Cache), the training step meets them 2 ms and 1 GB of traffic after the
pack kernel wrote them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd._lib import lib, check

dev = torch.device("cuda"); st = torch.cuda.current_stream().cuda_stream
B = 256
big_a = torch.empty(1 << 28, device=dev); big_b = torch.empty(1 << 28, device=dev)   # 1 GiB each


def timed(fn, flush, warm_w=None):
    ts = []
    for _ in range(6):
        if flush:
            big_b.copy_(big_a)             # 2 GiB of traffic: evicts L2 and the 256 MiB Infinity Cache
        if warm_w is not None:
            warm_w.sum()                   # touch the weights again (they fit the Infinity Cache, not L2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts[1:]), sorted(ts[1:])[len(ts[1:]) // 2]


for cin, cout, H in ((512, 512, 4), (256, 512, 7), (512, 512, 7), (1024, 256, 8), (128, 128, 28)):
    M = B * H * H
    x = torch.randn(M * cin, device=dev); wf = torch.randn(cout * 9 * cin, device=dev) * 0.02
    y = torch.empty(M * cout, device=dev); bias = torch.zeros(cout, device=dev)
    stat = torch.empty(lib.tdx_conv3x3_stat_tiles(B, H, H, cin, cout) * 2 * cout, device=dev)
    fn = lambda: check(lib.tdx_conv3x3_fwd(x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, H, cin, cout,
                                           4, None, None, None, None, stat.data_ptr(), st))
    hot = timed(fn, False); cold = timed(fn, True); warm = timed(fn, True, wf)
    print(f"fwd {cin}->{cout} @{H}x{H} tile {lib.tdx_conv3x3_tile_shape(B, H, H, cin, cout, 0)}: hot {hot[0]:.0f}/{hot[1]:.0f} us, "
          f"cold {cold[0]:.0f}/{cold[1]:.0f} us, cold but weights re-touched {warm[0]:.0f}/{warm[1]:.0f} us (min/median)", flush=True)
