#!/usr/bin/env python3
"""Where a workgroup of the 64x64 forward convolution spends its life (diagnostic knob conv_stamp): per workgroup
the 100 MHz counter at its first instruction, at the start and end of its main loop and at the end of its epilogue,
plus HW_ID.  Prints, for each layer shape: workgroups resident per CU (sum of lifetimes / launch span / 256),
in their main loop per CU, and the median prologue / loop / epilogue times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd._lib import lib, check

dev = torch.device("cuda")
st = torch.cuda.current_stream().cuda_stream
stamps = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
lib.tdx_diag_set_buffer(stamps.data_ptr(), stamps.numel() * stamps.element_size())
B = 256
for cin, cout, H in ((128, 128, 28), (64, 128, 28), (256, 256, 14), (512, 512, 7), (256, 64, 32), (64, 64, 32)):
    M = B * H * H
    x = torch.randn(M * cin, device=dev)
    wf = torch.randn(cout * 9 * cin, device=dev) * 0.02
    y = torch.empty(M * cout, device=dev)
    bias = torch.zeros(cout, device=dev)
    tiles = lib.tdx_conv3x3_stat_tiles(B, H, H, cin, cout)
    stat = torch.empty(tiles * 2 * cout, device=dev)
    shape = lib.tdx_conv3x3_tile_shape(B, H, H, cin, cout, 0)
    bm, bn = shape // 1000, shape % 1000
    nwg = ((M + bm - 1) // bm + 7) // 8 * 8 * (cout // bn)
    fn = lambda: check(lib.tdx_conv3x3_fwd(x.data_ptr(), wf.data_ptr(), bias.data_ptr(), y.data_ptr(), B, H, H, cin, cout, 4,
                                           None, None, None, None, stat.data_ptr(), st))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    stamps.zero_()
    lib.tdx_tune_set(b"conv_stamp", 1)
    fn()
    torch.cuda.synchronize()
    lib.tdx_tune_set(b"conv_stamp", 0)
    s = stamps.cpu()[: 8 * nwg].view(-1, 8)
    s = s[s[:, 1] > 0].double()
    entry, l0, l1, end = s[:, 6], s[:, 2], s[:, 3], s[:, 4]
    span = (end.max() - entry.min()) / 100
    life = (end - entry) / 100
    med = lambda v: float(v.median())
    hw = s[:, 7].long()
    cu = ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 0x1) + 32 * ((hw >> 13) & 0x7) + 256 * s[:, 5].long()   # cu, sh, se, xcc
    # per-CU concurrency profile
    import collections
    ev = collections.defaultdict(list)
    for c, a, b in zip(cu.tolist(), entry.tolist(), end.tolist()):
        ev[c].append((a, 1)); ev[c].append((b, -1))
    hist = collections.Counter(); mx = collections.Counter()
    t_lo, t_hi = float(entry.min()), float(end.max())
    for c, e in ev.items():
        e.sort()
        cur, last, m = 0, t_lo, 0
        for t, d in e:
            hist[cur] += t - last; last = t
            cur += d; m = max(m, cur)
        hist[0] += t_hi - last
        mx[m] += 1
    tot = sum(hist.values())
    print("   time share by resident workgroups on a CU:", {k: round(v / tot, 3) for k, v in sorted(hist.items())},
          "| CUs by their maximum:", dict(sorted(mx.items())))
    print(f"{cin}->{cout} @{H} tile {bm}x{bn}: {int(s.shape[0])} wgs, span {float(span):.1f} us, resident/CU {float(life.sum() / span / 256):.2f}, "
          f"in-loop/CU {float(((l1 - l0) / 100).sum() / span / 256):.2f}, prologue {med((l0 - entry) / 100):.1f} us, loop {med((l1 - l0) / 100):.1f}, "
          f"epilogue {med((end - l1) / 100):.1f}, distinct (xcc,se,sh,cu) {int(cu.unique().numel())}", flush=True)
lib.tdx_diag_set_buffer(None, 0)
