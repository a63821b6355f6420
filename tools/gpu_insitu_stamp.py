#!/usr/bin/env python3
"""Per-workgroup timeline of ONE convolution launch inside the real training step (diagnostic stamps,
knob conv_stamp = M selects the training-forward launch with M output pixels, -M the dgrad launch)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd._lib import lib
from tiny_diffusion_amd.diffusion import ForwardProcess, NoiseModel
from tiny_diffusion_amd.train import TrainStep

sel = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.manual_seed(0)
m = NoiseModel().cuda().train()
ts = TrainStep(m, ForwardProcess(), philox_seed=1)
x0 = torch.rand(256, 1, 28, 28, device="cuda") * 2 - 1
stamps = torch.zeros(16 * 8 * 8192, dtype=torch.int64, device="cuda")
lib.tdx_diag_set_buffer(stamps.data_ptr(), stamps.numel() * stamps.element_size())
for _ in range(30):
    ts.step(x0)
torch.cuda.synchronize()
lib.tdx_tune_set(b"conv_stamp", sel)
ts.step(x0)
torch.cuda.synchronize()
lib.tdx_tune_set(b"conv_stamp", 0)
if sel == 2:   # every training-forward conv of one step, absolute 100 MHz times
    allr = stamps.cpu().view(16, 8192, 8).double()
    base = None
    prev_end = None
    for k in range(16):
        r = allr[k]; r = r[r[:, 1] > 0]
        if not len(r):
            continue
        if base is None:
            base = r[:, 2].min()
        st_, en_, ep_ = (r[:, 2].min() - base) / 100, (r[:, 3].max() - base) / 100, (r[:, 4].max() - base) / 100
        gap = "" if prev_end is None else f" gap since previous conv's last workgroup {float(st_ - prev_end):7.1f} us"
        print(f"conv #{k:2d}: {len(r):5d} wgs, first loop start {float(st_):8.1f}, last loop end {float(en_):8.1f}, last epilogue end "
              f"{float(ep_):8.1f} (in-kernel {float(ep_ - st_):6.1f} us){gap}")
        prev_end = ep_
    lib.tdx_diag_set_buffer(None, 0)
    sys.exit(0)
s = stamps.cpu().view(-1, 8).double()
s = s[s[:, 1] > 0]
t0 = s[:, 2].min()
print(f"selected M={sel}: {s.shape[0]} workgroups")
print("loop start spread us", float((s[:, 2].max() - t0) / 100))
print("loop us min/med/max", [float(v / 100) for v in (s[:, 1].min(), s[:, 1].median(), s[:, 1].max())])
print("clock MHz med", float((s[:, 0] / s[:, 1] * 100).median()))
print("last loop end us", float((s[:, 3].max() - t0) / 100), "last epilogue end us", float((s[:, 4].max() - t0) / 100))
xcc = s[:, 5].long()
for x in range(8):
    sel_ = s[xcc == x]
    if len(sel_):
        print(f"  xcc {x}: n={len(sel_)} loop med {float(sel_[:, 1].median() / 100):.1f} us start med {float((sel_[:, 2].median() - t0) / 100):.1f}")
lib.tdx_diag_set_buffer(None, 0)
