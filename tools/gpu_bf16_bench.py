#!/usr/bin/env python3
"""The bf16-mode legs of bench.py on their own (for rocprofv3): MNIST UNet B=256 and the LAION UNet at
64x64 B=256, a few training steps each.   usage: gpu_bf16_bench.py [mnist|laion64|laion32] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tiny_diffusion_amd.train import TrainStep

which = sys.argv[1] if len(sys.argv) > 1 else "mnist"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dt = torch.float32 if os.environ.get("TDX_FP32") else torch.bfloat16
if os.environ.get("TDX_BF16_STORAGE") is not None:   # 0: bf16 MFMA operands only, fp32 tensors (the round-2 form)
    from tiny_diffusion_amd._lib import lib
    assert lib.tdx_tune_set(b"bf16_storage", int(os.environ["TDX_BF16_STORAGE"])) == 0
for knob in ("bf16_ring", "bf16_wgrad_swz", "bf16_wgrad9", "wgrad9_wgs", "bf16_materialize", "bf16_thin"):   # TDX_TUNE_bf16_ring=0 ...: A/B of the round-3 bf16 kernels
    v = os.environ.get("TDX_TUNE_" + knob)
    if v is not None:
        from tiny_diffusion_amd._lib import lib
        assert lib.tdx_tune_set(knob.encode(), int(v)) == 0
torch.manual_seed(0)
if which == "mnist":
    from tiny_diffusion_amd.diffusion import ForwardProcess, NoiseModel
    m = NoiseModel().cuda().train().set_compute_dtype(dt)
    ts = (TrainStep(m, ForwardProcess(), lr=1e-3, use_graph=True) if os.environ.get("TDX_GRAPH") == "1"   # whole step in one HIP graph
          else TrainStep(m, ForwardProcess(), lr=1e-3, philox_seed=1))
    x0 = torch.rand(256, 1, 28, 28, device="cuda") * 2 - 1
    args = (x0,)
else:
    from tiny_diffusion_amd.conditional_diffusion_laion import ForwardProcess, NoiseModel
    hw = 64 if which == "laion64" else 32
    m = NoiseModel(time_dim=768).cuda().train().set_compute_dtype(dt)
    ts = (TrainStep(m, ForwardProcess(), lr=1e-4, use_graph=True, max_grad_norm=10.0) if os.environ.get("TDX_GRAPH") == "1"
          else TrainStep(m, ForwardProcess(), lr=1e-4, philox_seed=1, max_grad_norm=10.0))
    args = (torch.randn(256, 4, hw, hw, device="cuda") * 0.8, torch.randn(256, 768, device="cuda"))
if os.environ.get("TDX_STREAM_MODE") is not None:   # 0: one stream, 2: main + weight-gradient stream, -1: default (three)
    m._stream_mode = int(os.environ["TDX_STREAM_MODE"])
for _ in range(5):
    ts.step(*args)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    ts.step(*args)
torch.cuda.synchronize()
print(f"{which} {dt}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step")
