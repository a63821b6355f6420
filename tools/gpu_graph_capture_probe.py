"""Capture one whole training step (TrainStep(use_graph=True)) with the library's helper streams forked
inside the capture (stream mode 1: three streams, 2: helpers only, -1: the network's default) and replay it; prints OK or dies.
Run once per configuration in its own process:  python tools/gpu_graph_capture_probe.py <net> <streams>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
net, streams = sys.argv[1], sys.argv[2]
import faulthandler

faulthandler.enable()
import torch

from tiny_diffusion_amd.train import TrainStep

if net == "laion":
    from tiny_diffusion_amd.conditional_diffusion_laion import ForwardProcess, NoiseModel
    m = NoiseModel(time_dim=768).cuda().train()
    x = torch.randn(8, 4, 32, 32, device="cuda")
    c = torch.randn(8, 768, device="cuda")
    kw = dict(max_grad_norm=10.0)
else:
    from tiny_diffusion_amd.diffusion import ForwardProcess, NoiseModel
    m = NoiseModel().cuda().train()
    x = torch.rand(16, 1, 28, 28, device="cuda") * 2 - 1
    c = None
    kw = {}
m._stream_mode = int(streams)   # -1 the network's default, 0 one stream, 1 three, 2 helpers only
ts = TrainStep(m, ForwardProcess(), lr=1e-4, use_graph=True, **kw)
for i in range(5):
    loss = ts.step(x, c)
    torch.cuda.synchronize()
    print(f"{net} streams={streams} step {i}: loss {float(loss):.5f} graph={ts._graph is not None}", flush=True)
print(f"{net} streams={streams}: OK")
