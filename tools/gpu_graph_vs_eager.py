#!/usr/bin/env python3
"""Eager vs captured TrainStep on the MNIST network: gradients and parameters after each step, per parameter
(diagnostic behind tests/test_gpu_unet.py::test_train_step_graph_capture_three_streams).
usage: gpu_graph_vs_eager.py [stream_mode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle.weights import make_state_dict
from tiny_diffusion_amd.diffusion import ForwardProcess, NoiseModel
from tiny_diffusion_amd.train import TrainStep

mode = int(sys.argv[1]) if len(sys.argv) > 1 else -1
fp = ForwardProcess()
g = torch.Generator().manual_seed(21)
xs = [(torch.rand(16, 1, 28, 28, generator=g) * 2 - 1).cuda() for _ in range(4)]
runs = {}
for tag, use_graph in (("eager", False), ("graph", True), ("graph2", True), ("eager2", False)):
    m = NoiseModel(); m.load_state_dict(make_state_dict(5, False)); m = m.cuda().train()
    m._stream_mode = mode
    ts = TrainStep(m, fp, lr=1e-4, use_graph=use_graph)
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    rec = []
    for x in xs:
        loss = float(ts.step(x))
        torch.cuda.synchronize()
        rec.append((loss, ts.flat_grad.clone(), ts.flat_param.clone()))
    runs[tag] = (rec, dict(ts.offsets))
for a, b in (("eager", "eager2"), ("graph", "graph2"), ("eager", "graph")):
    ra, off = runs[a]
    rb, _ = runs[b]
    for i, ((la, ga, pa), (lb, gb, pb)) in enumerate(zip(ra, rb)):
        gd = sorted(((ga[lo:hi] - gb[lo:hi]).abs().max().item() / max(ga[lo:hi].abs().max().item(), 1e-30), k) for k, (lo, hi) in off.items())[-2:]
        print(f"{a} vs {b} step {i}: loss {la:.7f} / {lb:.7f}; grads equal {torch.equal(ga, gb)}, params equal {torch.equal(pa, pb)}; "
              f"worst relative gradient difference {gd}")
