#!/usr/bin/env python3
"""Run the backward stage by stage and compare internal gradient buffers with the
fp64 oracle's intermediate gradients."""
import os, sys
import torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_cpu as R
from oracle.weights import make_state_dict
from tiny_diffusion_amd.conditional_diffusion import NoiseModel
cond, B, seed = True, 4, 2
sd = make_state_dict(seed, cond)
g = torch.Generator().manual_seed(17 + B)
x = torch.randn(B, 1, 28, 28, generator=g); noise = torch.randn(B, 1, 28, 28, generator=g)
t = torch.randint(0, 1000, (B,), generator=g); y = torch.randint(0, 10, (B,), generator=g)

dtype = torch.float64
params, buffers = R.split_state(sd)
params = {k: v.to(dtype).requires_grad_(True) for k, v in params.items()}
buffers = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in buffers.items()}
store = {}
orig = F.conv2d
def conv_hook(inp, w, b=None, padding=0):
    out = orig(inp, w, b, padding=padding)
    for k, v in params.items():
        if v is w:
            out.retain_grad()
            if inp.requires_grad: inp.retain_grad()
            store[k[:-7]] = (inp, out)
    return out
F.conv2d = conv_hook
taps = {}
eps = R.unet_forward(params, buffers, x.to(dtype), t, y, training=True, taps=taps)
F.conv2d = orig
for v in taps.values():
    if v.requires_grad: v.retain_grad()
F.mse_loss(eps, noise.to(dtype)).backward()

def nchw(v, Bn, H, Cc): return v.view(Bn, H, H, Cc).permute(0, 3, 1, 2)
def rel(a, b): return ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()

m = NoiseModel(); m.load_state_dict(sd); m = m.cuda().train()
out, plan, mode = m._run_forward(x.cuda(), t.cuda(), y.cuda())
d_out = (2.0 / out.numel()) * (out - noise.cuda())
flat, views = m._grad_buffers(out.device)
units = ["enc1.0", "enc1.3", "enc2.0", "enc2.3", "enc3.0", "enc3.3", "bottleneck.0", "dec3.0", "dec3.3", "dec2.0", "dec2.3", "dec1.0", "dec1.3"]
shape = {"enc1": (28, 128), "enc2": (14, 256), "enc3": (7, 512), "bott": (4, 512), "dec3": (8, 256), "dec2": (16, 128), "dec1": (32, 64)}
g_of = ["G2", "G1", "G1", "G2", "G2", "G1", "G1", "G1", "G2", "G2", "G1", "G1", "G2"]
# stage s (1..13) handles unit 13-s; before it runs, g_of[unit] holds g wrt the unit's activation
for s in range(0, 14):
    m._run_backward(plan, d_out, views, s, s + 1)
    torch.cuda.synchronize()
    if s == 13: break
    u = 12 - s           # next unit to be processed; its activation gradient is ready now
    name = units[u]
    H, Cc = shape[name[:4]]
    buf = plan.tensor(g_of[u])[: B * H * H * Cc]
    # oracle: gradient w.r.t. the unit's post-ReLU activation = grad of the input of the NEXT conv for x.0 units,
    # or the stage output tap for x.3 units / bottleneck
    stage = name.split(".")[0]
    if name.endswith(".0") and stage != "bottleneck":
        ref = store[stage + ".3"][0].grad
    else:
        ref = taps[{"enc1": "e1", "enc2": "e2", "enc3": "e3", "bottleneck": "b", "dec3": "d3", "dec2": "d2", "dec1": "d1"}[stage]].grad
    act = {"0": None}
    # activation value of this unit (fp64 oracle) to restrict the comparison to unmasked positions
    a64 = store[stage + ".3"][0] if (name.endswith(".0") and stage != "bottleneck") else taps[{"enc1": "e1", "enc2": "e2", "enc3": "e3", "bottleneck": "b", "dec3": "d3", "dec2": "d2", "dec1": "d1"}[stage]]
    mask = (a64.detach() > 0)
    got = nchw(buf, B, H, Cc).double().cpu()
    em = ((got - ref)[mask].norm() / ref[mask].norm()).item()
    print(f"after stage {s:2d}: g(act of {name:13s}) rel err {rel(nchw(buf, B, H, Cc), ref):.2e}  on unmasked positions {em:.2e}")
    # dy after this unit's BN backward will be checked at the next iteration via store[name][1].grad
    if s in (8, 10, 12, 1, 3, 5):
        pu = 12 - (s - 1)
        pname = units[pu]; pH, pC = shape[pname[:4]]
        dy = plan.tensor(g_of[pu])[: B * pH * pH * pC]
        refdy = store[pname][1].grad
        print(f"      dy of {pname}: rel err {rel(nchw(dy, B, pH, pC), refdy):.2e}")
    if False:
        pu = 12 - (s - 1)
        pname = units[pu]; pH, pC = shape[pname[:4]]
        dy = plan.tensor(g_of[pu])[: B * pH * pH * pC]
        # NOTE: g_of[pu] may have been overwritten by later ops in stage s; only meaningful if not reused

print("---- second pass: GS2 integrity and accuracy")
out, plan, mode = m._run_forward(x.cuda(), t.cuda(), y.cuda())
d_out = (2.0 / out.numel()) * (out - noise.cuda())
m._run_backward(plan, d_out, views, 0, 5); torch.cuda.synchronize()
gs2_a = plan.tensor("GS2").clone()
sref = (taps["e2"] + taps["t2"]).detach().requires_grad_(True)
R.bilinear_ac(sref, (16, 16)).backward(taps["e2a"].grad)
print("GS2 after stage 4 vs oracle:", rel(nchw(gs2_a, B, 14, 256), sref.grad))
m._run_backward(plan, d_out, views, 5, 9); torch.cuda.synchronize()
gs2_b = plan.tensor("GS2").clone()
print("GS2 changed between stage 4 and stage 9:", int((gs2_a != gs2_b).sum()))
g1 = plan.tensor("G1")[: B * 49 * 256].clone()   # g(e2p) before stage 9? no: stage 9 not yet run
m._run_backward(plan, d_out, views, 9, 10); torch.cuda.synchronize()
ge2p = plan.tensor("G1")[: B * 49 * 256]
print("g(e2p) vs oracle:", rel(nchw(ge2p, B, 7, 256), taps["e2p"].grad))
ga3 = plan.tensor("G2")[: B * 196 * 256]
ref = taps["e2"].grad
got = nchw(ga3, B, 14, 256).double().cpu()
diff = (got - ref)
print("g(e2) total err", (diff.norm() / ref.norm()).item())
# decompose: pooled part only
pooled_ref = ref - sref.grad
pooled_got = got - nchw(gs2_b, B, 14, 256).double().cpu()
print("pool-routed part err:", ((pooled_got - pooled_ref).norm() / pooled_ref.norm()).item())
d = (pooled_got - pooled_ref).abs()
top = d.flatten().topk(12).indices
a32 = None
for fi in top.tolist():
    n_, rem = divmod(fi, 256 * 14 * 14); c_, rem = divmod(rem, 196); h_, w_ = divmod(rem, 14)
    win = taps["e2"].detach()[n_, c_, (h_ // 2) * 2:(h_ // 2) * 2 + 2, (w_ // 2) * 2:(w_ // 2) * 2 + 2]
    gpuY = plan.tensor("Y3").view(B, 14, 14, 256)[n_, (h_ // 2) * 2:(h_ // 2) * 2 + 2, (w_ // 2) * 2:(w_ // 2) * 2 + 2, c_].cpu()
    ss = plan.tensor("ss3").cpu()
    gwin = torch.relu(gpuY * ss[c_] + ss[256 + c_])
    print("  ", (n_, c_, h_, w_), "got %.4e ref %.4e" % (pooled_got[n_, c_, h_, w_].item(), pooled_ref[n_, c_, h_, w_].item()),
          "win64", ["%.9g" % v for v in win.flatten().tolist()], "gpu win", ["%.9g" % v for v in gwin.flatten().tolist()])
