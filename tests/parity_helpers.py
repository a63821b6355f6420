"""Shared parity helpers of the GPU tests (test infrastructure): the max-pool routing the GPU
forward chose, and the fp64-calibrated bound for parameter gradients."""
import numpy as np
import torch

from oracle import ref_cpu as R


def rel_mse(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b) ** 2).mean().item() / max((b**2).mean().item(), 1e-30)


def is_pre_bn_bias(key):
    stage, idx, kind = (key.split(".") + ["", ""])[:3]
    return kind == "bias" and idx in ("0", "3") and stage[:3] in ("enc", "dec", "bot")


def _plan_of(m, B):
    return [p for key, p in m._plans.items() if key[1] == B][0]


def gpu_pool_idx_raw(m, B):
    """Arg-max index of every max-pool window exactly as the GPU forward decided it (no check)."""
    plan = _plan_of(m, B)
    out = {}
    for name, unit, H, Cc in (("e1", 1, 28, 128), ("e2", 3, 14, 256), ("e3", 5, 7, 512)):
        Y = plan.tensor(f"Y{unit}").view(B, H, H, Cc)
        ss = plan.tensor(f"ss{unit}")
        a = torch.relu(torch.addcmul(ss[Cc:2 * Cc], Y, ss[:Cc])).permute(0, 3, 1, 2).cpu()
        out[name] = R.pool_windows(a).argmax(dim=-1, keepdim=True)
    return out


def gpu_relu_masks_raw(m, B):
    """Active set of every ReLU exactly as the GPU forward decided it (no check)."""
    plan = _plan_of(m, B)
    masks = {}
    for u, (name, (H, Cc)) in enumerate(zip(UNIT_BN, UNIT_SHAPE)):
        Y = plan.tensor(f"Y{u}").view(B, H, H, Cc)
        ss = plan.tensor(f"ss{u}")
        masks[name] = (torch.addcmul(ss[Cc:2 * Cc], Y, ss[:Cc]).permute(0, 3, 1, 2) > 0).cpu()
    return masks


def check_choices(pool_idx, masks, cpu_args, training=True, pool_tol=1e-4, relu_tol=1e-5):
    """The sub-gradient choices of an implementation may differ from the exact (fp64) ones only at
    ties: a pooling window whose picked entry is within ``pool_tol`` of the maximum, a ReLU input
    within ``relu_tol`` of 0 (both relative to the tensor's RMS).  Returns the number of flips."""
    sd, x, t, noise, y = cpu_args
    p64, b64 = R.split_state(sd)
    p64 = {k: v.double() for k, v in p64.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in b64.items()}
    taps = {}
    with torch.no_grad():
        R.unet_forward(p64, b64, x.double(), t, y, training=training, taps=taps, pool_idx=pool_idx)
    flips = {}
    for name, idx in (pool_idx or {}).items():
        w64 = R.pool_windows(taps[name])
        differ = (idx != w64.argmax(dim=-1, keepdim=True)).squeeze(-1)
        if differ.any():
            rms = taps[name].pow(2).mean().sqrt().item()
            picked = torch.gather(w64, -1, idx)[differ].squeeze(-1)
            best = w64[differ].max(dim=-1).values
            gap = (best - picked) / torch.maximum(best.abs(), torch.tensor(rms, dtype=best.dtype))
            assert gap.max().item() < pool_tol, (name, int(differ.sum()), gap.max().item())
            flips[name] = int(differ.sum())
    for name, mask in (masks or {}).items():
        exact = taps["prebn_act:" + name]
        differ = mask != (exact > 0)
        if differ.any():
            rms = exact.pow(2).mean().sqrt().item()
            worst = (exact[differ].abs().max() / rms).item()
            assert worst < relu_tol, (name, int(differ.sum()), worst)
            flips[name] = int(differ.sum())
    return flips


def gpu_pool_routing(m, B, cpu_args, training=True):
    """Arg-max index of every max-pool window as the GPU forward decided it (from its own
    pre-BN tensors and scale/shift), checked against the exact routing: they may differ
    only where the two largest entries of a window agree to 1e-4 (an fp32 coin flip,
    either choice being a valid sub-gradient of max)."""
    plan = [p for (dev, b), p in m._plans.items() if b == B][0]
    taps = {}
    sd, x, t, noise, y = cpu_args
    p64, b64 = R.split_state(sd)
    p64 = {k: v.double() for k, v in p64.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in b64.items()}
    with torch.no_grad():
        R.unet_forward(p64, b64, x.double(), t, y, training=training, taps=taps)
    out = {}
    for name, unit, H, Cc in (("e1", 1, 28, 128), ("e2", 3, 14, 256), ("e3", 5, 7, 512)):
        Y = plan.tensor(f"Y{unit}").view(B, H, H, Cc)
        ss = plan.tensor(f"ss{unit}")
        a = torch.relu(torch.addcmul(ss[Cc:2 * Cc], Y, ss[:Cc])).permute(0, 3, 1, 2).cpu()
        win = R.pool_windows(a)
        idx = win.argmax(dim=-1, keepdim=True)
        w64 = R.pool_windows(taps[name])
        idx64 = w64.argmax(dim=-1, keepdim=True)
        differ = (idx != idx64).squeeze(-1)
        if differ.any():
            # a tie = the exact values of the entry the GPU picked and of the exact arg-max agree to 1e-4
            # of the larger of |value| and the tensor's RMS (relative to the value alone, a window whose
            # entries are all ~0 after the ReLU - exact maximum 1e-9 - would count as a gap of 100 %)
            rms = taps[name].pow(2).mean().sqrt().item()
            picked = torch.gather(w64, -1, idx)[differ].squeeze(-1)
            best = w64[differ].max(dim=-1).values
            gap = (best - picked) / torch.maximum(best.abs(), torch.tensor(rms, dtype=best.dtype))
            assert gap.max().item() < 1e-4, (name, int(differ.sum()), gap.max().item())
        out[name] = idx
    return out


UNIT_BN = ("enc1.1", "enc1.4", "enc2.1", "enc2.4", "enc3.1", "enc3.4", "bottleneck.1",
           "dec3.1", "dec3.4", "dec2.1", "dec2.4", "dec1.1", "dec1.4")
UNIT_SHAPE = ((28, 128), (28, 128), (14, 256), (14, 256), (7, 512), (7, 512), (4, 512),
              (8, 256), (8, 256), (16, 128), (16, 128), (32, 64), (32, 64))


def gpu_relu_masks(m, B, cpu_args, training=True, pool_idx=None, tol=1e-5):
    """Active set of every ReLU as the GPU forward decided it (sign of y*scale+shift from its own
    pre-BN tensors), keyed by BatchNorm module name for ``R.unet_forward(relu_masks=...)``.
    Checked against the exact (fp64) pre-activations: the sets may differ only where the exact
    normalised value is within ``tol`` of 0 relative to the layer's RMS - there the activation is 0
    to rounding either way and on/off are both valid sub-gradients."""
    plan = [p for (dev, b), p in m._plans.items() if b == B][0]
    sd, x, t, noise, y = cpu_args
    p64, b64 = R.split_state(sd)
    p64 = {k: v.double() for k, v in p64.items()}
    b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in b64.items()}
    taps = {}
    with torch.no_grad():
        R.unet_forward(p64, b64, x.double(), t, y, training=training, taps=taps, pool_idx=pool_idx)
    masks, flips = {}, {}
    for u, (name, (H, Cc)) in enumerate(zip(UNIT_BN, UNIT_SHAPE)):
        Y = plan.tensor(f"Y{u}").view(B, H, H, Cc)
        ss = plan.tensor(f"ss{u}")
        act = torch.addcmul(ss[Cc:2 * Cc], Y, ss[:Cc]).permute(0, 3, 1, 2).cpu()
        mask = act > 0
        exact = taps["prebn_act:" + name]
        differ = mask != (exact > 0)
        if differ.any():
            rms = exact.pow(2).mean().sqrt().item()
            worst = (exact[differ].abs().max() / rms).item()
            assert worst < tol, (name, int(differ.sum()), worst)
            flips[name] = int(differ.sum())
        masks[name] = mask
    return masks, flips


def grad_precision_failures(got, g32, g64, training, k_factor=10.0, floor=1e-4):
    """Per-parameter: ||g_gpu - g64|| / ||g64|| must be within k_factor x the fp32 CPU
    oracle's own distance from the fp64 ground truth (floor 1e-4: an fp32 MFMA dot product
    is one sequential fma chain, error ~sqrt(K) eps of sum|a*b|, and weight gradients
    cancel heavily - observed worst case 6e-5 on enc1.0.weight in eval mode).  A train-mode
    BatchNorm over a nearly constant channel multiplies rounding noise by up to
    1/sqrt(eps) ~ 300; that noise is a property of fp32 evaluation of this network
    (the reference has it too), so the bound is calibrated per case, not fixed."""
    errs_cpu = {}
    for k in g64:
        n64 = g64[k].norm().item()
        errs_cpu[k] = (g32[k].double() - g64[k]).norm().item() / max(n64, 1e-30)
    usable = [e for k, e in errs_cpu.items() if not (training and is_pre_bn_bias(k))]
    med = float(np.median(usable))
    bad = []
    for k, g in got.items():
        if training and is_pre_bn_bias(k):
            continue  # exactly-zero true gradient: both sides are pure rounding noise
        n64 = g64[k].norm().item()
        err = (g.detach().double().cpu() - g64[k]).norm().item() / max(n64, 1e-30)
        tol = max(k_factor * errs_cpu[k], k_factor * med, floor)
        if not err <= tol:
            bad.append((k, f"gpu {err:.2e}", f"cpu32 {errs_cpu[k]:.2e}", f"tol {tol:.2e}"))
    return bad


