"""CPU restatement of the tiny-diffusion DDPM hot path (test infrastructure).

Plain torch fp32 on CPU, functional, written from the reference's behaviour —
NOT imported by the product.  Every function cites the reference lines it
follows (paths relative to /root/reference).  It consumes a reference-format
``state_dict`` (OIHW conv weights, the key list of oracle/weights.py).

BatchNorm, ceil-mode max-pool and align-corners bilinear resize are written
out explicitly (index maths and all) so that they are a specification of the
HIP kernels rather than a second call into the same ATen code; the 3x3
convolution and the linear layers use ``F.conv2d`` / ``F.linear``.

Parity: pinned against golden vectors generated from the reference itself
(tools/make_golden.py -> tests/golden/, checked by tests/test_oracle_golden.py).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm2d default, diffusion.py:34
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------
# schedule / forward process — diffusion.py:165-190
# --------------------------------------------------------------------------
class Schedule:
    """diffusion.py:166-175: linear beta schedule, fp32 CPU tensors."""

    def __init__(self, num_timesteps=1000, beta_start=1e-4, beta_end=0.02):
        self.num_timesteps = num_timesteps
        self.betas = torch.linspace(beta_start, beta_end, num_timesteps)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)

    def p_sample_coeffs(self):
        """Per-t (c1, c2, sigma) with the reference's exact fp32 expressions
        (diffusion.py:272-274): x' = c1 * (x - c2 * eps) + sigma * z."""
        c1 = 1 / torch.sqrt(self.alphas)
        c2 = (1 - self.alphas) / torch.sqrt(1 - self.alphas_cumprod)
        sigma = torch.sqrt(self.betas)
        return c1, c2, sigma


def q_sample(sched: Schedule, x0, t, noise):
    """diffusion.py:177-190 with the noise supplied by the caller."""
    a = torch.sqrt(sched.alphas_cumprod[t]).view(-1, 1, 1, 1)
    b = torch.sqrt(1.0 - sched.alphas_cumprod[t]).view(-1, 1, 1, 1)
    return a * x0 + b * noise


def p_sample_step(sched: Schedule, x, eps, t: int, z):
    """One reverse step, diffusion.py:263-274 (z must be zeros at t == 0)."""
    alpha = sched.alphas[t]
    alpha_cumprod = sched.alphas_cumprod[t]
    beta = sched.betas[t]
    return (1 / torch.sqrt(alpha)) * (
        x - ((1 - alpha) / torch.sqrt(1 - alpha_cumprod)) * eps
    ) + torch.sqrt(beta) * z


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def batchnorm2d(x, weight, bias, running_mean, running_var, training, nbt=None):
    """nn.BatchNorm2d semantics (diffusion.py:34 etc.), NCHW.

    train: normalise with the biased batch variance over (N,H,W); update
    running stats in place with momentum 0.1 and the UNBIASED variance.
    eval: normalise with the running stats.
    """
    if training:
        n = x.shape[0] * x.shape[2] * x.shape[3]
        mean = x.mean(dim=(0, 2, 3))
        var = (x - mean.view(1, -1, 1, 1)).pow(2).mean(dim=(0, 2, 3))
        if running_mean is not None:
            with torch.no_grad():
                running_mean.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
                running_var.mul_(1 - BN_MOMENTUM).add_(
                    BN_MOMENTUM * var * (n / max(n - 1, 1))
                )
                if nbt is not None:
                    nbt.add_(1)
    else:
        mean, var = running_mean, running_var
    rstd = torch.rsqrt(var + BN_EPS)
    return (x - mean.view(1, -1, 1, 1)) * (rstd * weight).view(1, -1, 1, 1) + bias.view(
        1, -1, 1, 1
    )


def pool_windows(x):
    """(n,c,h,w) -> (n,c,ho,wo,4) windows in ATen scan order, -inf padded (ceil mode)."""
    n, c, h, w = x.shape
    ho, wo = (h + 1) // 2, (w + 1) // 2
    xp = F.pad(x, (0, 2 * wo - w, 0, 2 * ho - h), value=float("-inf"))
    return xp.view(n, c, ho, 2, wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, ho, wo, 4)


def maxpool2_ceil(x, idx=None):
    """nn.MaxPool2d(2, ceil_mode=True) (diffusion.py:101): 2x2/stride 2; the
    last window of an odd extent holds a single valid row/column (7 -> 4).

    Backward semantics matter: ATen routes the whole gradient to the FIRST maximum
    of a window in scan order (`val > maxval`), it does not split it among ties.
    argmax returns the first maximal index; gather's backward routes to exactly
    that element.

    ``idx`` (n,c,ho,wo,1) overrides the arg-max: tests use it to evaluate the oracle
    with the SAME routing decisions as the implementation under test - two window
    entries that agree to ~1e-6 are a coin flip in fp32, and either choice is a
    valid sub-gradient."""
    win = pool_windows(x)
    if idx is None:
        idx = win.detach().argmax(dim=-1, keepdim=True)
    return torch.gather(win, -1, idx).squeeze(-1)


def _ac_axis(n_in: int, n_out: int, dtype=torch.float32):
    """align_corners=True source coordinates for one axis, fp32 like ATen:
    scale = (in-1)/(out-1); src = scale*dst; i0 = floor(src); lam = src-i0."""
    if n_out > 1:
        scale = torch.tensor((n_in - 1) / (n_out - 1), dtype=torch.float32)
    else:
        scale = torch.tensor(0.0, dtype=torch.float32)
    dst = torch.arange(n_out, dtype=torch.float32)
    src = scale * dst
    i0 = src.floor().to(torch.int64).clamp_(max=n_in - 1)
    i1 = (i0 + 1).clamp_(max=n_in - 1)
    lam1 = src - i0.to(torch.float32)
    lam0 = 1.0 - lam1
    return i0, i1, lam0.to(dtype), lam1.to(dtype)


def bilinear_ac(x, size):
    """F.interpolate(mode='bilinear', align_corners=True) / nn.Upsample(2x)
    (diffusion.py:102, 135-159), NCHW."""
    ho, wo = size
    h0, h1, hl0, hl1 = _ac_axis(x.shape[2], ho, x.dtype)
    w0, w1, wl0, wl1 = _ac_axis(x.shape[3], wo, x.dtype)
    top = x[:, :, h0, :]
    bot = x[:, :, h1, :]
    wl0 = wl0.view(1, 1, 1, -1)
    wl1 = wl1.view(1, 1, 1, -1)
    t = top[:, :, :, w0] * wl0 + top[:, :, :, w1] * wl1
    b = bot[:, :, :, w0] * wl0 + bot[:, :, :, w1] * wl1
    return t * hl0.view(1, 1, -1, 1) + b * hl1.view(1, 1, -1, 1)


def conv_bn_relu(x, p, prefix_conv, prefix_bn, training, buffers, relu_mask=None, taps=None):
    """conv3x3 -> BatchNorm2d -> ReLU.  ``relu_mask`` (bool, NCHW) overrides the sign test of the
    ReLU: tests use it to evaluate the oracle with the SAME active set as the implementation under
    test - an activation within fp32 rounding of 0 is a coin flip (its value is ~0 either way, but
    its gradient path is on or off), and either choice is a valid sub-gradient of max(0, .).
    ``taps`` (dict) receives the normalised pre-activation under the key ``prefix_bn``."""
    y = F.conv2d(x, p[f"{prefix_conv}.weight"], p[f"{prefix_conv}.bias"], padding=1)
    y = batchnorm2d(
        y,
        p[f"{prefix_bn}.weight"],
        p[f"{prefix_bn}.bias"],
        buffers[f"{prefix_bn}.running_mean"],
        buffers[f"{prefix_bn}.running_var"],
        training,
        buffers.get(f"{prefix_bn}.num_batches_tracked"),
    )
    if taps is not None:
        taps["prebn_act:" + prefix_bn] = y
    if relu_mask is not None:
        return y * relu_mask.to(y.dtype)
    return F.relu(y)


def stage(x, p, name, training, buffers, relu_masks=None, taps=None):
    """[conv3x3 -> BN -> ReLU] x2, diffusion.py:32-39 and siblings."""
    rm = relu_masks or {}
    x = conv_bn_relu(x, p, f"{name}.0", f"{name}.1", training, buffers, rm.get(f"{name}.1"), taps)
    return conv_bn_relu(x, p, f"{name}.3", f"{name}.4", training, buffers, rm.get(f"{name}.4"), taps)


def time_embedding(p, t, y=None):
    """diffusion.py:111-113 (+ conditional_diffusion.py:121-125): raw integer t
    cast to float, Linear(1,256) -> SiLU -> Linear(256,256), plus E[y]."""
    # .float() in the reference; the weights' dtype here so that the same code also
    # runs as the fp64 ground truth used to calibrate gradient tolerances
    tf = t.unsqueeze(-1).to(p["time_embedding.0.weight"].dtype)
    h = F.linear(tf, p["time_embedding.0.weight"], p["time_embedding.0.bias"])
    h = h * torch.sigmoid(h)
    emb = F.linear(h, p["time_embedding.2.weight"], p["time_embedding.2.bias"])
    if y is not None:
        emb = emb + p["class_embedding.weight"][y]
    return emb


def split_state(sd):
    """state_dict -> (params, buffers); buffers are cloned so that train-mode
    forward can update them in place without touching the caller's dict."""
    params, buffers = OrderedDict(), OrderedDict()
    for k, v in sd.items():
        if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            buffers[k] = v.clone()
        else:
            params[k] = v
    return params, buffers


def unet_forward(p, buffers, x, t, y=None, training=False, taps=None, pool_idx=None, relu_masks=None):
    """NoiseModel.forward, diffusion.py:109-162 / conditional_diffusion.py:115-172.

    ``p``: parameter dict; ``buffers``: BN buffers (updated in place when
    training).  ``taps`` (optional dict) receives every named intermediate.
    ``pool_idx`` / ``relu_masks``: sub-gradient choices of the implementation under test
    (maxpool2_ceil, conv_bn_relu); keys "e1".."e3" / BatchNorm module names ("enc1.1", ...).
    """
    rm = relu_masks

    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    emb = tap("emb", time_embedding(p, t, y))
    x0 = tap("x0", F.conv2d(x, p["initial_conv.weight"], p["initial_conv.bias"], padding=1))
    e1 = tap("e1", stage(x0, p, "enc1", training, buffers, rm, taps))
    pool_idx = pool_idx or {}
    e1p = tap("e1p", maxpool2_ceil(e1, pool_idx.get("e1")))
    e2 = tap("e2", stage(e1p, p, "enc2", training, buffers, rm, taps))
    e2p = tap("e2p", maxpool2_ceil(e2, pool_idx.get("e2")))
    e3 = tap("e3", stage(e2p, p, "enc3", training, buffers, rm, taps))
    e3p = tap("e3p", maxpool2_ceil(e3, pool_idx.get("e3")))
    b = tap("b", conv_bn_relu(e3p, p, "bottleneck.0", "bottleneck.1", training, buffers,
                              (rm or {}).get("bottleneck.1"), taps))

    # 1x1 conv on a (B,256,1,1) map == linear, diffusion.py:130-132
    def proj(k):
        w = p[f"time_proj{k}.weight"]
        return F.linear(emb, w.view(w.shape[0], -1), p[f"time_proj{k}.bias"]).view(
            emb.shape[0], -1, 1, 1
        )

    t1, t2, t3 = tap("t1", proj(1)), tap("t2", proj(2)), tap("t3", proj(3))

    up_b = tap("up_b", bilinear_ac(b, (8, 8)))
    e3a = tap("e3a", bilinear_ac(e3 + t3, (8, 8)))
    d3 = tap("d3", stage(torch.cat([up_b, e3a], dim=1), p, "dec3", training, buffers, rm, taps))
    up_d3 = tap("up_d3", bilinear_ac(d3, (16, 16)))
    e2a = tap("e2a", bilinear_ac(e2 + t2, (16, 16)))
    d2 = tap("d2", stage(torch.cat([up_d3, e2a], dim=1), p, "dec2", training, buffers, rm, taps))
    up_d2 = tap("up_d2", bilinear_ac(d2, (32, 32)))
    e1a = tap("e1a", bilinear_ac(e1 + t1, (32, 32)))
    d1 = tap("d1", stage(torch.cat([up_d2, e1a], dim=1), p, "dec1", training, buffers, rm, taps))
    d1a = tap("d1a", bilinear_ac(d1, (28, 28)))
    out = F.conv2d(d1a, p["final_conv.weight"], p["final_conv.bias"], padding=1)
    return tap("out", out)


def train_step_grads(sd, x_t, t, noise, y=None, training=True, dtype=torch.float32, pool_idx=None,
                     taps=None, relu_masks=None):
    """Forward + MSE + backward (diffusion.py:228-235) on the oracle.

    Returns (loss, eps_hat, grads dict, updated buffers).  ``dtype=torch.float64``
    evaluates the same graph in double precision (ground truth for tolerances:
    train-mode BatchNorm over a near-dead channel amplifies fp32 rounding noise
    by up to 1/sqrt(eps) ~ 300x, in the reference just the same)."""
    params, buffers = split_state(sd)
    if dtype != torch.float32:
        params = OrderedDict((k, v.to(dtype)) for k, v in params.items())
        buffers = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v) for k, v in buffers.items())
        x_t, noise = x_t.to(dtype), noise.to(dtype)
    leaves = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in params.items())
    eps_hat = unet_forward(leaves, buffers, x_t, t, y, training=training, taps=taps, pool_idx=pool_idx,
                           relu_masks=relu_masks)
    loss = F.mse_loss(eps_hat, noise)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), eps_hat.detach(), OrderedDict(zip(leaves.keys(), grads)), buffers


def adam_step(params, grads, state, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam defaults (diffusion.py:211, 236), one step, in place."""
    b1, b2 = betas
    state["step"] = state.get("step", 0) + 1
    k = state["step"]
    for name, p in params.items():
        g = grads[name]
        m = state.setdefault(("m", name), torch.zeros_like(p))
        v = state.setdefault(("v", name), torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1**k
        bc2 = 1 - b2**k
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


@torch.no_grad()
def sample_chain(sd, sched: Schedule, x_T, noises, y=None, record=None):
    """sample(), diffusion.py:254-276 / conditional_diffusion.py:354-386, with
    recorded noise: ``noises[t]`` is the z used at step t (t = T-1 .. 1);
    step 0 uses zeros.  ``record`` (dict) receives x after the listed steps."""
    params, buffers = split_state(sd)
    x = x_T
    n = x.shape[0]
    for t in reversed(range(sched.num_timesteps)):
        tt = torch.full((n,), t, dtype=torch.long)
        eps = unet_forward(params, buffers, x, tt, y, training=False)
        z = noises[t] if t > 0 else torch.zeros_like(x)
        x = p_sample_step(sched, x, eps, t, z)
        if record is not None and t in record:
            record[t] = x.clone()
    return x
