"""CPU restatement of the LAION-shaped latent UNet (test infrastructure):
conditional_diffusion_laion.py:221-332 - sinusoidal timestep embedding, time MLP,
additive 768-d text conditioning, UNet 32/64/128/256 on (4,32,32) latents with floor-mode
pooling and exact 2x up-sampling (no resize-to-match).  Reuses the building blocks of
oracle/ref_cpu.py.  Parity: pinned by tests/golden/laion_*.npz (tools/make_golden.py)."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import ref_cpu as R


def timestep_embedding(t, dim: int):
    """get_timestep_embedding, conditional_diffusion_laion.py:222-232 (fp32; the dtype of
    the result follows `freq_dtype` only through the later cast in unet_forward)."""
    half = dim // 2
    freqs = torch.exp(-torch.log(torch.tensor(10000.0)) * torch.arange(half, dtype=torch.float32) / (half - 1))
    e = t[:, None].float() * freqs[None, :]
    e = torch.cat([torch.sin(e), torch.cos(e)], dim=-1)
    if dim % 2 == 1:
        e = torch.cat([e, torch.zeros_like(e[:, :1])], dim=-1)
    return e


def unet_forward(p, buffers, x, t, text_embeds, training=False, taps=None, pool_idx=None, relu_masks=None):
    """NoiseModel.forward, conditional_diffusion_laion.py:304-332."""

    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    pool_idx = pool_idx or {}
    dim = p["time_mlp.0.weight"].shape[0]
    sin = timestep_embedding(t, dim).to(p["time_mlp.0.weight"].dtype)
    h = F.linear(sin, p["time_mlp.0.weight"], p["time_mlp.0.bias"])
    h = h * torch.sigmoid(h)
    emb = tap("emb", F.linear(h, p["time_mlp.2.weight"], p["time_mlp.2.bias"]) + text_embeds)
    x0 = tap("x0", F.conv2d(x, p["initial_conv.weight"], p["initial_conv.bias"], padding=1))
    e1 = tap("e1", R.stage(x0, p, "enc1", training, buffers, relu_masks, taps))
    e1p = R.maxpool2_ceil(e1, pool_idx.get("e1"))  # even extents: floor == ceil mode
    e2 = tap("e2", R.stage(e1p, p, "enc2", training, buffers, relu_masks, taps))
    e2p = R.maxpool2_ceil(e2, pool_idx.get("e2"))
    e3 = tap("e3", R.stage(e2p, p, "enc3", training, buffers, relu_masks, taps))
    e3p = R.maxpool2_ceil(e3, pool_idx.get("e3"))
    b = tap("b", R.conv_bn_relu(e3p, p, "bottleneck.0", "bottleneck.1", training, buffers,
                                (relu_masks or {}).get("bottleneck.1"), taps))

    def proj(k):
        w = p[f"time_proj{k}.weight"]
        return F.linear(emb, w.view(w.shape[0], -1), p[f"time_proj{k}.bias"]).view(emb.shape[0], -1, 1, 1)

    t1, t2, t3 = proj(1), proj(2), proj(3)
    up = lambda v: R.bilinear_ac(v, (2 * v.shape[2], 2 * v.shape[3]))  # nn.Upsample(2, bilinear, align_corners)
    d3 = tap("d3", R.stage(torch.cat([up(b), e3 + t3], dim=1), p, "dec3", training, buffers, relu_masks, taps))
    d2 = tap("d2", R.stage(torch.cat([up(d3), e2 + t2], dim=1), p, "dec2", training, buffers, relu_masks, taps))
    d1 = tap("d1", R.stage(torch.cat([up(d2), e1 + t1], dim=1), p, "dec1", training, buffers, relu_masks, taps))
    return tap("out", F.conv2d(d1, p["final_conv.weight"], p["final_conv.bias"], padding=1))


def train_step_grads(sd, x_t, t, noise, text_embeds, training=True, dtype=torch.float32, pool_idx=None,
                     relu_masks=None):
    """forward + MSE + backward (conditional_diffusion_laion.py:466-469)."""
    params, buffers = R.split_state(sd)
    if dtype != torch.float32:
        params = OrderedDict((k, v.to(dtype)) for k, v in params.items())
        buffers = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v) for k, v in buffers.items())
        x_t, noise, text_embeds = x_t.to(dtype), noise.to(dtype), text_embeds.to(dtype)
    leaves = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in params.items())
    eps = unet_forward(leaves, buffers, x_t, t, text_embeds, training=training, pool_idx=pool_idx,
                       relu_masks=relu_masks)
    loss = F.mse_loss(eps, noise)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), eps.detach(), OrderedDict(zip(leaves.keys(), grads)), buffers


@torch.no_grad()
def sample_chain(sd, sched: R.Schedule, x_T, noises, text_embeds):
    """Reverse loop of sample(), conditional_diffusion_laion.py:574-587, recorded noise;
    the VAE decode that follows it in the reference is outside the path."""
    params, buffers = R.split_state(sd)
    x = x_T
    n = x.shape[0]
    for t in reversed(range(sched.num_timesteps)):
        tt = torch.full((n,), t, dtype=torch.long)
        eps = unet_forward(params, buffers, x, tt, text_embeds, training=False)
        z = noises[t] if t > 0 else torch.zeros_like(x)
        x = R.p_sample_step(sched, x, eps, t, z)
    return x
