"""CPU restatement of the latent-space DDPM (test infrastructure):
latent_diffusion.py:16-154 (MLP noise model on 20-d latents, 2-D q_sample), 308-347
(sample + vae.decode) and the MLP VAE of vae.py:37-62.  Functional torch, fp32 (fp64 on
request), no nn.Module.  Parity: pinned by tests/golden/latent_B32.npz, produced by the
reference's own classes (tools/make_golden.py)."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import ref_cpu as R


def batchnorm1d(x, p, buffers, name, training):
    """nn.BatchNorm1d forward incl. the running-statistics update (momentum 0.1, unbiased
    variance, eps 1e-5) - the (N, C) case of ref_cpu.batchnorm2d."""
    w, b = p[f"{name}.weight"], p[f"{name}.bias"]
    rm, rv = buffers[f"{name}.running_mean"], buffers[f"{name}.running_var"]
    if training:
        n = x.shape[0]
        mean = x.mean(dim=0)
        var = x.var(dim=0, unbiased=False)
        with torch.no_grad():
            rm.mul_(0.9).add_(0.1 * mean.detach().to(rm.dtype))
            rv.mul_(0.9).add_(0.1 * (var.detach() * n / max(n - 1, 1)).to(rv.dtype))
            buffers[f"{name}.num_batches_tracked"] += 1
    else:
        mean, var = rm, rv
    return (x - mean) / torch.sqrt(var + 1e-5) * w + b


def lin_bn_relu(x, p, buffers, lin, bn, training):
    y = F.linear(x, p[f"{lin}.weight"], p[f"{lin}.bias"])
    return torch.relu(batchnorm1d(y, p, buffers, bn, training))


def stage(x, p, buffers, name, training):
    x = lin_bn_relu(x, p, buffers, f"{name}.0", f"{name}.1", training)
    return lin_bn_relu(x, p, buffers, f"{name}.3", f"{name}.4", training)


def noise_forward(p, buffers, z, t, y, training=False, taps=None):
    """NoiseModel.forward, latent_diffusion.py:107-128.  The time/class signal is added to
    the DECODER path (b + t1, d3 + t2, d2 + t3), the skips are concatenated after it."""

    def tap(name, v):
        if taps is not None:
            taps[name] = v
        return v

    tf = t.unsqueeze(-1).to(z.dtype)
    h = F.linear(tf, p["time_embedding.0.weight"], p["time_embedding.0.bias"])
    h = h * torch.sigmoid(h)
    emb = F.linear(h, p["time_embedding.2.weight"], p["time_embedding.2.bias"]) + p["class_embedding.weight"][y]
    tap("emb", emb)
    x0 = tap("x0", F.linear(z, p["initial_fc.weight"], p["initial_fc.bias"]))
    e1 = tap("e1", stage(x0, p, buffers, "enc1", training))
    e2 = tap("e2", stage(e1, p, buffers, "enc2", training))
    e3 = tap("e3", stage(e2, p, buffers, "enc3", training))
    b = tap("b", lin_bn_relu(e3, p, buffers, "bottleneck.0", "bottleneck.1", training))
    t1 = F.linear(emb, p["time_proj1.weight"], p["time_proj1.bias"])
    t2 = F.linear(emb, p["time_proj2.weight"], p["time_proj2.bias"])
    t3 = F.linear(emb, p["time_proj3.weight"], p["time_proj3.bias"])
    d3 = tap("d3", stage(torch.cat([b + t1, e3], dim=1), p, buffers, "dec3", training))
    d2 = tap("d2", stage(torch.cat([d3 + t2, e2], dim=1), p, buffers, "dec2", training))
    d1 = tap("d1", stage(torch.cat([d2 + t3, e1], dim=1), p, buffers, "dec1", training))
    return tap("out", F.linear(d1, p["final_fc.weight"], p["final_fc.bias"]))


def train_step_grads(sd, z_t, t, noise, y, training=True, dtype=torch.float32):
    """forward + MSE + backward (latent_diffusion.py:214-219)."""
    params, buffers = R.split_state(sd)
    if dtype != torch.float32:
        params = OrderedDict((k, v.to(dtype)) for k, v in params.items())
        buffers = OrderedDict((k, v.to(dtype) if v.is_floating_point() else v) for k, v in buffers.items())
        z_t, noise = z_t.to(dtype), noise.to(dtype)
    leaves = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in params.items())
    eps = noise_forward(leaves, buffers, z_t, t, y, training=training)
    loss = F.mse_loss(eps, noise)
    grads = torch.autograd.grad(loss, list(leaves.values()))
    return loss.detach(), eps.detach(), OrderedDict(zip(leaves.keys(), grads)), buffers


def q_sample(sched: R.Schedule, z_0, t, noise):
    """ForwardProcess.q_sample on (B, latent_dim) tensors, latent_diffusion.py:143-154."""
    a = torch.sqrt(sched.alphas_cumprod[t]).view(-1, 1)
    b = torch.sqrt(1.0 - sched.alphas_cumprod[t]).view(-1, 1)
    return a * z_0 + b * noise


# ----------------------------------------------------------------------- VAE
def vae_encode(vp, x):
    """VAE.encode, vae.py:51-53: x (B,784) -> mu, logvar (B,20)."""
    h1 = torch.relu(F.linear(x, vp["fc1.weight"], vp["fc1.bias"]))
    return F.linear(h1, vp["fc21.weight"], vp["fc21.bias"]), F.linear(h1, vp["fc22.weight"], vp["fc22.bias"])


def vae_reparameterize(mu, logvar, eps):
    """VAE.reparameterize, vae.py:55-58, with the noise passed in."""
    return mu + eps * torch.exp(0.5 * logvar)


def vae_decode(vp, z):
    """VAE.decode, vae.py:60-62: z (B,20) -> (B,784) in (0,1)."""
    h3 = torch.relu(F.linear(z, vp["fc3.weight"], vp["fc3.bias"]))
    return torch.sigmoid(F.linear(h3, vp["fc4.weight"], vp["fc4.bias"]))


@torch.no_grad()
def sample_chain(sd, vp, sched: R.Schedule, z_T, noises, y):
    """Reverse loop of sample(), latent_diffusion.py:324-344, recorded noise, then
    vae.decode(z).view(-1,1,28,28) (346)."""
    params, buffers = R.split_state(sd)
    z = z_T
    n = z.shape[0]
    for t in reversed(range(sched.num_timesteps)):
        tt = torch.full((n,), t, dtype=torch.long)
        eps = noise_forward(params, buffers, z, tt, y, training=False)
        zz = noises[t] if t > 0 else torch.zeros_like(z)
        z = R.p_sample_step(sched, z, eps, t, zz)
    return z, vae_decode(vp, z).view(-1, 1, 28, 28)
