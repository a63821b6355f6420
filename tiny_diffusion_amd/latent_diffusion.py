"""Drop-in for the hot path of the reference's ``latent_diffusion.py`` (class-conditional DDPM
on the 20-d latents of the MLP VAE): ``NoiseModel(time_dim, num_classes, latent_dim)`` with
``forward(x, t, y)``, ``ForwardProcess`` (2-D ``q_sample``) and
``sample(vae, noise_model, diffusion, device, n_samples, y)`` (latent_diffusion.py:16-128,
131-154, 308-347)."""
from __future__ import annotations

import torch

from .schedule import ForwardProcess, sample_loop
from .unet import ARCH_LATENT, NoiseModelBase
from .vae import VAE, VAEConfig

__all__ = ["NoiseModel", "ForwardProcess", "sample", "VAE", "VAEConfig"]


class NoiseModel(NoiseModelBase):
    """eps_theta(z_t, t, y): 15 Linear, 13 BatchNorm1d, time/class signal added on the decoder
    path (latent_diffusion.py:16-128)."""

    def __init__(self, time_dim: int = ARCH_LATENT.time_dim, num_classes: int = 10, latent_dim: int = 20):
        if latent_dim != ARCH_LATENT.in_shape[0]:
            raise ValueError(f"libtdx is built for latent_dim={ARCH_LATENT.in_shape[0]} (reference default)")
        super().__init__(time_dim=time_dim, num_classes=num_classes, arch=ARCH_LATENT)
        self.latent_dim = latent_dim

    def forward(self, x, t, y):
        return self._forward_impl(x, t, y)


@torch.no_grad()
def sample(vae: VAE, noise_model: NoiseModel, diffusion: ForwardProcess, device, n_samples=16, y=None, **kw):
    """latent_diffusion.py:308-347: reverse loop over z (n,20), then ``vae.decode(z)`` viewed
    as (n,1,28,28); same argument errors."""
    if y is None:
        raise ValueError("Class labels 'y' must be provided for conditional generation.")
    if y.shape[0] != n_samples:
        raise ValueError("y must have shape (n_samples,)")
    vae.eval()
    z = sample_loop(noise_model, diffusion, device, n_samples, y, **kw)
    return vae.decode(z).view(-1, 1, 28, 28)
