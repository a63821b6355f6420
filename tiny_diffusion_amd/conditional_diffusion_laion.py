"""Drop-in for the hot path of the reference's ``conditional_diffusion_laion.py``
(text-conditioned latent DDPM): the same ``get_timestep_embedding``,
``NoiseModel(time_dim=768)`` with ``forward(x, t, text_embeds)``, ``ForwardProcess`` and
``sample(noise_model, diffusion, device, text_embeds, vae, scaling_factor)`` contracts
(conditional_diffusion_laion.py:222-232, 234-332, 335-358, 561-600).

Only the noise predictor and the reverse loop run on libtdx; the pretrained VAE and CLIP
text encoder the reference pulls from ``diffusers`` / ``transformers`` are external models
and stay whatever object the caller passes in (out of scope)."""
from __future__ import annotations

import torch

from . import _lib
from .schedule import ForwardProcess as _ForwardProcess, sample_loop
from .unet import ARCH_LAION, NoiseModelBase

__all__ = ["NoiseModel", "ForwardProcess", "get_timestep_embedding", "sample", "postprocess_images"]

TIME_DIM = ARCH_LAION.time_dim


def get_timestep_embedding(timesteps, embedding_dim):
    """conditional_diffusion_laion.py:222-232 on the device: ``tdx_timestep_embedding`` (the kernel
    the model's own time path runs).  (N,) timesteps -> (N, embedding_dim) fp32.  Integer timesteps go
    through the int64 kernel; floating-point ones (the reference takes ``timesteps[:, None].float()``, so a
    fractional t keeps its fraction) through ``tdx_timestep_embedding_f32``.  A CPU tensor raises: this
    package has no host path."""
    if not timesteps.is_cuda:
        raise _lib.TdxError("get_timestep_embedding runs on the GPU only (no CPU fallback)")
    if timesteps.is_floating_point():
        t, fn, name = timesteps.contiguous().to(torch.float32), _lib.lib.tdx_timestep_embedding_f32, "tdx_timestep_embedding_f32"
    else:
        t, fn, name = timesteps.contiguous().to(torch.int64), _lib.lib.tdx_timestep_embedding, "tdx_timestep_embedding"
    out = torch.empty((t.shape[0], int(embedding_dim)), dtype=torch.float32, device=t.device)
    if t.shape[0]:
        _lib.check(fn(t.data_ptr(), out.data_ptr(), t.shape[0], int(embedding_dim),
                      torch.cuda.current_stream(t.device).cuda_stream), name)
    return out


class NoiseModel(NoiseModelBase):
    """eps_theta(x_t, t, text_embeds) on (4,32,32) latents: sinusoidal embedding -> time_mlp,
    + text_embeds, UNet 32/64/128/256 (conditional_diffusion_laion.py:234-332)."""

    def __init__(self, time_dim: int = TIME_DIM):
        super().__init__(time_dim=time_dim, num_classes=0, arch=ARCH_LAION)

    def forward(self, x, t, text_embeds):
        return self._forward_impl(x, t, text_embeds)


class ForwardProcess(_ForwardProcess):
    """conditional_diffusion_laion.py:335-358 (q_sample(device, x_0, t) draws its own noise)."""


def postprocess_images(decoded):
    """(decoded / 2 + 0.5).clamp(0, 1) with NaN/Inf replaced by zeros, fp32
    (conditional_diffusion_laion.py:590-599)."""
    images = (decoded / 2 + 0.5).clamp(0, 1)
    images = torch.where(torch.logical_or(torch.isnan(images), torch.isinf(images)),
                         torch.zeros_like(images), images)
    return images.to(torch.float32)


@torch.no_grad()
def sample(noise_model: NoiseModel, diffusion: ForwardProcess, device, text_embeds=None, vae=None,
           scaling_factor=1.0, **kw):
    """conditional_diffusion_laion.py:561-600: the reverse loop over latents, then
    ``vae.decode(x / scaling_factor).sample`` and the image post-processing.  With
    ``vae=None`` the latents are returned (the decoder is an external pretrained model)."""
    if text_embeds is None:
        raise ValueError("Text embeddings must be provided for conditional generation.")
    n_samples = text_embeds.shape[0]
    x = sample_loop(noise_model, diffusion, device, n_samples, text_embeds, **kw)
    if vae is None:
        return x
    decoded = vae.decode(x / scaling_factor).sample
    return postprocess_images(decoded)
