"""Drop-in for the ``VAE`` / ``VAEConfig`` classes of the reference's ``vae.py`` (37-67,
15-26): the frozen encoder / decoder either side of the latent DDPM (latent_diffusion.py:
205-206, 346).  ``encode``, ``reparameterize``, ``decode`` and ``forward`` run on libtdx;
the VAE's own training script (vae.py:69-213: MNIST download, BCE+KLD loop, wandb) is not
part of the path and is not reproduced.  Unlike the reference module, importing this one has
no side effects."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any

import torch
import torch.nn as nn

from . import _lib
from ._lib import lib, check

__all__ = ["VAE", "VAEConfig"]


@dataclass
class VAEConfig:
    """vae.py:15-26 (a pydantic model there; plain dataclass here, same fields/defaults)."""
    latent_dim: int = 20
    hidden_dim: int = 400
    input_dim: int = 784
    batch_size: int = 128
    epochs: int = 100
    learning_rate: float = 1e-3
    device: Any = None
    checkpoint_dir: str = "checkpoints"
    n_images_to_log: int = 8

    def __post_init__(self):
        if self.device is None:
            self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class VAE(nn.Module):
    """MLP VAE 784 -> 400 -> 20 -> 400 -> 784 (vae.py:37-67); same parameter names, shapes
    and default initialisation order (fc1, fc21, fc22, fc3, fc4)."""

    def __init__(self, config: VAEConfig):
        super().__init__()
        self.config = config
        self.fc1 = nn.Linear(config.input_dim, config.hidden_dim)
        self.fc21 = nn.Linear(config.hidden_dim, config.latent_dim)
        self.fc22 = nn.Linear(config.hidden_dim, config.latent_dim)
        self.fc3 = nn.Linear(config.latent_dim, config.hidden_dim)
        self.fc4 = nn.Linear(config.hidden_dim, config.input_dim)

    def _ptrs(self, names):
        ts = []
        for n in names:
            lin = getattr(self, n)
            ts += [lin.weight, lin.bias]
        for t in ts:
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.TdxError("VAE parameters must be contiguous fp32 CUDA tensors (call .to('cuda'))")
        return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

    @staticmethod
    def _check(x):
        if not x.is_cuda:
            raise _lib.TdxError("tiny_diffusion_amd runs on MI355X only: got a CPU tensor and there is no "
                                "CPU fallback (the CPU restatement lives in oracle/ and is test-only)")
        return x.contiguous().float()

    def _ws(self, B, dev):
        return torch.empty(lib.tdx_vae_workspace_floats(B, self.config.hidden_dim), dtype=torch.float32, device=dev)

    @torch.no_grad()
    def encode(self, x):
        """vae.py:51-53: x (B,784) -> (mu, logvar).  Inference only (the latent DDPM uses the
        VAE frozen, under no_grad: latent_diffusion.py:204-206)."""
        x = self._check(x)
        c = self.config
        if x.dim() != 2 or x.shape[1] != c.input_dim:
            raise ValueError(f"x must be (B,{c.input_dim})")
        B = x.shape[0]
        mu = torch.empty(B, c.latent_dim, dtype=torch.float32, device=x.device)
        logvar = torch.empty_like(mu)
        ws = self._ws(B, x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        check(lib.tdx_vae_encode(x.data_ptr(), self._ptrs(("fc1", "fc21", "fc22")), mu.data_ptr(), logvar.data_ptr(),
                                 ws.data_ptr(), B, c.input_dim, c.hidden_dim, c.latent_dim, st), "tdx_vae_encode")
        return mu, logvar

    @torch.no_grad()
    def reparameterize(self, mu, logvar, eps=None):
        """vae.py:55-58; ``eps=None`` draws ``torch.randn_like(std)`` like the reference."""
        mu, logvar = self._check(mu), self._check(logvar)
        eps = torch.randn_like(mu) if eps is None else self._check(eps)
        z = torch.empty_like(mu)
        st = torch.cuda.current_stream(mu.device).cuda_stream
        check(lib.tdx_vae_reparameterize(mu.data_ptr(), logvar.data_ptr(), eps.data_ptr(), z.data_ptr(), mu.numel(), st),
              "tdx_vae_reparameterize")
        return z

    @torch.no_grad()
    def decode(self, z):
        """vae.py:60-62: z (B,20) -> sigmoid output (B,784)."""
        z = self._check(z)
        c = self.config
        if z.dim() != 2 or z.shape[1] != c.latent_dim:
            raise ValueError(f"z must be (B,{c.latent_dim})")
        B = z.shape[0]
        out = torch.empty(B, c.input_dim, dtype=torch.float32, device=z.device)
        ws = self._ws(B, z.device)
        st = torch.cuda.current_stream(z.device).cuda_stream
        check(lib.tdx_vae_decode(z.data_ptr(), self._ptrs(("fc3", "fc4")), out.data_ptr(), ws.data_ptr(), B,
                                 c.input_dim, c.hidden_dim, c.latent_dim, st), "tdx_vae_decode")
        return out

    def forward(self, x):
        """vae.py:64-67."""
        mu, logvar = self.encode(x.reshape(-1, self.config.input_dim))
        z = self.reparameterize(mu, logvar)
        return self.decode(z), mu, logvar
