"""Drop-in for the hot path of the reference's ``diffusion_transformer.py``: ``TransformerBlock``,
``NoiseModel(time_dim, num_classes, latent_dim, num_heads, num_layers, dropout)`` with
``forward(x, t, y)``, ``ForwardProcess`` and ``sample(vae, ...)`` (diffusion_transformer.py:16-132,
284-323).

The reference feeds its blocks a LENGTH-1 sequence (``x.unsqueeze(0)``, line 99, with
``nn.MultiheadAttention`` in its default sequence-first layout), so the softmax runs over a single
key and ``attention(x, x, x) == out_proj(v_proj(x))``; in train mode the attention-weight dropout
zeroes or rescales that single weight per (row, head).  That is what runs here (Q and K projections
receive exactly-zero gradients, as in the reference).  The modules below own the parameters under
the reference's names (``nn.MultiheadAttention.in_proj_weight`` etc.) and give the reference's
default initialisation; their torch ``forward`` is never called."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib, ops
from .schedule import ForwardProcess, sample_loop
from .vae import VAE, VAEConfig

__all__ = ["TransformerBlock", "NoiseModel", "ForwardProcess", "sample", "VAE", "VAEConfig"]


class TransformerBlock(nn.Module):
    """diffusion_transformer.py:16-35 (post-norm block: x = LN(x + drop(attn)); x = LN(x + drop(ff)))."""

    def __init__(self, dim, num_heads, ff_dim, dropout=0.1):
        super().__init__()
        self.attention = nn.MultiheadAttention(dim, num_heads, dropout=dropout)
        self.norm1 = nn.LayerNorm(dim)
        self.ff = nn.Sequential(nn.Linear(dim, ff_dim), nn.GELU(), nn.Linear(ff_dim, dim), nn.Dropout(dropout))
        self.norm2 = nn.LayerNorm(dim)
        self.dropout = nn.Dropout(dropout)
        self.dim, self.num_heads, self.p = dim, num_heads, float(dropout)

    def forward(self, x, rng=None):
        """x: (B, dim) - the reference's (1, B, dim) with the length-1 sequence axis dropped."""
        D = self.dim
        drop = self.training and self.p > 0.0
        att = self.attention
        v = ops.linear(x, att.in_proj_weight[2 * D:], att.in_proj_bias[2 * D:])
        if drop:  # dropout of the (single) attention weight of every head
            v = ops.dropout(v, self.p, D // self.num_heads, *rng())
        a = ops.linear(v, att.out_proj.weight, att.out_proj.bias)
        if drop:
            a = ops.dropout(a, self.p, 1, *rng())
        x = ops.layer_norm(ops.add(x, a), self.norm1.weight, self.norm1.bias, self.norm1.eps)
        f = ops.gelu(ops.linear(x, self.ff[0].weight, self.ff[0].bias))
        f = ops.linear(f, self.ff[2].weight, self.ff[2].bias)
        if drop:  # ff's own nn.Dropout, then self.dropout (lines 25, 33)
            f = ops.dropout(f, self.p, 1, *rng())
            f = ops.dropout(f, self.p, 1, *rng())
        return ops.layer_norm(ops.add(x, f), self.norm2.weight, self.norm2.bias, self.norm2.eps)


class _Shape:
    in_shape = (20,)


class NoiseModel(nn.Module):
    """diffusion_transformer.py:38-107."""

    def __init__(self, time_dim=256, num_classes=10, latent_dim=20, num_heads=4, num_layers=4, dropout=0.05):
        super().__init__()
        if time_dim % 64 or time_dim > 1024:
            raise ValueError("libtdx LayerNorm needs time_dim % 64 == 0 and <= 1024")
        self.time_dim = time_dim
        self.latent_dim = latent_dim
        self.time_embedding = nn.Sequential(nn.Linear(1, time_dim), nn.SiLU(), nn.Linear(time_dim, time_dim))
        self.class_embedding = nn.Embedding(num_classes, time_dim)
        self.input_proj = nn.Linear(latent_dim, time_dim)
        self.pos_encoding = nn.Parameter(torch.randn(1, 1, time_dim))
        self.transformer_blocks = nn.ModuleList(
            [TransformerBlock(time_dim, num_heads, time_dim * 4, dropout) for _ in range(num_layers)])
        self.final_layer = nn.Sequential(nn.LayerNorm(time_dim), nn.Linear(time_dim, latent_dim))
        self._arch = _Shape()
        self._arch.in_shape = (latent_dim,)
        # dropout masks: Philox stream (seed drawn from torch's generator once per forward, so
        # torch.manual_seed reproduces a run), one offset per dropout site
        self._calls = 0

    def _rng_factory(self):
        seed = int(torch.randint(0, 2**62, (1,)).item()) if self.training else 0
        state = {"off": 0}

        def rng():
            state["off"] += 1
            return seed, state["off"]

        return rng

    def forward(self, x, t, y):
        if not x.is_cuda:
            raise _lib.TdxError("tiny_diffusion_amd runs on MI355X only: got a CPU tensor and there is no "
                                "CPU fallback (the CPU restatement lives in oracle/ and is test-only)")
        if x.dim() != 2 or x.shape[1] != self.latent_dim:
            raise ValueError(f"x must be (B,{self.latent_dim})")
        if t.shape != (x.shape[0],) or y.shape != (x.shape[0],):
            raise ValueError("t and y must have shape (B,)")
        rng = self._rng_factory()
        te = self.time_embedding
        tn = (t / 1000).unsqueeze(-1).float()                       # line 86
        h = ops.silu(ops.linear(tn, te[0].weight, te[0].bias))
        emb = ops.add(ops.linear(h, te[2].weight, te[2].bias), ops.embedding(self.class_embedding.weight, y))
        h = ops.add(ops.linear(x, self.input_proj.weight, self.input_proj.bias), emb)   # lines 92-95
        h = ops.add(h, self.pos_encoding.view(-1))                   # line 98 (broadcast over the batch)
        for blk in self.transformer_blocks:
            h = blk(h, rng)
        fl = self.final_layer
        return ops.linear(ops.layer_norm(h, fl[0].weight, fl[0].bias, fl[0].eps), fl[1].weight, fl[1].bias)

    # sample_loop's entry point (inference forward, no autograd graph)
    @torch.no_grad()
    def _run_forward(self, x, t, y, mode=2):
        return self.forward(x, t, y), None, mode


@torch.no_grad()
def sample(vae: VAE, noise_model: NoiseModel, diffusion: ForwardProcess, device, n_samples=16, y=None, **kw):
    """diffusion_transformer.py:284-323 (identical to latent_diffusion.sample)."""
    if y is None:
        raise ValueError("Class labels 'y' must be provided for conditional generation.")
    if y.shape[0] != n_samples:
        raise ValueError("y must have shape (n_samples,)")
    vae.eval()
    z = sample_loop(noise_model, diffusion, device, n_samples, y, **kw)
    return vae.decode(z).view(-1, 1, 28, 28)
