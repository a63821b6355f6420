"""CPU restatement of the "transformer" noise model (test infrastructure):
diffusion_transformer.py:16-107 with dropout inactive (eval mode, or dropout=0).  The reference
runs nn.MultiheadAttention on a length-1 sequence (line 99), so the softmax is over one key and
attention(x, x, x) == out_proj(v_proj(x)); this restatement keeps the general formula (scores,
softmax, weighted sum) so that the identity is checked rather than assumed.  Parity: pinned by
tests/golden/transformer_B16.npz (tools/make_golden.py)."""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F


def mha_seq1(x, p, prefix, num_heads):
    """nn.MultiheadAttention(dim, heads)(x, x, x) for x of shape (1, B, D), sequence-first."""
    L, B, D = x.shape
    w, b = p[f"{prefix}.in_proj_weight"], p[f"{prefix}.in_proj_bias"]
    q, k, v = F.linear(x, w, b).chunk(3, dim=-1)
    hd = D // num_heads
    # (L, B, H, hd) -> (B*H, L, hd)
    q, k, v = (u.reshape(L, B * num_heads, hd).transpose(0, 1) for u in (q, k, v))
    att = torch.softmax(q @ k.transpose(1, 2) / hd ** 0.5, dim=-1)
    o = (att @ v).transpose(0, 1).reshape(L, B, D)
    return F.linear(o, p[f"{prefix}.out_proj.weight"], p[f"{prefix}.out_proj.bias"])


def noise_forward(p, z, t, y, num_heads=4, taps=None):
    """NoiseModel.forward, diffusion_transformer.py:82-107 (no dropout)."""
    tn = (t / 1000).unsqueeze(-1).to(z.dtype)
    h = F.linear(tn, p["time_embedding.0.weight"], p["time_embedding.0.bias"])
    h = h * torch.sigmoid(h)
    emb = F.linear(h, p["time_embedding.2.weight"], p["time_embedding.2.bias"]) + p["class_embedding.weight"][y]
    x = F.linear(z, p["input_proj.weight"], p["input_proj.bias"]) + emb
    x = x.unsqueeze(0) + p["pos_encoding"]
    D = x.shape[-1]
    i = 0
    while f"transformer_blocks.{i}.norm1.weight" in p:
        pre = f"transformer_blocks.{i}"
        a = mha_seq1(x, p, f"{pre}.attention", num_heads)
        x = F.layer_norm(x + a, (D,), p[f"{pre}.norm1.weight"], p[f"{pre}.norm1.bias"])
        f = F.linear(F.gelu(F.linear(x, p[f"{pre}.ff.0.weight"], p[f"{pre}.ff.0.bias"])),
                     p[f"{pre}.ff.2.weight"], p[f"{pre}.ff.2.bias"])
        x = F.layer_norm(x + f, (D,), p[f"{pre}.norm2.weight"], p[f"{pre}.norm2.bias"])
        if taps is not None:
            taps[f"block{i}"] = x.squeeze(0)
        i += 1
    x = x.squeeze(0)
    x = F.layer_norm(x, (D,), p["final_layer.0.weight"], p["final_layer.0.bias"])
    return F.linear(x, p["final_layer.1.weight"], p["final_layer.1.bias"])


def train_step_grads(sd, z_t, t, noise, y, dtype=torch.float32):
    params = OrderedDict((k, v.to(dtype).clone().requires_grad_(True)) for k, v in sd.items())
    eps = noise_forward(params, z_t.to(dtype), t, y)
    loss = F.mse_loss(eps, noise.to(dtype))
    grads = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    grads = [torch.zeros_like(v) if g is None else g for g, v in zip(grads, params.values())]
    return loss.detach(), eps.detach(), OrderedDict(zip(params.keys(), grads))
