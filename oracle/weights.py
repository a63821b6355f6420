"""Portable, seeded state_dict generator (test infrastructure).

The reference ships no trained checkpoint and a full state_dict is 44.7 MB, so
the goldens are pinned to weights that both boxes can regenerate bit-for-bit:
``numpy.random.RandomState(seed)`` (the legacy MT19937 stream numpy freezes
across versions) walks the reference's ``state_dict`` key list in registration
order (diffusion.py:16-107 / conditional_diffusion.py:19-113) and fills every
tensor.  BatchNorm statistics are deliberately non-trivial (at default init an
eval-mode BN is ~identity and would hide bugs).
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np
import torch

TIME_DIM = 256

# (name, cin, cout) of the six two-conv stages, diffusion.py:32-95
_STAGES = [
    ("enc1", 64, 128),
    ("enc2", 128, 256),
    ("enc3", 256, 512),
    ("dec3", 1024, 256),
    ("dec2", 512, 128),
    ("dec1", 256, 64),
]


def key_shapes(cond: bool, num_classes: int = 10, time_dim: int = TIME_DIM):
    """Ordered (key, shape, kind) list == reference ``state_dict()`` order."""
    out = []

    def conv(name, cin, cout, k):
        out.append((f"{name}.weight", (cout, cin, k, k), "conv_w"))
        out.append((f"{name}.bias", (cout,), "bias"))

    def bn(name, c):
        out.append((f"{name}.weight", (c,), "bn_w"))
        out.append((f"{name}.bias", (c,), "bn_b"))
        out.append((f"{name}.running_mean", (c,), "bn_rm"))
        out.append((f"{name}.running_var", (c,), "bn_rv"))
        out.append((f"{name}.num_batches_tracked", (), "bn_nbt"))

    out.append(("time_embedding.0.weight", (time_dim, 1), "lin_w"))
    out.append(("time_embedding.0.bias", (time_dim,), "bias"))
    out.append(("time_embedding.2.weight", (time_dim, time_dim), "lin_w"))
    out.append(("time_embedding.2.bias", (time_dim,), "bias"))
    if cond:
        out.append(("class_embedding.weight", (num_classes, time_dim), "emb"))
    conv("initial_conv", 1, 64, 3)
    for name, cin, cout in _STAGES[:3]:
        conv(f"{name}.0", cin, cout, 3)
        bn(f"{name}.1", cout)
        conv(f"{name}.3", cout, cout, 3)
        bn(f"{name}.4", cout)
    conv("bottleneck.0", 512, 512, 3)
    bn("bottleneck.1", 512)
    for name, cin, cout in _STAGES[3:]:
        conv(f"{name}.0", cin, cout, 3)
        bn(f"{name}.1", cout)
        conv(f"{name}.3", cout, cout, 3)
        bn(f"{name}.4", cout)
    conv("final_conv", 64, 1, 3)
    conv("time_proj1", time_dim, 128, 1)
    conv("time_proj2", time_dim, 256, 1)
    conv("time_proj3", time_dim, 512, 1)
    return out


def make_state_dict(seed: int = 0, cond: bool = False, time_scale: float = 1.0, time_dim: int = TIME_DIM):
    """Reference-format state_dict (OIHW conv weights, fp32, int64 counters).

    ``time_scale`` multiplies ``time_embedding.0.weight`` only; 1.0 reproduces
    the reference's un-normalised-t regime (pre-activations ~ t * N(0,2)).
    """
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for key, shape, kind in key_shapes(cond, time_dim=time_dim):
        if kind in ("conv_w", "lin_w"):
            fan_in = int(np.prod(shape[1:]))
            a = rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)
            if key == "time_embedding.0.weight":
                # raw t in [0, 1000) feeds this layer (diffusion.py:111); keep the
                # pre-activation O(1)..O(10) so SiLU is exercised in its curved part
                a = a * (time_scale / 100.0)
        elif kind == "bias":
            a = rs.standard_normal(shape) * 0.01
        elif kind == "bn_w":
            a = 1.0 + rs.standard_normal(shape) * 0.1
        elif kind == "bn_b":
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_rm":
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_rv":
            a = 1.0 + np.abs(rs.standard_normal(shape) * 0.1)
        elif kind == "bn_nbt":
            sd[key] = torch.tensor(1, dtype=torch.int64)
            continue
        elif kind == "emb":
            a = rs.standard_normal(shape)
        else:  # pragma: no cover
            raise AssertionError(kind)
        sd[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


def state_dict_sha256(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        if v.dtype == torch.float32:
            h.update(v.contiguous().numpy().tobytes())
    return h.hexdigest()


# --------------------------------------------------------------------------
# conditional_diffusion_laion.py NoiseModel: latent UNet on
# (4,32,32) with 768-d sinusoidal time embedding + additive text conditioning
# --------------------------------------------------------------------------
LAION_TIME_DIM = 768
_LAION_STAGES = [
    ("enc1", 32, 64),
    ("enc2", 64, 128),
    ("enc3", 128, 256),
    ("dec3", 512, 256),
    ("dec2", 384, 128),
    ("dec1", 192, 64),
]


def key_shapes_laion(time_dim: int = LAION_TIME_DIM):
    """Ordered (key, shape, kind) == state_dict() of the reference's LAION NoiseModel
    (conditional_diffusion_laion.py:234-301)."""
    out = []

    def conv(name, cin, cout, k):
        out.append((f"{name}.weight", (cout, cin, k, k), "conv_w"))
        out.append((f"{name}.bias", (cout,), "bias"))

    def bn(name, c):
        out.append((f"{name}.weight", (c,), "bn_w"))
        out.append((f"{name}.bias", (c,), "bn_b"))
        out.append((f"{name}.running_mean", (c,), "bn_rm"))
        out.append((f"{name}.running_var", (c,), "bn_rv"))
        out.append((f"{name}.num_batches_tracked", (), "bn_nbt"))

    out.append(("time_mlp.0.weight", (time_dim, time_dim), "lin_w"))
    out.append(("time_mlp.0.bias", (time_dim,), "bias"))
    out.append(("time_mlp.2.weight", (time_dim, time_dim), "lin_w"))
    out.append(("time_mlp.2.bias", (time_dim,), "bias"))
    conv("initial_conv", 4, 32, 3)
    for name, cin, cout in _LAION_STAGES[:3]:
        conv(f"{name}.0", cin, cout, 3)
        bn(f"{name}.1", cout)
        conv(f"{name}.3", cout, cout, 3)
        bn(f"{name}.4", cout)
    conv("bottleneck.0", 256, 256, 3)
    bn("bottleneck.1", 256)
    for name, cin, cout in _LAION_STAGES[3:]:
        conv(f"{name}.0", cin, cout, 3)
        bn(f"{name}.1", cout)
        conv(f"{name}.3", cout, cout, 3)
        bn(f"{name}.4", cout)
    conv("final_conv", 64, 4, 3)
    conv("time_proj1", time_dim, 64, 1)
    conv("time_proj2", time_dim, 128, 1)
    conv("time_proj3", time_dim, 256, 1)
    return out


def make_state_dict_laion(seed: int = 0, time_dim: int = LAION_TIME_DIM):
    """Seeded reference-format state_dict of the LAION NoiseModel (same recipe as
    make_state_dict: portable numpy legacy RNG, non-trivial BN statistics)."""
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for key, shape, kind in key_shapes_laion(time_dim):
        if kind in ("conv_w", "lin_w"):
            fan_in = int(np.prod(shape[1:]))
            a = rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        elif kind == "bias":
            a = rs.standard_normal(shape) * 0.01
        elif kind == "bn_w":
            a = 1.0 + rs.standard_normal(shape) * 0.1
        elif kind == "bn_b":
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_rm":
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_rv":
            a = 1.0 + np.abs(rs.standard_normal(shape) * 0.1)
        elif kind == "bn_nbt":
            sd[key] = torch.tensor(1, dtype=torch.int64)
            continue
        else:  # pragma: no cover
            raise AssertionError(kind)
        sd[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


# --------------------------------------------------------------------------
# latent_diffusion.py NoiseModel: MLP on 20-d VAE latents,
# and the MLP VAE of vae.py:37-62
# --------------------------------------------------------------------------
# (name, in, mid/out) of the two-Linear stages, latent_diffusion.py:37-97
_LATENT_STAGES = [
    ("enc1", 512, 512, 256),
    ("enc2", 256, 256, 128),
    ("enc3", 128, 128, 64),
    ("dec3", 128, 128, 128),
    ("dec2", 256, 256, 256),
    ("dec1", 512, 512, 512),
]


def key_shapes_latent(num_classes: int = 10, time_dim: int = TIME_DIM, latent_dim: int = 20):
    """Ordered (key, shape, kind) == state_dict() of latent_diffusion.NoiseModel
    (latent_diffusion.py:16-105)."""
    out = []

    def lin(name, cin, cout):
        out.append((f"{name}.weight", (cout, cin), "lin_w"))
        out.append((f"{name}.bias", (cout,), "bias"))

    def bn(name, c):
        out.append((f"{name}.weight", (c,), "bn_w"))
        out.append((f"{name}.bias", (c,), "bn_b"))
        out.append((f"{name}.running_mean", (c,), "bn_rm"))
        out.append((f"{name}.running_var", (c,), "bn_rv"))
        out.append((f"{name}.num_batches_tracked", (), "bn_nbt"))

    def stage(name, cin, mid, cout):
        lin(f"{name}.0", cin, mid)
        bn(f"{name}.1", mid)
        lin(f"{name}.3", mid, cout)
        bn(f"{name}.4", cout)

    lin("time_embedding.0", 1, time_dim)
    lin("time_embedding.2", time_dim, time_dim)
    out.append(("class_embedding.weight", (num_classes, time_dim), "emb"))
    lin("initial_fc", latent_dim, 512)
    for s in _LATENT_STAGES[:3]:
        stage(*s)
    lin("bottleneck.0", 64, 64)
    bn("bottleneck.1", 64)
    for s in _LATENT_STAGES[3:]:
        stage(*s)
    lin("final_fc", 512, latent_dim)
    lin("time_proj1", time_dim, 64)
    lin("time_proj2", time_dim, 128)
    lin("time_proj3", time_dim, 256)
    return out


def _fill(rs, key_shape_kinds, time_first=None):
    sd = OrderedDict()
    for key, shape, kind in key_shape_kinds:
        if kind in ("conv_w", "lin_w"):
            fan_in = int(np.prod(shape[1:]))
            a = rs.standard_normal(shape) * np.sqrt(2.0 / fan_in)
            if key == time_first:
                a = a / 100.0  # raw t in [0, 1000) feeds this layer (latent_diffusion.py:108-109)
        elif kind == "bias":
            a = rs.standard_normal(shape) * 0.01
        elif kind == "bn_w":
            a = 1.0 + rs.standard_normal(shape) * 0.1
        elif kind == "bn_b":
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_rm":
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_rv":
            a = 1.0 + np.abs(rs.standard_normal(shape) * 0.1)
        elif kind == "bn_nbt":
            sd[key] = torch.tensor(1, dtype=torch.int64)
            continue
        elif kind == "emb":
            a = rs.standard_normal(shape)
        else:  # pragma: no cover
            raise AssertionError(kind)
        sd[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return sd


def make_state_dict_latent(seed: int = 0):
    """Seeded reference-format state_dict of latent_diffusion.NoiseModel."""
    return _fill(np.random.RandomState(seed), key_shapes_latent(), "time_embedding.0.weight")


def key_shapes_vae(input_dim: int = 784, hidden_dim: int = 400, latent_dim: int = 20):
    """state_dict() of vae.VAE (vae.py:42-49)."""
    out = []
    for name, cin, cout in (("fc1", input_dim, hidden_dim), ("fc21", hidden_dim, latent_dim),
                            ("fc22", hidden_dim, latent_dim), ("fc3", latent_dim, hidden_dim),
                            ("fc4", hidden_dim, input_dim)):
        out.append((f"{name}.weight", (cout, cin), "lin_w"))
        out.append((f"{name}.bias", (cout,), "bias"))
    return out


def make_state_dict_vae(seed: int = 0):
    return _fill(np.random.RandomState(seed), key_shapes_vae())


# --------------------------------------------------------------------------
# diffusion_transformer.py NoiseModel: 4 post-norm blocks on a length-1 sequence
# --------------------------------------------------------------------------
def key_shapes_transformer(time_dim: int = 256, num_classes: int = 10, latent_dim: int = 20, num_layers: int = 4):
    """Ordered (key, shape, kind) == state_dict() of diffusion_transformer.NoiseModel (38-80)."""
    D = time_dim
    out = [("pos_encoding", (1, 1, D), "emb")]   # nn.Parameter registered on the module itself comes first
    for name, cin, cout in (("time_embedding.0", 1, D), ("time_embedding.2", D, D)):
        out += [(f"{name}.weight", (cout, cin), "lin_w"), (f"{name}.bias", (cout,), "bias")]
    out.append(("class_embedding.weight", (num_classes, D), "emb"))
    out += [("input_proj.weight", (D, latent_dim), "lin_w"), ("input_proj.bias", (D,), "bias")]
    for i in range(num_layers):
        p = f"transformer_blocks.{i}"
        out += [(f"{p}.attention.in_proj_weight", (3 * D, D), "lin_w"), (f"{p}.attention.in_proj_bias", (3 * D,), "bias"),
                (f"{p}.attention.out_proj.weight", (D, D), "lin_w"), (f"{p}.attention.out_proj.bias", (D,), "bias"),
                (f"{p}.norm1.weight", (D,), "bn_w"), (f"{p}.norm1.bias", (D,), "bn_b"),
                (f"{p}.ff.0.weight", (4 * D, D), "lin_w"), (f"{p}.ff.0.bias", (4 * D,), "bias"),
                (f"{p}.ff.2.weight", (D, 4 * D), "lin_w"), (f"{p}.ff.2.bias", (D,), "bias"),
                (f"{p}.norm2.weight", (D,), "bn_w"), (f"{p}.norm2.bias", (D,), "bn_b")]
    out += [("final_layer.0.weight", (D,), "bn_w"), ("final_layer.0.bias", (D,), "bn_b"),
            ("final_layer.1.weight", (latent_dim, D), "lin_w"), ("final_layer.1.bias", (latent_dim,), "bias")]
    return out


def make_state_dict_transformer(seed: int = 0):
    sd = _fill(np.random.RandomState(seed), key_shapes_transformer())
    sd["pos_encoding"] = sd["pos_encoding"] * 0.5
    return sd
