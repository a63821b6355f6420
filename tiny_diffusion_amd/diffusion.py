"""Drop-in for the hot path of the reference's ``diffusion.py`` (unconditional
MNIST DDPM): ``NoiseModel``, ``ForwardProcess``, ``sample`` with the same
signatures (diffusion.py:16, 109, 166, 177, 255), computed by libtdx on MI355X."""
from __future__ import annotations

import torch

from .schedule import ForwardProcess, sample_loop
from .unet import NoiseModelBase, TIME_DIM

__all__ = ["NoiseModel", "ForwardProcess", "sample"]


class NoiseModel(NoiseModelBase):
    """UNet noise predictor eps_theta(x_t, t), diffusion.py:11-162."""

    def __init__(self, time_dim: int = TIME_DIM):
        super().__init__(time_dim=time_dim, num_classes=0)

    def forward(self, x, t):
        return self._forward_impl(x, t, None)


@torch.no_grad()
def sample(noise_model: NoiseModel, diffusion: ForwardProcess, device, n_samples=16, **kw):
    """diffusion.py:254-276: leaves the model in eval mode, returns x_0 in ~[-1, 1].
    Extra keyword arguments (x_T, noises, use_graph, philox_seed) are extensions."""
    return sample_loop(noise_model, diffusion, device, n_samples, None, **kw)
