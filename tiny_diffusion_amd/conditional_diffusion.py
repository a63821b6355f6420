"""Drop-in for the hot path of the reference's ``conditional_diffusion.py``
(class-conditional MNIST DDPM): same ``NoiseModel(time_dim, num_classes)``,
``forward(x, t, y)``, ``ForwardProcess`` and ``sample(..., y)`` contracts
(conditional_diffusion.py:19, 115, 174, 354-386)."""
from __future__ import annotations

import torch

from .schedule import ForwardProcess, sample_loop
from .unet import NoiseModelBase, TIME_DIM

__all__ = ["NoiseModel", "ForwardProcess", "sample"]


class NoiseModel(NoiseModelBase):
    """eps_theta(x_t, t, y): time embedding + nn.Embedding(num_classes, 256)[y],
    conditional_diffusion.py:14-172."""

    def __init__(self, time_dim: int = TIME_DIM, num_classes: int = 10):
        super().__init__(time_dim=time_dim, num_classes=num_classes)

    def forward(self, x, t, y):
        return self._forward_impl(x, t, y)


@torch.no_grad()
def sample(noise_model: NoiseModel, diffusion: ForwardProcess, device, n_samples=16, y=None, **kw):
    """conditional_diffusion.py:354-386, including its argument errors."""
    if y is None:
        raise ValueError("Class labels 'y' must be provided for conditional generation.")
    if y.shape[0] != n_samples:
        raise ValueError("y must have shape (n_samples,)")
    return sample_loop(noise_model, diffusion, device, n_samples, y, **kw)
