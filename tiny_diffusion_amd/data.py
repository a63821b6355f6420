"""On-device input path: the reference feeds its training loop from a
host ``DataLoader`` over ``torchvision.datasets.MNIST`` with ``ToTensor`` +
``Normalize((0.5,), (0.5,))`` (diffusion.py:202-209, 216).  At >10 k images/s that loader
would be the bottleneck, and the whole uint8 dataset is 47 MB: keep it in HBM and fuse the
minibatch gather with the two transforms (``tdx_u8_gather_normalize``, bit-exact)."""
from __future__ import annotations

from typing import Iterator, Optional

import torch

from . import _lib
from ._lib import lib, check


class DeviceImageDataset:
    """uint8 images (N, H, W) or (N, 1, H, W) resident on the GPU; yields normalised fp32
    minibatches (B, 1, H, W) without touching the host."""

    def __init__(self, images_u8: torch.Tensor, device="cuda", mean: float = 0.5, std: float = 0.5):
        if images_u8.dtype != torch.uint8:
            raise ValueError("images must be uint8 (raw pixel values 0..255)")
        if images_u8.dim() == 4:
            if images_u8.shape[1] != 1:
                raise ValueError("single-channel images expected")
            images_u8 = images_u8[:, 0]
        if images_u8.dim() != 3:
            raise ValueError("images must be (N,H,W) or (N,1,H,W)")
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.TdxError("DeviceImageDataset lives on the GPU (no CPU fallback)")
        self.data = images_u8.contiguous().to(dev)
        self.n, self.h, self.w = self.data.shape
        if (self.h * self.w) % 4:
            raise ValueError("H*W must be a multiple of 4")
        self.mean, self.std = float(mean), float(std)

    def __len__(self):
        return self.n

    def batch(self, idx: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Normalised images for ``idx`` (int64 device tensor; None = the first len(out) or all rows)."""
        dev = self.data.device
        if idx is not None:
            idx = idx.to(dev).contiguous().to(torch.int64)
            B = idx.numel()
            if B and (int(idx.min()) < 0 or int(idx.max()) >= self.n):
                raise IndexError("dataset index out of range")
        else:
            B = self.n if out is None else out.shape[0]
        if out is None:
            out = torch.empty((B, 1, self.h, self.w), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        check(lib.tdx_u8_gather_normalize(self.data.data_ptr(), None if idx is None else idx.data_ptr(),
                                          out.data_ptr(), B, self.h * self.w, self.mean, self.std, st),
              "tdx_u8_gather_normalize")
        return out

    def epoch(self, batch_size: int, shuffle: bool = True, generator=None) -> Iterator[torch.Tensor]:
        """One pass in minibatches, like DataLoader(dataset, batch_size, shuffle=True)
        (diffusion.py:209); the permutation is drawn on the device."""
        dev = self.data.device
        order = torch.randperm(self.n, device=dev, generator=generator) if shuffle else torch.arange(self.n, device=dev)
        for i in range(0, self.n, batch_size):
            idx = order[i:i + batch_size]
            st = torch.cuda.current_stream(dev).cuda_stream
            out = torch.empty((idx.numel(), 1, self.h, self.w), dtype=torch.float32, device=dev)
            check(lib.tdx_u8_gather_normalize(self.data.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(),
                                              self.h * self.w, self.mean, self.std, st), "tdx_u8_gather_normalize")
            yield out
