"""On-the-fly degradation primitives on the device -- the first slice of the OTF pipeline.

Built: `filter2d` (utils/img_process_util.py:8-32: the per-sample 21 x 21 blur / sinc kernels of realesrgan_dataset.py), the
8-bit clamp/round (`realesrgan_model.py:621`) and the training pair pool (`_dequeue_and_enqueue`, `realesrgan_model.py:404-453`).
`filter2d` is pinned to the reference function's own outputs (tests/golden/filter2d.pt).  NOT built: the stochastic stage list of
RealESRGANModel.feed_data (lens / chromatic / motion / demosaic / noise / exposure / resize / compression round-trips through PIL).
"""
from __future__ import annotations

import torch

from .. import _lib as L
from ..ops import stream_ptr


def filter2d(img: torch.Tensor, kernel: torch.Tensor) -> torch.Tensor:
    """img (b, c, h, w) fp32 on the GPU; kernel (b, k, k) -- one per sample -- or (1, k, k) for all."""
    if not img.is_cuda:
        raise RuntimeError("otf.filter2d runs on the GPU only; there is no CPU path in this package")
    k = kernel.size(-1)
    if k % 2 != 1:
        raise ValueError("Wrong kernel size")
    b, c, h, w = img.shape
    if kernel.size(0) not in (1, b):
        raise ValueError(f"kernel batch {kernel.size(0)} does not match image batch {b}")
    img = img.contiguous().float()
    kern = kernel.to(img.device, torch.float32).contiguous()
    out = torch.empty_like(img)
    L.check(L.lib().tr_filter2d(stream_ptr(), img.data_ptr(), kern.data_ptr(), 0 if kernel.size(0) == 1 else 1, b, c, h, w, k, out.data_ptr()), "tr_filter2d")
    return out


def clamp_round_u8(x: torch.Tensor) -> torch.Tensor:
    """torch.clamp((x * 255.0).round(), 0, 255) / 255.0"""
    x = x.contiguous().float()
    y = torch.empty_like(x)
    L.check(L.lib().tr_quantize_u8(stream_ptr(), x.data_ptr(), y.data_ptr(), x.numel()), "tr_quantize_u8")
    return y


class PairPool:
    """Training pair pool (realesrgan_model.py:404-453): batches are parked in a `queue_size` pool; once it is full every call
    shuffles the pool, hands out its first b pairs and replaces them with the incoming ones (more degradation diversity per batch)."""

    def __init__(self, queue_size: int = 120) -> None:
        self.queue_size, self.queue_lr, self.queue_gt, self.queue_ptr = queue_size, None, None, 0

    @torch.no_grad()
    def __call__(self, lq: torch.Tensor, gt: torch.Tensor):
        b = lq.shape[0]
        if self.queue_lr is None:
            assert self.queue_size % b == 0, f"queue size {self.queue_size} should be divisible by batch size {b}"
            self.queue_lr = torch.zeros(self.queue_size, *lq.shape[1:], device=lq.device)
            self.queue_gt = torch.zeros(self.queue_size, *gt.shape[1:], device=gt.device)
            self.queue_ptr = 0
        if self.queue_ptr == self.queue_size:
            idx = torch.randperm(self.queue_size)
            self.queue_lr, self.queue_gt = self.queue_lr[idx], self.queue_gt[idx]
            lq_out, gt_out = self.queue_lr[0:b].clone(), self.queue_gt[0:b].clone()
            self.queue_lr[0:b], self.queue_gt[0:b] = lq.clone(), gt.clone()
            return lq_out, gt_out
        self.queue_lr[self.queue_ptr:self.queue_ptr + b] = lq.clone()
        self.queue_gt[self.queue_ptr:self.queue_ptr + b] = gt.clone()
        self.queue_ptr += b
        return lq, gt
