"""Validation metrics on the device: `calculate_psnr`, `calculate_ssim` under the reference's registry names
(/root/reference/traiNNer/metrics/psnr_ssim.py:13-61,101-156) and `calculate_metric` (metrics/__init__.py).

The reference rounds both images to 8-bit levels (`tensor2img`), crops, optionally takes BT.601 luma and computes in float64 numpy /
OpenCV on the host.  Here the same arithmetic runs in HIP (`csrc/metrics.hip`: quantise, crop, luma, double-precision sums, an
11 x 11 Gaussian(1.5) window applied to the valid region) on fp32 (1, C, H, W) tensors in [0, 1]; only per-block partial sums come back.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from .. import _lib as L
from ..ops import stream_ptr
from ..utils.registry import METRIC_REGISTRY

_G11 = None


def _gauss11():
    """cv2.getGaussianKernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)), normalised, float64."""
    global _G11
    if _G11 is None:
        g = [math.exp(-((i - 5) ** 2) / (2 * 1.5 ** 2)) for i in range(11)]
        s = sum(g)
        _G11 = (C.c_double * 11)(*[v / s for v in g])
    return _G11


def _planes(img: torch.Tensor, crop_border: int, test_y_channel: bool):
    if img.dim() == 4:
        assert img.shape[0] == 1, "metrics take one image at a time (batch 1), as the reference's validation loop does"
        img = img[0]
    img = img.contiguous().float()
    if not img.is_cuda:
        raise RuntimeError("metrics run on the GPU only; there is no CPU path in this package")
    c, h, w = img.shape
    cc = 1 if (test_y_channel and c == 3) else c
    out = torch.empty(cc, h - 2 * crop_border, w - 2 * crop_border, dtype=torch.float32, device=img.device)
    L.check(L.lib().tr_metric_prepare(stream_ptr(), img.data_ptr(), c, h, w, crop_border, 1 if test_y_channel else 0, out.data_ptr()), "tr_metric_prepare")
    return out


def _finish(partials: torch.Tensor, n: int) -> float:
    return float(sum(partials[:n].cpu().tolist()))          # fixed order: reproducible


@METRIC_REGISTRY.register()
def calculate_psnr(img: torch.Tensor, img2: torch.Tensor, crop_border: int, test_y_channel: bool = False, **kwargs) -> float:
    assert img.shape == img2.shape, f"Image shapes are different: {img.shape}, {img2.shape}."
    a, b = _planes(img, crop_border, test_y_channel), _planes(img2, crop_border, test_y_channel)
    part = torch.empty(1024, dtype=torch.float64, device=a.device)
    n = C.c_int(0)
    L.check(L.lib().tr_metric_sqerr(stream_ptr(), a.data_ptr(), b.data_ptr(), a.numel(), part.data_ptr(), C.byref(n)), "tr_metric_sqerr")
    mse = _finish(part, n.value) / a.numel()
    return float("inf") if mse == 0 else 10.0 * math.log10(255.0 * 255.0 / mse)


@METRIC_REGISTRY.register()
def calculate_ssim(img: torch.Tensor, img2: torch.Tensor, crop_border: int, test_y_channel: bool = False, **kwargs) -> float:
    assert img.shape == img2.shape, f"Image shapes are different: {img.shape}, {img2.shape}."
    a, b = _planes(img, crop_border, test_y_channel), _planes(img2, crop_border, test_y_channel)
    nc, h, w = a.shape
    part = torch.empty(1024, dtype=torch.float64, device=a.device)
    n = C.c_int(0)
    L.check(L.lib().tr_metric_ssim(stream_ptr(), a.data_ptr(), b.data_ptr(), nc, h, w, _gauss11(), part.data_ptr(), C.byref(n)), "tr_metric_ssim")
    return _finish(part, n.value) / (nc * (h - 10) * (w - 10))


def calculate_metric(data: dict, opt: dict, device=None) -> float:
    """metrics/__init__.py `calculate_metric`: opt = {type: calculate_psnr, crop_border: 4, test_y_channel: false, ...}."""
    opt = dict(opt)
    fn = METRIC_REGISTRY.get(opt.pop("type"))
    opt.pop("better", None)
    return fn(**data, **opt)
