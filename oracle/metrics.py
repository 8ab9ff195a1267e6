"""PSNR / SSIM / tiled-inference restatement (test infrastructure; see oracle/__init__.py).

  calculate_psnr, calculate_ssim, _ssim   /root/reference/traiNNer/metrics/psnr_ssim.py:13-61,101-156,201-232
  tensor2img rounding                     /root/reference/traiNNer/utils/img_util.py (clamp, x255, round, uint8)
  to_y_channel / rgb2ycbcr(y_only)        /root/reference/traiNNer/metrics/metric_util.py:34-47, utils/color_util.py
  infer_tiled                             /root/reference/traiNNer/models/sr_model.py:830-900
Pinned: `psnr` (the numpy calculate_psnr incl. to_y_channel) and `psnr_pt` against the reference functions' own outputs (tests/golden/psnr.pt,
made by tests/golden/make_golden_twins.py; tests/test_twins_cpu.py).  Parity unpinned: SSIM -- the reference versions call cv2 (absent here); the
Gaussian window is cv2.getGaussianKernel's closed form and `filter2D(...)[5:-5, 5:-5]` is a valid-region correlation, restated with float64 conv2d.
"""
from __future__ import annotations

import math

import torch
from torch.nn import functional as F


def to_u8_levels(img: torch.Tensor) -> torch.Tensor:
    return (img.clamp(0, 1) * 255.0).round()


def _prep(img, crop_border, test_y_channel):
    x = to_u8_levels(img[0] if img.dim() == 4 else img)
    if crop_border:
        x = x[:, crop_border:-crop_border, crop_border:-crop_border]
    if test_y_channel and x.shape[0] == 3:
        f = x.float() / 255.0
        y = (65.481 * f[0] + 128.553 * f[1] + 24.966 * f[2] + 16.0) / 255.0
        x = (y * 255.0)[None]
    return x.double()


def psnr(img, img2, crop_border, test_y_channel=False):
    a, b = _prep(img, crop_border, test_y_channel), _prep(img2, crop_border, test_y_channel)
    mse = ((a - b) ** 2).mean().item()
    return float("inf") if mse == 0 else 10.0 * math.log10(255.0 * 255.0 / mse)


def psnr_pt(img, img2, crop_border, test_y_channel=False):
    """calculate_psnr_pt (/root/reference/traiNNer/metrics/psnr_ssim.py:64-98): (n, 3/1, h, w) in [0, 1], no 8-bit rounding, one value per image;
    Y = rgb2ycbcr_pt(y_only) (utils/color_util.py: (65.481 R + 128.553 G + 24.966 B + 16) / 255).  Pinned: tests/golden/psnr.pt."""
    if crop_border:
        img = img[:, :, crop_border:-crop_border, crop_border:-crop_border]
        img2 = img2[:, :, crop_border:-crop_border, crop_border:-crop_border]
    if test_y_channel:
        wgt = torch.tensor([[65.481], [128.553], [24.966]]).to(img)
        img = (torch.matmul(img.permute(0, 2, 3, 1), wgt).permute(0, 3, 1, 2) + 16.0) / 255.0
        img2 = (torch.matmul(img2.permute(0, 2, 3, 1), wgt).permute(0, 3, 1, 2) + 16.0) / 255.0
    mse = ((img.double() - img2.double()) ** 2).mean(dim=[1, 2, 3])
    return 10.0 * torch.log10(1.0 / (mse + 1e-8))


def ssim(img, img2, crop_border, test_y_channel=False):
    a, b = _prep(img, crop_border, test_y_channel), _prep(img2, crop_border, test_y_channel)
    g = torch.tensor([math.exp(-((i - 5) ** 2) / (2 * 1.5 ** 2)) for i in range(11)], dtype=torch.float64)
    g = g / g.sum()
    win = torch.outer(g, g)[None, None]
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for ch in range(a.shape[0]):
        x, y = a[ch][None, None], b[ch][None, None]
        f = lambda t: F.conv2d(t, win)  # noqa: E731
        mu1, mu2 = f(x), f(y)
        s1, s2, s12 = f(x * x) - mu1 ** 2, f(y * y) - mu2 ** 2, f(x * y) - mu1 * mu2
        vals.append((((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s1 + s2 + c2))).mean().item())
    return sum(vals) / len(vals)


def infer_tiled(net, lq, tile_size, tile_overlap, scale):
    b, c, h, w = lq.shape
    assert b == 1
    if h <= tile_size and w <= tile_size:
        return net(lq)
    pad_h = (tile_size - (h % tile_size)) % tile_size if h > tile_size else 0
    pad_w = (tile_size - (w % tile_size)) % tile_size if w > tile_size else 0
    lq = F.pad(lq, (0, pad_w, 0, pad_h), mode="reflect")
    _, _, hp, wp = lq.shape
    output = torch.zeros((1, c, hp * scale, wp * scale))
    wmap = torch.zeros_like(output)
    hr = tile_size * scale
    wy = 1 - torch.abs(torch.linspace(0, 1, hr) - 0.5) * 2
    weight = torch.outer(wy, wy)[None, None]
    stride = tile_size - tile_overlap
    ty = max(1, (hp - tile_overlap + stride - 1) // stride)
    tx = max(1, (wp - tile_overlap + stride - 1) // stride)
    for y in range(ty):
        for x in range(tx):
            y0, x0 = y * stride, x * stride
            patch = lq[:, :, y0:min(y0 + tile_size, hp), x0:min(x0 + tile_size, wp)]
            ph, pw = patch.shape[-2:]
            pb = max(tile_size - ph, 0) if ph < tile_size else 0
            pr = max(tile_size - pw, 0) if pw < tile_size else 0
            if pb > 0 or pr > 0:
                patch = F.pad(patch, (0, min(pr, pw - 1), 0, min(pb, ph - 1)), mode="reflect")
            out = net(patch)[:, :, : ph * scale, : pw * scale]
            wpatch = weight[:, :, : ph * scale, : pw * scale]
            output[:, :, y0 * scale:y0 * scale + ph * scale, x0 * scale:x0 * scale + pw * scale] += out * wpatch
            wmap[:, :, y0 * scale:y0 * scale + ph * scale, x0 * scale:x0 * scale + pw * scale] += wpatch
    return (output / wmap.clamp(min=1e-6))[:, :, : h * scale, : w * scale]
