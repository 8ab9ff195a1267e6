"""OTF degradation stages -- CPU restatement (test infrastructure only, see oracle/__init__.py).

Follows, function by function,
  resize_pt / _resample / _lanczos / _ramp / _sinc     /root/reference/traiNNer/data/degradations.py:961-1021
  filter2d                                             /root/reference/traiNNer/utils/img_process_util.py:8-32   (pinned: tests/golden/filter2d.pt)
  motion blur / exposure / colour temperature / sensor noise        /root/reference/traiNNer/models/paragon_otf_degradations.py:251-294,345-414
  oversharpening, aliasing                             /root/reference/traiNNer/models/realesrgan_model.py:193-241,365-402
  demosaicing artifacts                                /root/reference/traiNNer/models/realesrgan_model.py:312-363 (cv2.demosaicing restated: unpinned)
  the stage order and RNG draws of feed_data           /root/reference/traiNNer/models/realesrgan_model.py:456-664
Pinned (round 3): models/paragon_otf_degradations.py DOES load by path here (numpy / torch / PIL + a two-line RNG stand-in), so every codec-free
static stage -- motion blur, lens distortion, exposure, colour temperature, sensor noise, rolling shutter, oversharpening, chromatic aberration,
aliasing -- is checked against the reference function's own seeded outputs (tests/golden/paragon_otf.pt, tests/test_twins_cpu.py); resizes and
filter2d against theirs (otf_degradations.pt, filter2d.pt).  Still unpinned: the demosaicing stage (cv2.demosaicing is absent) and the codec
round trips (PIL encoders; statistical only, as in the reference).  RealESRGANModel.feed_data itself has no CPU path (hard-coded .cuda(), SURVEY.md
section 9): the stage ORDER is restated from realesrgan_model.py:456-664.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch.nn import functional as F

ANTIALIAS_MODES = {"bicubic", "bilinear"}


def _sinc(x):
    return torch.where(x != 0, torch.sin(math.pi * x) / (math.pi * x), x.new_ones([]))


def _lanczos(x, a):
    cond = torch.logical_and(-a < x, x < a)
    out = torch.where(cond, _sinc(x) * _sinc(x / a), x.new_zeros([]))
    return out / out.sum()


def _ramp(ratio, width):
    n = math.ceil(width / ratio + 1)
    out = torch.empty([n])
    cur = 0
    for i in range(out.shape[0]):
        out[i] = cur
        cur += ratio
    return torch.cat([-out[1:].flip([0]), out])[1:-1]


def _resample(inp, size, a=3):
    n, c, h, w = inp.shape
    dh, dw = size
    inp = inp.reshape([n * c, 1, h, w])
    if dh < h:
        kh = _lanczos(_ramp(dh / h, a), a).to(inp.device, inp.dtype)
        ph = (kh.shape[0] - 1) // 2
        inp = F.conv2d(F.pad(inp, (0, 0, ph, ph), "reflect"), kh[None, None, :, None], padding=0)
    if dw < w:
        kw = _lanczos(_ramp(dw / w, a), a).to(inp.device, inp.dtype)
        pw = (kw.shape[0] - 1) // 2
        inp = F.conv2d(F.pad(inp, (pw, pw, 0, 0), "reflect"), kw[None, None, None, :], padding=0)
    inp = inp.view([n, c, h, w])
    return F.interpolate(inp, size, mode="bicubic", align_corners=False).clamp(0, 1)


def resize_pt(img, mode, scale_factor=0, size=(0, 0)):
    if scale_factor == 0 and tuple(size) == (0, 0):
        raise ValueError("scale_factor or size is required")
    if scale_factor != 0:
        size = (round(img.shape[2] * scale_factor), round(img.shape[3] * scale_factor))
    if mode == "lanczos":
        return _resample(img, size)
    return F.interpolate(img, size=size, mode=mode, antialias=mode in ANTIALIAS_MODES).clamp(0, 1)


def filter2d(img, kernel):
    k = kernel.size(-1)
    b, c, h, w = img.size()
    if k % 2 == 1:
        img = F.pad(img, (k // 2, k // 2, k // 2, k // 2), mode="reflect")
    else:
        raise ValueError("Wrong kernel size")
    ph, pw = img.size()[-2:]
    if kernel.size(0) == 1:
        img = img.view(b * c, 1, ph, pw)
        kernel = kernel.view(1, 1, k, k)
        return F.conv2d(img, kernel, padding=0).view(b, c, h, w)
    img = img.view(1, b * c, ph, pw)
    kernel = kernel.view(b, 1, k, k).repeat(1, c, 1, 1).view(b * c, 1, k, k)
    return F.conv2d(img, kernel, groups=b * c).view(b, c, h, w)


def motion_blur(img, kernel, ks):
    return F.conv2d(img, kernel.reshape(1, 1, ks, ks).repeat(img.size(1), 1, 1, 1), padding=ks // 2, groups=img.size(1))


def exposure(img, factor):
    return torch.clamp(img * factor, 0, 1)


def color_temperature(img, shift):
    r, g, b = img[:, 0:1], img[:, 1:2], img[:, 2:3]
    if shift > 0:
        r = r * (1 + shift * 0.3)
        g = g * (1 + shift * 0.1)
    else:
        b = b * (1 - shift * 0.3)
        g = g * (1 - shift * 0.1)
    return torch.clamp(torch.cat([r, g, b], dim=1), 0, 1)


def sensor_noise(img, noise, std):
    return torch.clamp(img + noise * std, 0, 1)


def oversharpen(img, strength):
    wt = torch.ones(1, 1, 5, 5) / 25
    blurred = F.conv2d(img, wt.repeat(img.size(1), 1, 1, 1), padding=2, groups=img.size(1))
    return torch.clamp(img + (img - blurred) * strength, 0, 1)


def aliasing(img, scale_factor):
    h, w = img.shape[2:]
    down = F.interpolate(img, size=(int(h * scale_factor), int(w * scale_factor)), mode="nearest")
    return F.interpolate(down, size=(h, w), mode="nearest")


def clamp_round(x):
    return torch.clamp((x * 255.0).round(), 0, 255) / 255.0


def lens_distortion(img, strength):
    """paragon_otf_degradations.py:310-343 (the `ij` meshgrid is stacked as (rows, cols), i.e. the grid's x coordinate runs over rows)."""
    b, _, h, w = img.shape
    gx, gy = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    r = torch.sqrt(gx ** 2 + gy ** 2)
    rd = r * (1 + strength * r ** 2)
    rd[r == 0] = 0
    r[r == 0] = 1e-6
    grid = torch.stack([gx * (rd / r), gy * (rd / r)], dim=-1).unsqueeze(0).repeat(b, 1, 1, 1)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="reflection", align_corners=False)


def rolling_shutter(img, strength):
    """paragon_otf_degradations.py:430-456."""
    b, _, h, w = img.shape
    slant = strength * h / w
    gy, gx = torch.meshgrid(torch.linspace(-1, 1, h), torch.linspace(-1, 1, w), indexing="ij")
    grid = torch.stack([gx + slant * gy, gy], dim=-1).unsqueeze(0).repeat(b, 1, 1, 1)
    return F.grid_sample(img, grid, mode="bilinear", padding_mode="reflection", align_corners=False)


def chromatic_aberration(img):
    """realesrgan_model.py:262-311."""
    b = img.shape[0]
    out = []
    for ch, s in ((0, 1.001), (1, None), (2, 0.999)):
        x = img[:, ch:ch + 1]
        if s is not None:
            theta = torch.tensor([[[s, 0, 0], [0, s, 0]]], dtype=torch.float32).repeat(b, 1, 1)
            x = F.grid_sample(x, F.affine_grid(theta, x.size(), align_corners=False), align_corners=False, mode="bilinear")
        out.append(x)
    return torch.clamp(torch.cat(out, dim=1), 0, 1)


def demosaic_artifacts(img):
    """realesrgan_model.py:327-361 for an (N, 3, H, W) float batch: 8-bit quantisation, the reference's mosaic (channel 2 at even/even sites,
    channel 0 at odd/odd, channel 1 elsewhere) and cv2.demosaicing(bayer, COLOR_BAYER_BG2BGR).  OpenCV (opencv-python, unpinned in the reference's
    pyproject) is absent here, so its bilinear demosaicing is restated from the published algorithm (imgproc/demosaicing.cpp, Bayer2RGB_):
    interior pixels average the 2 or 4 nearest same-colour sites in integers with round-half-up, the first / last columns copy their inner
    neighbour and then the first / last rows copy theirs.  PARITY UNPINNED for this function: no cv2 output is available to check it against."""
    x = img.detach().cpu().float().numpy()
    n, c, h, w = x.shape
    assert c == 3 and h >= 3 and w >= 3
    q = (np.clip(x, 0, 1) * np.float32(255)).astype(np.uint8)
    out = np.zeros((n, 3, h, w), np.uint8)
    for b in range(n):
        m = np.zeros((h, w), np.int32)
        m[0::2, 0::2] = q[b, 2, 0::2, 0::2]
        m[0::2, 1::2] = q[b, 1, 0::2, 1::2]
        m[1::2, 0::2] = q[b, 1, 1::2, 0::2]
        m[1::2, 1::2] = q[b, 0, 1::2, 1::2]
        o = np.zeros((3, h, w), np.int32)
        for y in range(1, h - 1):
            for xx in range(1, w - 1):
                ctr = m[y, xx]
                if (y + xx) & 1:
                    horiz = (m[y, xx - 1] + m[y, xx + 1] + 1) >> 1
                    vert = (m[y - 1, xx] + m[y + 1, xx] + 1) >> 1
                    o[1, y, xx] = ctr
                    if y & 1:
                        o[0, y, xx], o[2, y, xx] = horiz, vert
                    else:
                        o[2, y, xx], o[0, y, xx] = horiz, vert
                else:
                    cross = (m[y - 1, xx] + m[y, xx - 1] + m[y, xx + 1] + m[y + 1, xx] + 2) >> 2
                    diag = (m[y - 1, xx - 1] + m[y - 1, xx + 1] + m[y + 1, xx - 1] + m[y + 1, xx + 1] + 2) >> 2
                    o[1, y, xx] = cross
                    if y & 1:
                        o[0, y, xx], o[2, y, xx] = ctr, diag
                    else:
                        o[2, y, xx], o[0, y, xx] = ctr, diag
        o[:, 1:h - 1, 0] = o[:, 1:h - 1, 1]
        o[:, 1:h - 1, w - 1] = o[:, 1:h - 1, w - 2]
        o[:, 0, :] = o[:, 1, :]
        o[:, h - 1, :] = o[:, h - 2, :]
        out[b] = o.astype(np.uint8)
    return torch.from_numpy(out).float() / 255.0
