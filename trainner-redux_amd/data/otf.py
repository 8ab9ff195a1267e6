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


# ======================================================================================================================================
# Second slice: the stage list of RealESRGANModel.feed_data (models/realesrgan_model.py:456-664) on the device.
# Resizes are separable gathers (tr_resample_axis) over (index, weight) tables built here on the host -- a few hundred floats per call.
# ======================================================================================================================================
import io
import logging
import math
import random

import numpy as np

from .. import ops as _ops

ANTIALIAS_MODES = {"bicubic", "bilinear"}        # data/degradations.py:960


def _aa_filter(mode: str, x: np.ndarray) -> np.ndarray:
    """ATen's antialias filters (UpSampleKernel.cpp HelperInterpLinear / HelperInterpCubic::aa_filter, a = -0.5)."""
    x = np.abs(x)
    if mode == "bilinear":
        return np.where(x < 1.0, 1.0 - x, 0.0)
    a = -0.5
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0, np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))


def _tables_aa(mode: str, n_in: int, n_out: int):
    """F.interpolate(mode, antialias=True, align_corners=False): per output index a window [xmin, xmin + xsize) of normalised weights
    (ATen `_compute_indices_min_size_weights_aa`)."""
    interp = 2 if mode == "bilinear" else 4
    scale = n_in / n_out
    support = interp * 0.5 * scale if scale >= 1.0 else interp * 0.5
    inv = 1.0 / scale if scale >= 1.0 else 1.0
    taps = int(math.ceil(support)) * 2 + 1
    idx = np.zeros((n_out, taps), np.int32)
    wts = np.zeros((n_out, taps), np.float32)
    for i in range(n_out):
        center = scale * (i + 0.5)
        xmin = max(int(center - support + 0.5), 0)
        xsize = min(int(center + support + 0.5), n_in) - xmin
        j = np.arange(xsize)
        w = _aa_filter(mode, (j + xmin - center + 0.5) * inv).astype(np.float32)
        tot = w.sum(dtype=np.float32)
        if tot != 0:
            w = w / tot
        idx[i, :] = xmin
        idx[i, :xsize] = xmin + j
        wts[i, :xsize] = w
    return idx, wts


def _tables_nearest(n_in: int, n_out: int, exact: bool):
    """mode="nearest-exact": floor((i + 0.5) * scale); mode="nearest" (legacy): floor(i * scale); scale = n_in / n_out in fp32 like ATen."""
    scale = np.float32(n_in) / np.float32(n_out)
    i = np.arange(n_out, dtype=np.float32)
    src = np.floor((i + np.float32(0.5)) * scale) if exact else np.floor(i * scale)
    idx = np.minimum(src.astype(np.int64), n_in - 1).astype(np.int32).reshape(n_out, 1)
    return idx, np.ones((n_out, 1), np.float32)


def _tables_bicubic(n_in: int, n_out: int):
    """F.interpolate(mode="bicubic", align_corners=False) without antialias: 4 taps, A = -0.75, border by index clamping."""
    A = -0.75
    scale = n_in / n_out
    idx = np.zeros((n_out, 4), np.int32)
    wts = np.zeros((n_out, 4), np.float32)
    for i in range(n_out):
        real = np.float32(scale) * np.float32(i + 0.5) - np.float32(0.5)
        fl = math.floor(real)
        t = float(np.float32(real) - np.float32(fl))

        def c1(x):
            return ((A + 2) * x - (A + 3)) * x * x + 1

        def c2(x):
            return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A

        wts[i] = [c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)]
        idx[i] = [min(max(fl - 1 + k, 0), n_in - 1) for k in range(4)]
    return idx, wts


def _lanczos_kernel(ratio: float, a: int = 3) -> np.ndarray:
    """`_lanczos(_ramp(ratio, a), a)` (data/degradations.py:967-984), fp32 like the reference's CPU tensors."""
    n = math.ceil(a / ratio + 1)
    out = np.empty(n, np.float32)
    cur = 0.0
    for i in range(n):
        out[i] = cur
        cur += ratio
    x = np.concatenate([-out[1:][::-1], out])[1:-1].astype(np.float32)

    def sinc(v):
        pv = np.float32(math.pi) * v
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(v != 0, np.sin(pv) / pv, np.float32(1.0)).astype(np.float32)

    cond = (-a < x) & (x < a)
    k = np.where(cond, sinc(x) * sinc(x / np.float32(a)), np.float32(0.0)).astype(np.float32)
    return (k / k.sum(dtype=np.float32)).astype(np.float32)


def _tables_reflect_conv(n: int, kernel: np.ndarray):
    """F.pad(reflect, (L - 1) // 2) + F.conv2d with a 1-D kernel: same length out, taps = len(kernel)."""
    L = len(kernel)
    pad = (L - 1) // 2
    pos = np.arange(n)[:, None] + np.arange(L)[None, :] - pad
    pos = np.where(pos < 0, -pos, pos)
    pos = np.where(pos >= n, 2 * (n - 1) - pos, pos)
    if pos.min() < 0 or pos.max() >= n:
        raise ValueError("Padding size should be less than the corresponding input dimension")     # torch's reflect-pad error
    return pos.astype(np.int32), np.broadcast_to(kernel[None, :], (n, L)).astype(np.float32).copy()


_table_cache: dict = {}


def _dev_tables(key, build, device):
    k = (key, str(device))
    t = _table_cache.get(k)
    if t is None:
        idx, wts = build()
        t = (torch.from_numpy(np.ascontiguousarray(idx)).to(device), torch.from_numpy(np.ascontiguousarray(wts)).to(device), idx.shape[1])
        if len(_table_cache) > 512:
            _table_cache.clear()
        _table_cache[k] = t
    return t


def _resample(img: torch.Tensor, axis: int, out_len: int, tables, clamp: bool) -> torch.Tensor:
    idx, wts, taps = tables
    b, c, h, w = img.shape
    out = torch.empty((b, c, out_len, w) if axis == 0 else (b, c, h, out_len), dtype=torch.float32, device=img.device)
    L.check(L.lib().tr_resample_axis(stream_ptr(), img.data_ptr(), out.data_ptr(), b * c, h, w, axis, out_len, idx.data_ptr(), wts.data_ptr(), taps,
                                     1 if clamp else 0), "tr_resample_axis")
    return out


def resize_pt(img: torch.Tensor, mode: str, scale_factor: float = 0, size: tuple[int, int] = (0, 0), clamp: bool = True) -> torch.Tensor:
    """data/degradations.py:996-1021.  Modes: bilinear / bicubic (antialiased), nearest-exact, nearest (clamp=False: the aliasing stage),
    lanczos (reflect-padded Lanczos-3 prefilter on the axes that shrink, then bicubic)."""
    if not img.is_cuda:
        raise RuntimeError("otf.resize_pt runs on the GPU only; there is no CPU path in this package")
    if scale_factor == 0 and tuple(size) == (0, 0):
        raise ValueError("scale_factor or size is required")
    if scale_factor != 0:
        size = (round(img.shape[2] * scale_factor), round(img.shape[3] * scale_factor))
    img = img.contiguous().float()
    h, w = img.shape[2:]
    dh, dw = int(size[0]), int(size[1])
    dev = img.device
    if mode == "lanczos":
        if dh < h:
            img = _resample(img, 0, h, _dev_tables(("lz", h, dh), lambda: _tables_reflect_conv(h, _lanczos_kernel(dh / h)), dev), False)
        if dw < w:
            img = _resample(img, 1, w, _dev_tables(("lz", w, dw), lambda: _tables_reflect_conv(w, _lanczos_kernel(dw / w)), dev), False)
        img = _resample(img, 1, dw, _dev_tables(("bc", w, dw), lambda: _tables_bicubic(w, dw), dev), False)
        return _resample(img, 0, dh, _dev_tables(("bc", h, dh), lambda: _tables_bicubic(h, dh), dev), True)
    if mode in ANTIALIAS_MODES:
        mk = lambda n_in, n_out: _dev_tables((mode, n_in, n_out), lambda: _tables_aa(mode, n_in, n_out), dev)  # noqa: E731
    elif mode in ("nearest-exact", "nearest"):
        mk = lambda n_in, n_out: _dev_tables((mode, n_in, n_out), lambda: _tables_nearest(n_in, n_out, mode == "nearest-exact"), dev)  # noqa: E731
    else:
        raise NotImplementedError(f"resize mode {mode!r} is not built on the HIP path (bilinear, bicubic, nearest-exact, nearest, lanczos)")
    img = _resample(img, 1, dw, mk(w, dw), False)
    return _resample(img, 0, dh, mk(h, dh), clamp)


def crop(img: torch.Tensor, top: int, left: int, h: int, w: int) -> torch.Tensor:
    img = img.contiguous().float()
    b, c, ih, iw = img.shape
    out = torch.empty(b, c, h, w, dtype=torch.float32, device=img.device)
    L.check(L.lib().tr_crop_nchw(stream_ptr(), img.data_ptr(), out.data_ptr(), b * c, ih, iw, top, left, h, w), "tr_crop_nchw")
    return out


def paired_random_crop(img_gt: torch.Tensor, img_lq: torch.Tensor, gt_patch_size: int, scale: int, gt_path: str | None = None, rnd=random):
    """Tensor branch of data/transforms.py:69-137 (same checks, same two `random.randint` draws)."""
    h_lq, w_lq = img_lq.shape[-2:]
    h_gt, w_gt = img_gt.shape[-2:]
    lq_patch_size = gt_patch_size // scale
    if h_gt != h_lq * scale or w_gt != w_lq * scale:
        raise ValueError(f"Scale mismatches. GT ({h_gt}, {w_gt}) is not {scale}x ", f"multiplication of LQ ({h_lq}, {w_lq}). {gt_path}")
    if h_lq < lq_patch_size or w_lq < lq_patch_size:
        raise ValueError(f"LQ ({h_lq}, {w_lq}) is smaller than patch size ", f"({lq_patch_size}, {lq_patch_size}). ", f"Please remove {gt_path}.")
    top = rnd.randint(0, h_lq - lq_patch_size)
    left = rnd.randint(0, w_lq - lq_patch_size)
    lq = crop(img_lq, top, left, lq_patch_size, lq_patch_size)
    gt = crop(img_gt, int(top * scale), int(left * scale), gt_patch_size, gt_patch_size)
    return gt, lq


def gain_noise_clamp(img: torch.Tensor, gains, noise: torch.Tensor | None = None, noise_std: float = 0.0) -> torch.Tensor:
    """clamp(img * gain[channel] + noise * noise_std, 0, 1)."""
    img = img.contiguous().float()
    b, c, h, w = img.shape
    g = [float(gains)] * c if isinstance(gains, (int, float)) else [float(v) for v in gains]
    out = torch.empty_like(img)
    import ctypes as C
    L.check(L.lib().tr_gain_noise_clamp(stream_ptr(), img.data_ptr(), noise.data_ptr() if noise is not None else None, out.data_ptr(), b, c, h * w,
                                        (C.c_float * c)(*g), float(noise_std)), "tr_gain_noise_clamp")
    return out


def conv_shared_zeropad(img: torch.Tensor, kernel: torch.Tensor, pad: int) -> torch.Tensor:
    """F.conv2d(img, kernel.repeat(c, 1, 1, 1), padding=pad, groups=c) for one k x k kernel."""
    img = img.contiguous().float()
    b, c, h, w = img.shape
    k = kernel.shape[-1]
    kern = kernel.reshape(k, k).to(img.device, torch.float32).contiguous()
    oh, ow = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    out = torch.empty(b, c, oh, ow, dtype=torch.float32, device=img.device)
    L.check(L.lib().tr_filter2d_zeropad(stream_ptr(), img.data_ptr(), kern.data_ptr(), b * c, h, w, k, pad, out.data_ptr()), "tr_filter2d_zeropad")
    return out


def axpby_clamp01(a: torch.Tensor, wa: float, b: torch.Tensor, wb: float) -> torch.Tensor:
    a, b = a.contiguous().float(), b.contiguous().float()
    assert a.shape == b.shape
    out = torch.empty_like(a)
    L.check(L.lib().tr_axpby_clamp01(stream_ptr(), a.data_ptr(), float(wa), b.data_ptr(), float(wb), out.data_ptr(), a.numel()), "tr_axpby_clamp01")
    return out


def warp_bilinear(img: torch.Tensor, mode: int, params) -> torch.Tensor:
    """grid_sample(bilinear, align_corners=False) over the analytic grids of the lens (0) / rolling-shutter (1) / chromatic (2) stages."""
    import ctypes as C

    img = img.contiguous().float()
    b, c, h, w = img.shape
    p = [float(v) for v in params] + [1.0] * (4 - len(params))
    out = torch.empty_like(img)
    L.check(L.lib().tr_warp_bilinear(stream_ptr(), img.data_ptr(), out.data_ptr(), b, c, h, w, mode, (C.c_float * 4)(*p)), "tr_warp_bilinear")
    return out


def bayer_round_trip(img: torch.Tensor) -> torch.Tensor:
    """Demosaicing artifacts (realesrgan_model.py:312-363): 8-bit mosaic + OpenCV-bilinear reconstruction of a 3-channel batch, on the device."""
    img = img.contiguous().float()
    b, c, h, w = img.shape
    if c != 3:
        raise ValueError("the demosaicing stage indexes three colour channels (realesrgan_model.py:342-345)")
    out = torch.empty_like(img)
    L.check(L.lib().tr_bayer_round_trip(stream_ptr(), img.data_ptr(), out.data_ptr(), b, h, w), "tr_bayer_round_trip")
    return out


def motion_blur_kernel(kernel_size: int, angle: float) -> torch.Tensor:
    """models/paragon_otf_degradations.py:276-294."""
    center = kernel_size // 2
    ca, sa = math.cos(math.radians(angle)), math.sin(math.radians(angle))
    k = torch.zeros(kernel_size, kernel_size)
    for i in range(kernel_size):
        for j in range(kernel_size):
            if abs((i - center) * ca + (j - center) * sa) < 0.5:
                k[i, j] = 1.0
    return k / k.sum()


def compress_round_trip(img: torch.Tensor, format_name: str, quality: float, host: torch.Tensor | None = None) -> torch.Tensor:
    """`ParagonOTF._compress_with_format` body (models/paragon_otf_degradations.py:119-158): a per-sample PIL encode / decode on the HOST --
    the reference does exactly this (`.cpu()` -> uint8 -> codec -> back), so this stage is codec-bound, not a kernel; a format whose PIL
    plugin is missing (HEIF here) passes the sample through unchanged, as in the reference."""
    from PIL import Image

    fmt = {"jpeg": "JPEG", "webp": "WEBP", "avif": "AVIF", "heif": "HEIF"}.get(format_name)
    if host is None:                     # `host`: the same pixels already copied to (pinned) host memory by an earlier async D2H
        host = img.detach().float().cpu()
    if fmt is None:
        return host.to(img.device)

    def one(i: int) -> torch.Tensor:
        arr = (host[i].clamp(0, 1).numpy() * 255).astype("uint8").transpose(1, 2, 0)
        if arr.shape[2] == 1:
            arr = arr[:, :, 0]
        buf = io.BytesIO()
        try:
            Image.fromarray(arr).save(buf, format=fmt, quality=int(quality))
            buf.seek(0)
            comp = np.array(Image.open(buf).convert("RGB"))
            return torch.from_numpy(comp).float().div(255.0).permute(2, 0, 1)
        except Exception:
            return host[i]               # on error (or a missing plugin), use the original -- :156-158

    # the samples of a batch are independent and PIL's codecs release the interpreter lock: encode / decode them side by side (the
    # reference walks them one by one; an AVIF encode is ~30 ms per 72 x 72 tile, i.e. 250 ms for a batch of 8 on one core)
    n = host.shape[0]
    out = list(_codec_pool().map(one, range(n))) if n > 1 else [one(0)]
    return torch.stack(out, 0).to(img.device)


_CODEC_POOL = None


def _codec_pool():
    global _CODEC_POOL
    if _CODEC_POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor

        _CODEC_POOL = ThreadPoolExecutor(max_workers=max(1, min(8, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)),
                                         thread_name_prefix="otf-codec")
    return _CODEC_POOL


_OTF_DEFAULTS = dict(   # utils/redux_options.py:503-901
    p_clean=0, lens_distort_prob=0, lens_distort_strength_range=(-0.3, 0.3), rolling_shutter_strength_range=(-0.1, 0.1), chromatic_aberration_prob=0, motion_blur_prob=0, motion_blur_kernel_size=(5, 15), motion_blur_angle_range=(0, 360),
    blur_prob=0, demosaic_prob=0, sensor_noise_prob=0, sensor_noise_std_range=(0.01, 0.1), rolling_shutter_prob=0, exposure_prob=0,
    exposure_factor_range=(0.5, 2.0), color_temp_prob=0, color_temp_shift_range=(-0.2, 0.2), oversharpen_prob=0, oversharpen_strength=(1.0, 2.0),
    aliasing_prob=0, aliasing_scale_range=(0.6, 0.9), resize_mode_list3=["bilinear", "bicubic", "nearest-exact", "lanczos"],
    resize_mode_prob3=[0.25, 0.25, 0.25, 0.25], compression_formats=["jpeg", "webp", "avif", "heif"], compression_weights=[0.60, 0.25, 0.10, 0.05],
    compression_jpeg_range=(45, 95), compression_webp_range=(60, 85), compression_avif_range=(65, 90), compression_heif_range=(70, 90),
    recompression_prob=0, recompression_formats=["jpeg", "webp", "avif", "heif"], recompression_weights=[0.50, 0.35, 0.10, 0.05], editing_prob=0,
    editing_exposure_prob=0, editing_exposure_range=(0.9, 1.1), editing_oversharpen_prob=0, editing_oversharpen_strength=(1.0, 1.3), queue_size=120,
    enable_sequences=False)
_UNBUILT = ()


class OTFDegrader:
    """The training branch of RealESRGANModel.feed_data (models/realesrgan_model.py:456-664), stage for stage and RNG draw for RNG draw
    (numpy Generator = the reference's `RNG.get_rng()`, python `random` for the resize mode / crop / motion-kernel size, torch for the noise
    field and the pool shuffle), so that a seeded run takes the same decisions.  The demosaicing stage (an OpenCV round trip per sample on the
    host in the reference) runs as one device kernel restating cv2's bilinear demosaicing (cv2 is absent here: that stage's values are unpinned)."""

    def __init__(self, opt, rng: np.random.Generator | None = None, rnd=random) -> None:
        self.o = {k: (getattr(opt, k, None) if getattr(opt, k, None) is not None else v) for k, v in _OTF_DEFAULTS.items()}
        for k in _UNBUILT:
            if self.o[k]:
                raise NotImplementedError(f"OTF stage `{k}` > 0 is not built on the HIP path yet")
        if self.o["enable_sequences"]:
            # realesrgan_model.py:63-72 builds a SequenceController when this is set and never consults it (no other use in the reference):
            # the feed is the same stage list either way, so the switch is accepted and changes nothing here either.
            logging.getLogger("traiNNer").info("enable_sequences: the reference constructs its sequence controller but feed_data never uses it; "
                                               "the OTF feed runs its fixed stage list")
        seed = getattr(opt, "manual_seed", None)
        self.rng = rng if rng is not None else np.random.default_rng(seed if seed is not None else 0)
        self.rnd = rnd
        self.scale = int(opt.scale)
        ds = opt.datasets.get("train") if hasattr(opt, "datasets") else None
        self.gt_size = getattr(ds, "gt_size", None) if ds is not None else None
        if self.gt_size is None and ds is not None and getattr(ds, "lq_size", None) is not None:
            self.gt_size = ds.lq_size * self.scale                   # train.py:103-110 derives the one from the other
        self.pool = PairPool(self.o["queue_size"])
        self._pinned: dict = {}

    def _gate(self, key: str) -> bool:
        return self.rng.uniform() < self.o[key]

    @torch.no_grad()
    def degrade(self, gt: torch.Tensor, kernel1: torch.Tensor, sinc_kernel: torch.Tensor) -> torch.Tensor:
        """GT batch -> LQ batch at 1/scale (everything before the clamp/round, crop and pool)."""
        return self._after_codec_stages(self._before_codec_stages(gt, kernel1, sinc_kernel))

    @torch.no_grad()
    def _before_codec_stages(self, gt: torch.Tensor, kernel1: torch.Tensor, sinc_kernel: torch.Tensor) -> torch.Tensor:
        """Optics, sensor and ISP stages up to the final resize + sinc filter (realesrgan_model.py:505-569): device work only."""
        o, rng = self.o, self.rng
        ori_h, ori_w = gt.shape[2:4]
        out = gt
        if self._gate("lens_distort_prob"):              # one draw per gate, taken or not, like the reference
            out = warp_bilinear(out, 0, [rng.uniform(o["lens_distort_strength_range"][0], o["lens_distort_strength_range"][1])])
        if self._gate("chromatic_aberration_prob") and out.shape[1] == 3:
            out = warp_bilinear(out, 2, [1.001, 1.0, 0.999])
        if self._gate("motion_blur_prob"):
            ks = self.rnd.randint(o["motion_blur_kernel_size"][0], o["motion_blur_kernel_size"][1])
            angle = rng.uniform(o["motion_blur_angle_range"][0], o["motion_blur_angle_range"][1])
            out = conv_shared_zeropad(out, motion_blur_kernel(ks, angle), ks // 2)
        if self._gate("blur_prob"):
            out = filter2d(out, kernel1)
        if self._gate("demosaic_prob"):
            out = bayer_round_trip(out)
        if self._gate("sensor_noise_prob"):
            std = rng.uniform(o["sensor_noise_std_range"][0], o["sensor_noise_std_range"][1])
            out = gain_noise_clamp(out, 1.0, torch.randn_like(out), std)
        if self._gate("rolling_shutter_prob"):
            st = rng.uniform(o["rolling_shutter_strength_range"][0], o["rolling_shutter_strength_range"][1])
            out = warp_bilinear(out, 1, [st * out.shape[2] / out.shape[3]])
        if self._gate("exposure_prob"):
            out = gain_noise_clamp(out, rng.uniform(o["exposure_factor_range"][0], o["exposure_factor_range"][1]))
        if self._gate("color_temp_prob") and out.shape[1] == 3:
            shift = rng.uniform(o["color_temp_shift_range"][0], o["color_temp_shift_range"][1])
            gains = (1 + shift * 0.3, 1 + shift * 0.1, 1.0) if shift > 0 else (1.0, 1 - shift * 0.1, 1 - shift * 0.3)
            out = gain_noise_clamp(out, gains)
        if self._gate("oversharpen_prob"):
            s = rng.uniform(o["oversharpen_strength"][0], o["oversharpen_strength"][1])
            blurred = conv_shared_zeropad(out, torch.full((5, 5), 1.0 / 25.0), 2)
            out = axpby_clamp01(out, 1.0 + s, blurred, -s)
        if self._gate("aliasing_prob"):
            f = rng.uniform(o["aliasing_scale_range"][0], o["aliasing_scale_range"][1])
            h, w = out.shape[2:]
            out = resize_pt(resize_pt(out, "nearest", size=(int(h * f), int(w * f)), clamp=False), "nearest", size=(h, w), clamp=False)
        assert len(o["resize_mode_list3"]) == len(o["resize_mode_prob3"]), "resize_mode_list3 and resize_mode_prob3 must be the same length"
        mode = self.rnd.choices(o["resize_mode_list3"], weights=o["resize_mode_prob3"])[0]
        out = resize_pt(out, size=(ori_h // self.scale, ori_w // self.scale), mode=mode)
        return filter2d(out, sinc_kernel)

    @torch.no_grad()
    def _after_codec_stages(self, out: torch.Tensor, host: torch.Tensor | None = None) -> torch.Tensor:
        """Compression round(s) and the editing stage (realesrgan_model.py:575-609)."""
        o, rng = self.o, self.rng
        out = self._compress(out, o["compression_formats"], o["compression_weights"], host)
        if rng.uniform() < o["recompression_prob"]:
            out = self._compress(out, o["recompression_formats"], o["recompression_weights"])
        if rng.uniform() < o["editing_prob"]:
            if rng.uniform() < o["editing_exposure_prob"]:
                out = gain_noise_clamp(out, rng.uniform(o["editing_exposure_range"][0], o["editing_exposure_range"][1]))
            rng.uniform()                                # editing_oversharpen_prob: the reference draws and then does nothing (:604-609)
        return out

    def _compress(self, img, formats, weights, host=None):
        fmt = str(self.rng.choice(formats, p=weights))
        qr = self.o.get(f"compression_{fmt}_range")
        if qr is None:
            return img
        return compress_round_trip(img, fmt, self.rng.uniform(qr[0], qr[1]), host)

    def __call__(self, gt: torch.Tensor, kernel1: torch.Tensor, kernel2: torch.Tensor, sinc_kernel: torch.Tensor):
        """-> (lq, gt) training pair: p_clean pass-through or the stage list, 8-bit rounding, paired crop, pair pool."""
        return self.finish(self.start(gt, kernel1, kernel2, sinc_kernel))

    @torch.no_grad()
    def start(self, gt: torch.Tensor, kernel1: torch.Tensor, kernel2: torch.Tensor, sinc_kernel: torch.Tensor) -> dict:
        """First half of the feed: everything up to the host codec, plus an asynchronous copy of the pre-codec batch into pinned memory.
        `finish` does the rest.  Splitting here lets a caller enqueue a training step between the two halves, so that the codec's
        device -> host hop waits on an event that completed long ago instead of draining the launch queue (RNG draw order is unchanged:
        start(1) finish(1) start(2) finish(2) ... is the order of the unsplit calls)."""
        if self.gt_size is None:
            raise ValueError("datasets.train needs gt_size or lq_size for the OTF crop")
        if self.rng.uniform() < self.o["p_clean"]:
            return dict(gt=gt, clean=True)
        out = self._before_codec_stages(gt, kernel1, sinc_kernel)
        host = self._pinned.get(tuple(out.shape))          # one pinned staging buffer per shape, reused (a fresh pinned allocation per step
        if host is None:                                    # costs a host-side device synchronisation)
            host = self._pinned[tuple(out.shape)] = torch.empty(out.shape, dtype=torch.float32, pin_memory=True)
        host.copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return dict(gt=gt, clean=False, out=out, host=host, event=ev)

    @torch.no_grad()
    def finish(self, st: dict):
        gt = st["gt"]
        if st["clean"]:
            lq = gt.clone()
        else:
            st["event"].synchronize()
            lq = self._after_codec_stages(st["out"], st["host"])
        lq = clamp_round_u8(lq)
        # (the clean pass-through keeps GT resolution, so with scale > 1 the crop below raises "Scale mismatches" -- as the reference does)
        gt, lq = paired_random_crop(gt, lq, self.gt_size, self.scale, rnd=self.rnd)
        lq, gt = self.pool(lq, gt)
        return lq.contiguous(), gt
