"""Blur / sinc kernel synthesis for the OTF feed (host side, float64 numpy) -- what RealESRGANDataset draws per sample.

Same draws in the same order as /root/reference/traiNNer/data/degradations.py:215-514 (`random.choices` for the kernel family, then
`RNG.get_rng()` = a numpy Generator for sigma_x [, sigma_y, rotation] [, the beta coin and beta] [, the multiplicative noise field]), so a
seeded run produces the same kernels; pinned to the reference functions' own outputs (tests/golden/otf_degradations.pt).
The quadratic form g^T Sigma^-1 g of the rotated covariance (:22-75) is evaluated in the rotated frame instead of through a 2x2 inverse.
"""
from __future__ import annotations

import math
import random

import numpy as np

_FAMILY = {"iso": ("gauss", True), "aniso": ("gauss", False), "generalized_iso": ("generalized", True), "generalized_aniso": ("generalized", False),
           "plateau_iso": ("plateau", True), "plateau_aniso": ("plateau", False)}


def quad_form(kernel_size: int, sig_x: float, sig_y: float, theta: float) -> np.ndarray:
    """q[i, j] = g^T Sigma^-1 g, g = (x_j, y_i) on the centred grid of :40-59, Sigma = R(theta) diag(sx^2, sy^2) R(theta)^T (:22-37)."""
    ax = np.arange(-kernel_size // 2 + 1.0, kernel_size // 2 + 1.0)
    xx, yy = np.meshgrid(ax, ax)
    c, s = math.cos(theta), math.sin(theta)
    u, v = c * xx + s * yy, -s * xx + c * yy
    return (u / sig_x) ** 2 + (v / sig_y) ** 2


def random_mixed_kernels(kernel_list, kernel_prob, kernel_size=21, sigma_x_range=(0.6, 5), sigma_y_range=(0.6, 5), rotation_range=(-math.pi, math.pi),
                         betag_range=(0.5, 8), betap_range=(0.5, 8), noise_range=None, *, rng: np.random.Generator, rnd=random) -> np.ndarray:
    kind = rnd.choices(kernel_list, kernel_prob)[0]
    if kind not in _FAMILY:
        raise ValueError("Unknown kernel type")
    family, iso = _FAMILY[kind]
    assert kernel_size % 2 == 1, "Kernel size must be an odd number."
    assert sigma_x_range[0] < sigma_x_range[1], "Wrong sigma_x_range."
    sx = rng.uniform(sigma_x_range[0], sigma_x_range[1])
    if iso:
        sy, rot = sx, 0.0
    else:
        assert sigma_y_range[0] < sigma_y_range[1], "Wrong sigma_y_range."
        assert rotation_range[0] < rotation_range[1], "Wrong rotation_range."
        sy = rng.uniform(sigma_y_range[0], sigma_y_range[1])
        rot = rng.uniform(rotation_range[0], rotation_range[1])
    q = quad_form(kernel_size, sx, sy, rot)
    if family == "gauss":
        k = np.exp(-0.5 * q)
    else:
        br = betag_range if family == "generalized" else betap_range
        beta = rng.uniform(br[0], 1) if rng.uniform() < 0.5 else rng.uniform(1, br[1])
        k = np.exp(-0.5 * np.power(q, beta)) if family == "generalized" else np.reciprocal(np.power(q, beta) + 1)
    k = k / np.sum(k)
    if noise_range is not None and family != "plateau":          # random_mixed_kernels passes noise_range=None to the plateau families (:451,461)
        assert noise_range[0] < noise_range[1], "Wrong noise range."
        k = k * rng.uniform(noise_range[0], noise_range[1], size=k.shape)
    return k / np.sum(k)


def circular_lowpass_kernel(cutoff: float, kernel_size: int, pad_to: int = 0) -> np.ndarray:
    """2-D sinc (jinc) low-pass filter, :472-514."""
    from scipy import special

    assert kernel_size % 2 == 1, "Kernel size must be an odd number."
    c = (kernel_size - 1) / 2
    y, x = np.mgrid[0:kernel_size, 0:kernel_size].astype(np.float64)
    r = np.sqrt((y - c) ** 2 + (x - c) ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        k = cutoff * special.j1(cutoff * r) / (2 * np.pi * r)
    k[(kernel_size - 1) // 2, (kernel_size - 1) // 2] = cutoff ** 2 / (4 * np.pi)
    k = k / np.sum(k)
    if pad_to > kernel_size:
        p = (pad_to - kernel_size) // 2
        k = np.pad(k, ((p, p), (p, p)))
    return k
