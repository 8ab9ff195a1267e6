"""Oracle restatement of UNetDiscriminatorSN (test infrastructure, see oracle/__init__.py).

Follows /root/reference/traiNNer/archs/discriminator_arch.py:98-160 and torch.nn.utils.spectral_norm (old-style hook,
one power iteration per training-mode forward, eps 1e-12, u/v updated in place).  PINNED: tests/golden/unet_discriminator_sn.pt
holds the reference module's own output, updated u/v, input- and parameter-gradients (tests/golden/make_golden.py), and
tests/test_oracle_cpu.py checks this file against them.
"""
from __future__ import annotations

import torch
from torch.nn import functional as F

from .precision import FP32, Policy

SLOPE = 0.2
SN_LAYERS = ("conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv8")


def spectral_weight(w_orig, u, v, training: bool, eps: float = 1e-12):
    """-> (W / sigma, u', v'); u', v' are the post-iteration vectors (detached), gradient flows through W and sigma only."""
    wm = w_orig.reshape(w_orig.shape[0], -1)
    if training:
        with torch.no_grad():
            v = F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps)
            u = F.normalize(torch.mv(wm, v), dim=0, eps=eps)
    sigma = torch.dot(u, torch.mv(wm, v))
    return w_orig / sigma, u, v


def forward(sd: dict, x: torch.Tensor, training: bool = True, skip_connection: bool = True, policy: Policy = FP32):
    """sd: reference-style state dict (convN.weight_orig / weight_u / weight_v, conv0/conv9 weight+bias).
    Returns (logits N x 1 x H x W, {name: (u', v')})."""
    new_uv = {}

    def sn(name):
        w, u, v = spectral_weight(sd[f"{name}.weight_orig"], sd[f"{name}.weight_u"], sd[f"{name}.weight_v"], training)
        new_uv[name] = (u, v)
        return w

    lrelu = lambda t: F.leaky_relu(t, SLOPE)  # noqa: E731
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)  # noqa: E731
    x0 = lrelu(policy.conv2d(x, sd["conv0.weight"], sd["conv0.bias"], 1, 1))
    x1 = lrelu(policy.conv2d(x0, sn("conv1"), None, 2, 1))
    x2 = lrelu(policy.conv2d(x1, sn("conv2"), None, 2, 1))
    x3 = lrelu(policy.conv2d(x2, sn("conv3"), None, 2, 1))
    x4 = lrelu(policy.conv2d(up(x3), sn("conv4"), None, 1, 1))
    if skip_connection:
        x4 = x4 + x2
    x5 = lrelu(policy.conv2d(up(x4), sn("conv5"), None, 1, 1))
    if skip_connection:
        x5 = x5 + x1
    x6 = lrelu(policy.conv2d(up(x5), sn("conv6"), None, 1, 1))
    if skip_connection:
        x6 = x6 + x0
    out = lrelu(policy.conv2d(x6, sn("conv7"), None, 1, 1))
    out = lrelu(policy.conv2d(out, sn("conv8"), None, 1, 1))
    out = policy.conv2d(out, sd["conv9.weight"], sd["conv9.bias"], 1, 1)
    return out, new_uv
