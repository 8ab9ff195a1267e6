"""Oracle restatement of the loss heads on the hot path (test infrastructure, see oracle/__init__.py).

Follows:
  * l1 / mse / charbonnier + reductions  /root/reference/traiNNer/losses/basic_loss.py:16-28,
                                         /root/reference/traiNNer/losses/loss_util.py:8-59
  * GANLoss vanilla / lsgan / hinge / wgan(_softplus)
                                         /root/reference/traiNNer/losses/gan_loss.py:13-126
These two reference files load by path in this container; tests/golden/pixel_gan_losses.pt holds
their outputs (tests/golden/make_golden.py) and tests/test_oracle_golden.py pins this file to them.
"""
from __future__ import annotations

import torch
from torch.nn import functional as F


def _reduce(t, reduction):
    if reduction == "none":
        return t
    if reduction == "mean":
        return t.mean()
    if reduction == "sum":
        return t.sum()
    raise ValueError(f"Unsupported reduction mode: {reduction}")


def _weighted(elem, weight, reduction):
    # loss_util.py:28-59
    if weight is not None:
        assert weight.dim() == elem.dim()
        assert weight.size(1) == 1 or weight.size(1) == elem.size(1)
        elem = elem * weight
    if weight is None or reduction == "sum":
        return _reduce(elem, reduction)
    if reduction == "mean":
        denom = weight.sum() if weight.size(1) > 1 else weight.sum() * elem.size(1)
        return elem.sum() / denom
    return elem


def l1(pred, target, weight=None, reduction="mean"):
    return _weighted((pred - target).abs(), weight, reduction)


def mse(pred, target, weight=None, reduction="mean"):
    return _weighted((pred - target) ** 2, weight, reduction)


def charbonnier(pred, target, weight=None, reduction="mean", eps=1e-12):
    return _weighted(torch.sqrt((pred - target) ** 2 + eps), weight, reduction)


def gan(pred, target_is_real: bool, is_disc: bool = False, gan_type: str = "vanilla",
        real_label_val: float = 1.0, fake_label_val: float = 0.0, loss_weight: float = 1.0):
    """gan_loss.py:99-126.  Note: the class returns the *unweighted* value for D and for G alike in
    this fork (`return loss` -- the trainer applies loss_weight, sr_model.py:641,763-768)."""
    if gan_type == "hinge":
        if is_disc:
            x = -pred if target_is_real else pred
            return F.relu(1 + x).mean()
        return -pred.mean()
    if gan_type in ("wgan", "wgan_softplus"):
        if gan_type == "wgan":
            return -pred.mean() if target_is_real else pred.mean()
        return F.softplus(-pred).mean() if target_is_real else F.softplus(pred).mean()
    label = real_label_val if target_is_real else fake_label_val
    tgt = pred.new_ones(pred.size()) * label
    if gan_type == "vanilla":
        return F.binary_cross_entropy_with_logits(pred, tgt)
    if gan_type == "lsgan":
        return F.mse_loss(pred, tgt)
    raise NotImplementedError(gan_type)
