"""Pixel losses `l1loss` / `mseloss` / `charbonnierloss` on the fused HIP reduction kernel.

Same class names, constructor arguments and error behaviour as
/root/reference/traiNNer/losses/basic_loss.py:36-139 (ValueError for an unknown reduction; `loss_weight` is carried
for the trainer, which applies it: sr_model.py:641).  `forward(pred, target)` returns a 0-d tensor; the element-wise
`weight` argument and `reduction="none"` are not on the hot path and raise NotImplementedError.

Two entry points:
  * `forward` -- autograd-compatible (used when a foreign trainer calls `loss.backward()`);
  * `hip_value_and_grad` -- what SRModel uses: one kernel computes `scale * loss` into a device scalar AND
    accumulates `scale * dloss/dpred` into the generator's output-gradient buffer; nothing is saved or re-read.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn

from .. import _lib as L
from .. import ops
from ..utils.registry import LOSS_REGISTRY

_reduction_modes = ["none", "mean", "sum"]


class _PixelLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, kind, eps, sum_reduction):
        pred_c, target_c = pred.contiguous(), target.contiguous()
        if pred_c.dtype != target_c.dtype:
            target_c = target_c.to(pred_c.dtype)
        out = torch.zeros(1, dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred_c, dtype=torch.float32)
        w = float(pred_c.numel()) if sum_reduction else 1.0
        ops.pixel_loss(kind, pred_c, target_c, out, value_weight=w, eps=eps, grad=grad)
        ctx.save_for_backward(grad)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        n = grad.numel()
        if n % 4 == 0:
            flat = grad.view(n // 4, 4)
            ops.axpby(ops.view(flat), float(g), ops.NULL_VIEW, 0.0, ops.view(flat), n // 4)
            return grad, None, None, None, None
        return grad * g, None, None, None, None


class _PixelLoss(nn.Module):
    kind = L.LOSS_L1

    def __init__(self, loss_weight: float, reduction: str = "mean") -> None:
        super().__init__()
        if reduction not in _reduction_modes:
            raise ValueError(f"Unsupported reduction mode: {reduction}. Supported ones are: {_reduction_modes}")
        self.loss_weight = loss_weight
        self.reduction = reduction
        self.eps = 1e-12

    def _check(self, pred: Tensor, target: Tensor, weight):
        if weight is not None or self.reduction == "none":
            raise NotImplementedError("element-wise weights / reduction='none' are not on the HIP hot path")
        if pred.shape != target.shape:
            raise ValueError(f"shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")

    def forward(self, pred: Tensor, target: Tensor, weight: Tensor | None = None, **kwargs) -> Tensor:
        self._check(pred, target, weight)
        return _PixelLossFn.apply(pred, target, self.kind, self.eps, self.reduction == "sum")

    def hip_value_and_grad(self, pred: Tensor, target: Tensor, *, value_weight: float, value_out: Tensor, total_weight: float,
                           total_out: Tensor | None, grad_weight: float, grad: Tensor | None, first: bool) -> list[str] | None:
        """One kernel: value_out[0] = value_weight*loss ; total_out[0] += total_weight*loss ;
        grad (+)= grad_weight * dloss/dpred (`first` -> overwrite).  Returns None (scalar head)."""
        self._check(pred, target, None)
        k = float(pred.numel()) if self.reduction == "sum" else 1.0
        ops.pixel_loss(self.kind, pred, target.to(pred.dtype) if target.dtype != pred.dtype else target, value_out,
                       value_weight=value_weight * k, total_out=total_out, total_weight=total_weight * k, grad=grad,
                       grad_weight=grad_weight * k, accumulate_grad=not first, eps=self.eps)
        return None


@LOSS_REGISTRY.register()
class L1Loss(_PixelLoss):
    kind = L.LOSS_L1


@LOSS_REGISTRY.register()
class MSELoss(_PixelLoss):
    kind = L.LOSS_MSE


@LOSS_REGISTRY.register()
class CharbonnierLoss(_PixelLoss):
    kind = L.LOSS_CHARBONNIER

    def __init__(self, loss_weight: float, reduction: str = "mean", eps: float = 1e-12) -> None:
        super().__init__(loss_weight, reduction)
        self.eps = eps
