"""Iteration-scheduled loss weights: the `IterativeLossWrapper` surface of the reference
(/root/reference/traiNNer/losses/iterative_loss_wrapper.py:26-289) in front of the HIP loss heads.

The reference wraps a loss module and multiplies its VALUE by a weight that depends on the training iteration; `SRModel` then multiplies by
`abs(loss.loss_weight)` (the configured base weight) a second time (sr_model.py:623-663).  On the HIP path a loss head writes value, total and
d(loss)/d(output) in one launch with host-side scalar weights, so nothing has to be wrapped around a tensor: `SRModel.optimize_parameters`
asks `get_current_weight(current_iter)` for the factor and hands `loss_weight * factor` to the head (`loss_module.hip_value_and_grad`).
`forward(..., current_iter=)` keeps the reference's call form for foreign (autograd) heads and for callers outside the trainer.

Schedule (reference :87-150): 0 before `start_iter` and from `disable_after` on; `target_weight` when there is no ramp
(`target_iter <= start_iter`); otherwise base -> target over [start_iter, target_iter] by `linear`, `cosine` (half-cosine easing) or `step`
(jump at `target_iter`); an unknown `schedule_type` warns and ramps linearly.
"""
from __future__ import annotations

import math
import warnings
from typing import Any

import torch
from torch import nn


class IterativeLossWrapper(nn.Module):
    def __init__(self, loss_module: nn.Module, loss_weight: float = 1.0, start_iter: int = 0, target_iter: int | None = None,
                 target_weight: float | None = None, disable_after: int | None = None, schedule_type: str = "linear",
                 warn_on_unused: bool = True) -> None:
        super().__init__()
        self.loss_module = loss_module
        self.base_loss_weight = loss_weight
        self.loss_weight = loss_weight             # what SRModel multiplies by on top of the schedule factor
        self.start_iter = start_iter
        self.target_iter = start_iter if target_iter is None else target_iter
        self.target_weight = loss_weight if target_weight is None else target_weight
        self.disable_after = disable_after
        self.schedule_type = schedule_type.lower()
        self.warn_on_unused = warn_on_unused
        if self.target_iter < self.start_iter:
            raise ValueError(f"target_iter ({self.target_iter}) must be >= start_iter ({self.start_iter})")
        if self.target_weight < 0:
            raise ValueError(f"target_weight ({self.target_weight}) must be non-negative")
        self._last_effective_weight = 0.0
        self._has_warned_unused = False

    # attributes of the wrapped head that SRModel / the heads' own plumbing look at
    @property
    def returns_dict(self) -> bool:
        return bool(getattr(self.loss_module, "returns_dict", False))

    def set_compute_dtype(self, dtype) -> None:
        if hasattr(self.loss_module, "set_compute_dtype"):
            self.loss_module.set_compute_dtype(dtype)

    def set_fast_matmul(self, on: bool) -> None:
        if hasattr(self.loss_module, "set_fast_matmul"):
            self.loss_module.set_fast_matmul(on)

    def _calculate_schedule_weight(self, current_iter: int) -> float:
        if self.disable_after is not None and current_iter >= self.disable_after:
            return 0.0
        if current_iter < self.start_iter:
            return 0.0
        if self.target_iter <= self.start_iter:
            return self.target_weight
        t = min(1.0, max(0.0, (current_iter - self.start_iter) / (self.target_iter - self.start_iter)))
        span = self.target_weight - self.base_loss_weight
        if self.schedule_type == "cosine":
            return self.base_loss_weight + span * (0.5 * (1 - math.cos(math.pi * t)))
        if self.schedule_type == "step":
            return self.target_weight if current_iter >= self.target_iter else self.base_loss_weight
        if self.schedule_type != "linear":
            warnings.warn(f"Unknown schedule_type '{self.schedule_type}', using linear", stacklevel=2)
        return self.base_loss_weight + span * t

    def get_current_weight(self, current_iter: int) -> float:
        return self._calculate_schedule_weight(current_iter)

    def step_weight(self, current_iter: int) -> float:
        """`get_current_weight` with the bookkeeping of the reference's forward (last weight, the one-time 'unused' warning, :166-183)."""
        w = self._calculate_schedule_weight(current_iter)
        self._last_effective_weight = w
        if self.warn_on_unused and w < 1e-6 and not self._has_warned_unused:
            warnings.warn(f"Loss {self.loss_module.__class__.__name__} has very small effective weight ({w:.2e}) at iteration {current_iter}. "
                          f"This may indicate configuration issues.", UserWarning, stacklevel=3)
            self._has_warned_unused = True
        return w

    def is_active(self, current_iter: int) -> bool:
        return self._calculate_schedule_weight(current_iter) > 1e-6

    def get_schedule_info(self) -> dict[str, Any]:
        return {"base_loss_weight": self.base_loss_weight, "start_iter": self.start_iter, "target_iter": self.target_iter,
                "target_weight": self.target_weight, "disable_after": self.disable_after, "schedule_type": self.schedule_type}

    def forward(self, *args, current_iter: int | None = None, **kwargs) -> Any:
        """The reference's tensor form (:152-223): value(s) of the wrapped module times the factor of this iteration."""
        if current_iter is None:
            raise ValueError("current_iter must be provided to IterativeLossWrapper")
        w = self.step_weight(current_iter)
        base = self.loss_module(*args, **kwargs)
        if isinstance(base, dict):
            return {k: v * w for k, v in base.items()}
        if w == 0.0:
            return torch.zeros((), device=base.device, dtype=base.dtype) if isinstance(base, torch.Tensor) else base * 0.0
        return base * w

    def __repr__(self) -> str:
        return f"IterativeLossWrapper({self.loss_module.__class__.__name__}, {self.get_schedule_info()})"


def create_iterative_loss(loss_module: nn.Module, loss_config: dict[str, Any]) -> IterativeLossWrapper:
    """Reference :248-289: the schedule keys of a loss option block, defaults start_iter 0 / linear / warn."""
    keys = ("start_iter", "target_iter", "target_weight", "disable_after", "schedule_type", "warn_on_unused")
    kw = {k: loss_config[k] for k in keys if k in loss_config}
    kw.setdefault("start_iter", 0)
    kw.setdefault("schedule_type", "linear")
    kw.setdefault("warn_on_unused", True)
    return IterativeLossWrapper(loss_module=loss_module, loss_weight=loss_config["loss_weight"], **kw)
