"""`build_loss` -- registry lookup with the reference's call convention
(/root/reference/traiNNer/losses/__init__.py:42-119): every `*_loss.py` here is imported at package import;
`build_loss({"type": T, "loss_weight": w, **kw})` -> `LOSS_REGISTRY.get(T)(loss_weight=w, **kw)`.
Iteration-schedule keys (`start_iter`, `target_iter`, ... :79-110) are split off and the loss is wrapped in `IterativeLossWrapper`
(iterative_loss_wrapper.py), whose factor SRModel folds into the host-side weights of the HIP head.
"""
from __future__ import annotations

import importlib
import os
from copy import deepcopy
from typing import Any

from torch import nn

from ..utils.registry import LOSS_REGISTRY

__all__ = ["build_loss"]

_here = os.path.dirname(os.path.abspath(__file__))
_loss_modules = [
    importlib.import_module(f"{__name__}.{os.path.splitext(f)[0]}") for f in sorted(os.listdir(_here)) if f.endswith("_loss.py")
]

_SCHEDULE_KEYS = ("start_iter", "target_iter", "target_weight", "disable_after", "schedule_type", "warn_on_unused", "loss_decay",
                  "loss_decay_inflection")


def build_loss(loss_opt: dict[str, Any]) -> nn.Module:
    opt = deepcopy(loss_opt)
    loss_type = opt.pop("type")
    sched = {k: opt.pop(k) for k in _SCHEDULE_KEYS if k in opt}      # never reach the loss constructor (:92-96)
    loss = LOSS_REGISTRY.get(loss_type)(**opt)
    if sched:
        from .iterative_loss_wrapper import create_iterative_loss

        return create_iterative_loss(loss, {**loss_opt, **sched})
    return loss
