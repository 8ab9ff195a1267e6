"""Exponential moving average of the generator, call-compatible with how the reference uses `ema_pytorch.EMA`
(/root/reference/traiNNer/models/sr_model.py:226-238 ctor kwargs, :826-828 update, :1182-1205 save of `.ema_model`).

ema-pytorch is not importable here; its published update rule is restated (values "parity unpinned", see
oracle/step.py): `step` starts at 0; update(): s = step; step += 1; s <= update_after_step -> copy the online weights;
first update after that -> copy once more (`initted`); else decay = clamp(1 - (1 + step - update_after_step - 1) ** -power, 0, beta), ema = lerp(ema, online, 1 - decay).
When both models keep their parameters in flat arenas the lerp is folded into the optimizer sweep
(optimizers.AdamW.set_step_extras); `update()` is the stand-alone form (one tr_axpby over the arena).
"""
from __future__ import annotations

import copy

import torch
from torch import nn

from .. import ops


class EMA(nn.Module):
    def __init__(self, model: nn.Module, ema_model: nn.Module | None = None, beta: float = 0.9999, update_after_step: int = 100,
                 update_every: int = 10, inv_gamma: float = 1.0, power: float = 2 / 3, min_value: float = 0.0,
                 allow_different_devices: bool = False, update_model_with_ema_every: int | None = None, **unused) -> None:
        super().__init__()
        self.beta, self.update_after_step, self.update_every = beta, update_after_step, update_every
        self.inv_gamma, self.power, self.min_value = inv_gamma, power, min_value
        self.update_model_with_ema_every = update_model_with_ema_every
        self.online_model = [model]  # not registered: the online net is owned by the trainer
        self.ema_model = ema_model if ema_model is not None else copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.register_buffer("initted", torch.tensor(False))
        self.register_buffer("step", torch.tensor(0))

    @property
    def model(self) -> nn.Module:
        return self.online_model[0]

    def forward(self, *a, **k):
        return self.ema_model(*a, **k)

    def get_current_decay(self) -> float:
        epoch = max(int(self.step) - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.0
        value = 1 - (1 + epoch / self.inv_gamma) ** -self.power
        return min(max(value, self.min_value), self.beta)

    def advance(self) -> float | None:
        """Advance the step counter and return the decay the caller must apply this step
        (0.0 = copy the online weights, None = no update this step)."""
        s = int(self.step)
        self.step += 1
        n = self.update_model_with_ema_every
        self._switch_due = bool(n) and s % int(n) == 0
        if s % self.update_every != 0:
            return None
        if s <= self.update_after_step:
            return 0.0
        if not bool(self.initted):
            self.initted.fill_(True)
            return 0.0  # copy followed by a lerp of identical tensors == copy
        return self.get_current_decay()

    def _arenas(self):
        on, em = self.model, self.ema_model
        if hasattr(on, "flat_parameters") and hasattr(em, "flat_parameters"):
            return on.flat_parameters()[0], em.flat_parameters()[0]
        return None

    def ema_arena(self) -> torch.Tensor | None:
        ar = self._arenas()
        return None if ar is None else ar[1]

    def _lerp(self, decay: float) -> None:
        ar = self._arenas()
        if ar is not None and ar[0].numel() == ar[1].numel() and ar[0].numel() % 4 == 0:
            p, e = ar[0].view(-1, 4), ar[1].view(-1, 4)
            ops.axpby(ops.view(e), decay, ops.view(p), 1.0 - decay, ops.view(e), p.shape[0])
            return
        with torch.no_grad():
            for pe, po in zip(self.ema_model.parameters(), self.model.parameters()):
                pe.lerp_(po.to(pe.device), 1.0 - decay)
            for be, bo in zip(self.ema_model.buffers(), self.model.buffers()):
                be.copy_(bo)

    @torch.no_grad()
    def maybe_update_model_with_ema(self) -> None:
        """`update_model_with_ema_every` (the trainer's `ema_switch_iter`, sr_model.py:223-234): at the end of every N-th update() ema-pytorch
        copies the EMA weights back into the online model (`update_model_with_ema_beta` 0 = plain copy, the only value the reference
        uses).  Call after the lerp of the step `advance()` announced it for."""
        if not getattr(self, "_switch_due", False):
            return
        self._switch_due = False
        ar = self._arenas()
        if ar is not None and ar[0].numel() == ar[1].numel():
            ar[0].copy_(ar[1])
            return
        for pe, po in zip(self.ema_model.parameters(), self.model.parameters()):
            po.data.copy_(pe.data.to(po.device))
        for be, bo in zip(self.ema_model.buffers(), self.model.buffers()):
            bo.copy_(be.to(bo.device))

    @torch.no_grad()
    def update(self) -> None:
        d = self.advance()
        if d is not None:
            self._lerp(d)
        self.maybe_update_model_with_ema()
