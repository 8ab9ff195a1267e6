"""`build_optimizer` + the fused AdamW the hot path uses.

Registry contract of /root/reference/traiNNer/optimizers/__init__.py:30-47: `OPTIMIZER_REGISTRY.get(type)(params, **opt)`.
All of the reference's templates use `type: AdamW` (e.g. options/_templates/train/ESRGAN/ESRGAN_fidelity.yml:89-93).
`AdamW` here is a `torch.optim.Optimizer` (param_groups / state_dict / LR schedulers keep working) whose `step()` is ONE
kernel per parameter arena: gradient-norm clip scale, AdamW update, EMA lerp and zero_grad in a single HBM sweep.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Any

import torch

from .. import ops
from ..utils.registry import OPTIMIZER_REGISTRY

__all__ = ["AdamW", "build_optimizer"]


def _arena_of(t: torch.Tensor):
    """(flat float32 tensor over the whole storage, element offset of `t` in it) when `t` is a contiguous
    slice of a larger buffer (a parameter arena), else None."""
    if t.dtype != torch.float32 or not t.is_contiguous():
        return None
    st = t.untyped_storage()
    nfloats = st.nbytes() // 4
    if nfloats <= t.numel():
        return None
    flat = torch.empty(0, dtype=torch.float32, device=t.device).set_(st, 0, (nfloats,))
    return flat, t.storage_offset()


@OPTIMIZER_REGISTRY.register()
class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, **unused: Any):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._flat: dict = {}
        self._ema_arena = None
        self._ema_decay = 1.0
        self._clip = 0.0
        self._grad_norm = None
        self._grad_scale_world = 1
        self._inv_scale = 1.0

    def load_state_dict(self, state_dict) -> None:
        """torch.optim.AdamW-compatible per-parameter state (`step`, `exp_avg`, `exp_avg_sq`): the loaded tensors are adopted by the
        flat moment arenas at the next step()."""
        super().load_state_dict(state_dict)
        self._flat = {}

    # -- fused extras -------------------------------------------------------------------------------
    def set_step_extras(self, *, ema_arena: torch.Tensor | None = None, ema_decay: float = 1.0, clip: float = 0.0,
                        grad_norm: torch.Tensor | None = None, inv_scale: float = 1.0) -> None:
        """EMA target / decay, clip threshold, the device-resident gradient norm and the loss-scale reciprocal (GradScaler.unscale_
        folded into the sweep) for the NEXT step()."""
        self._ema_arena, self._ema_decay, self._clip, self._grad_norm, self._inv_scale = ema_arena, ema_decay, clip, grad_norm, inv_scale

    def _group_arenas(self, group):
        """Split a param group into arenas (fused) and loose tensors.  The split is cached per group under a cheap signature (count,
        storage pointers of the first / last parameter and gradient, number of absent gradients): walking 700 parameters through
        `_arena_of` cost 16 ms of host time per step() on ESRGAN -- a third of the step at the launch-bound tile sizes."""
        ps = group["params"]
        first, last = ps[0], ps[-1]
        sig = (len(ps), first.data_ptr(), last.data_ptr(), first.grad.data_ptr() if first.grad is not None else 0,
               last.grad.data_ptr() if last.grad is not None else 0, sum(1 for p in ps if p.grad is None))
        cache = self.__dict__.setdefault("_group_cache", {})
        hit = cache.get(id(group))
        if hit is not None and hit[0] == sig:
            return hit[1], list(hit[2])
        arenas, loose = self._group_arenas_uncached(group)
        cache[id(group)] = (sig, arenas, tuple(loose))
        return arenas, list(loose)

    def _group_arenas_uncached(self, group):
        arenas: dict = defaultdict(list)
        loose = []
        for p in group["params"]:
            if p.grad is None:
                continue
            a = _arena_of(p.data)
            g = _arena_of(p.grad)
            if a is None or g is None or a[1] != g[1] or a[0].numel() != g[0].numel():
                loose.append(p)
            else:
                arenas[(a[0].data_ptr(), g[0].data_ptr())].append((p, a[0], g[0]))
        return arenas, loose

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            arenas, loose = self._group_arenas(group)
            partial: list = []
            for key, members in arenas.items():
                pa, ga = members[0][1], members[0][2]
                covered = sum((p.numel() + 3) // 4 * 4 for p, _, _ in members)
                st = self._flat.get(key)
                if covered != pa.numel():
                    # a partial arena (some of the network's parameters were not given to this optimizer): per-tensor launches, then the
                    # two things the fused sweep would have done for the trainer -- zero the gradient arena and lerp the EMA arena
                    loose.extend(p for p, _, _ in members)
                    partial.append((pa, ga))
                    continue
                if st is None:
                    # ONE step tensor shared by every parameter of the arena (state_dict() still reports it per parameter): bumping a
                    # tensor per parameter per step cost ~470 tiny ops on SwinIR-M
                    st = dict(step=0, m=torch.zeros_like(pa), v=torch.zeros_like(pa), step_t=torch.tensor(0.0))
                    self._flat[key] = st
                    for p, _, _ in members:  # expose views so state_dict()/load_state_dict() see per-parameter state
                        off = p.data.storage_offset()
                        old = self.state.get(p) or {}
                        views = dict(step=st["step_t"], exp_avg=st["m"][off:off + p.numel()].view(p.shape),
                                     exp_avg_sq=st["v"][off:off + p.numel()].view(p.shape))
                        if "exp_avg" in old:   # restored by load_state_dict() (resume_training, base_model.py:794-814): move it into the arenas
                            views["exp_avg"].copy_(old["exp_avg"])
                            views["exp_avg_sq"].copy_(old["exp_avg_sq"])
                            st["step"] = int(float(old["step"]))
                            st["step_t"].fill_(float(old["step"]))
                        self.state[p] = views
                st["step"] += 1
                ema = self._ema_arena if (self._ema_arena is not None and self._ema_arena.numel() == pa.numel()) else None
                ops.adamw_ema(pa, ga, st["m"], st["v"], ema, lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"],
                              weight_decay=group["weight_decay"], step=st["step"], ema_decay=self._ema_decay, clip=self._clip,
                              grad_norm=self._grad_norm, zero_grad=True, inv_scale=self._inv_scale)
                st["step_t"] += 1
            for p in loose:
                st = self.state[p]
                if not st:
                    st.update(step=torch.tensor(0.0), exp_avg=torch.zeros_like(p.data), exp_avg_sq=torch.zeros_like(p.data))
                st["step"] += 1
                if not p.data.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("fused AdamW needs contiguous parameters and gradients")
                ops.adamw_ema(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], None, lr=group["lr"], beta1=b1, beta2=b2, eps=group["eps"],
                              weight_decay=group["weight_decay"], step=int(st["step"]), clip=self._clip, grad_norm=self._grad_norm,
                              zero_grad=False, inv_scale=self._inv_scale)
            for pa, ga in partial:
                ga.zero_()
                ema = self._ema_arena
                if ema is not None and ema.numel() == pa.numel() and pa.numel() % 4 == 0:
                    d = float(self._ema_decay)
                    ops.axpby(ops.view(ema.view(-1, 4)), d, ops.view(pa.view(-1, 4)), 1.0 - d, ops.view(ema.view(-1, 4)), pa.numel() // 4)
        return loss


def build_optimizer(params, opt: dict[str, Any]) -> torch.optim.Optimizer:
    opt = dict(opt)
    optim_type = opt.pop("type")
    return OPTIMIZER_REGISTRY.get(optim_type)(params, **opt)
