"""Flat parameter / gradient arenas shared by the HIP generators and discriminators.

fp32 master weights of a network live in ONE contiguous buffer and their gradients in a second one of the same shape;
`nn.Parameter`s are views.  The optimizer sweep (`tr_adamw_ema`), the EMA lerp and the data-parallel all-reduce then touch
one buffer each, while `state_dict()` keeps the reference's key layout.  Survives `.to()` / `channels_last`
(base_model.py:202-244 does both), `optimizer.zero_grad(set_to_none=True)` and `copy.deepcopy` (the EMA twin).
"""
from __future__ import annotations

import copy

import torch
from torch import nn

_TRANSIENT = ("_ws", "_saved", "_plan", "_pk", "_trigger", "_arena", "_grad_arena", "_offsets", "_slabs")


class FlatParamModule(nn.Module):
    def _init_arena_state(self) -> None:
        self._arena = None
        self._grad_arena = None
        self._plan = None
        self._ws: dict = {}
        self._saved = None
        self._trigger = None

    def _flatten(self) -> None:
        ps = list(self.parameters())
        dev = ps[0].device
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        arena = torch.zeros(total, dtype=torch.float32, device=dev)
        garena = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(ps, offs):
            if p.dtype != torch.float32:
                raise TypeError("master weights stay float32; select bf16 compute with `compute_dtype`")
            arena[o:o + p.numel()].copy_(p.data.reshape(-1))
            if p.grad is not None:
                garena[o:o + p.numel()].copy_(p.grad.reshape(-1))
            p.data = arena[o:o + p.numel()].view(p.shape)
            p.grad = garena[o:o + p.numel()].view(p.shape) if p.requires_grad else None
        self._arena, self._grad_arena, self._offsets = arena, garena, offs
        self._plan = None
        self._ws = {}

    def __deepcopy__(self, memo):
        # an EMA twin must not inherit workspaces, saved activations or slabs that alias the online net
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in _TRANSIENT:
                continue
            new.__dict__[k] = copy.deepcopy(v, memo)
        new._ws, new._saved, new._plan, new._trigger, new._arena, new._grad_arena = {}, None, None, None, None, None
        return new

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._arena = None  # storage moved: rebuild lazily
        return out

    def _ensure_arena(self) -> None:
        ps = list(self.parameters())
        if self._arena is None or ps[0].data_ptr() != self._arena.data_ptr() or ps[0].device != self._arena.device:
            self._flatten()
        elif ps[0].requires_grad and (ps[0].grad is None or ps[-1].grad is None or ps[0].grad.data_ptr() != self._grad_arena.data_ptr()):
            # optimizer.zero_grad(set_to_none=True) dropped the views: re-attach, zeroed
            self._grad_arena.zero_()
            for p, o in zip(ps, self._offsets):
                p.grad = self._grad_arena[o:o + p.numel()].view(p.shape)

    def flat_parameters(self) -> tuple[torch.Tensor, torch.Tensor]:
        """(params arena, grads arena) -- consumed by the fused AdamW/EMA sweep and the gradient all-reduce."""
        self._ensure_arena()
        return self._arena, self._grad_arena

    def _pool(self, phase: str, dev, cdt):
        """Activation / gradient buffers are drawn in a fixed order every step, so the n-th request of a phase reuses the n-th
        buffer of the previous step (no allocator traffic, no per-step memset).  Saved activations therefore belong to the LATEST
        training forward: run its backward before the next forward of this module."""
        pool = self._ws.setdefault(phase, [])
        idx = [0]

        def take(*shape, dt=cdt):
            i = idx[0]
            idx[0] += 1
            if i < len(pool):
                t = pool[i]
                if t.shape == tuple(shape) and t.dtype == dt and t.device == dev:
                    return t
                del pool[i:]                        # geometry changed (another tile size / batch): rebuild from here on
            t = torch.zeros(*shape, dtype=dt, device=dev)
            pool.append(t)
            return t

        return take
