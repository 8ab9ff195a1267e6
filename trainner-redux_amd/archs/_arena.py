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

_TRANSIENT = ("_keep_state", "_ws", "_saved", "_plan", "_pk", "_trigger", "_arena", "_grad_arena", "_offsets", "_slabs", "_gs", "_plist", "_frozen_arena", "_offset_of", "grad_exchange")


class FlatParamModule(nn.Module):
    # opt-in (the trainer maps the reference's `use_compile` switch onto it, redux_options.py `use_compile`; env TRAINNER_HIP_GRAPHS=1 forces it):
    # training steps replay forward_hip / backward_hip as two hipGraphs instead of enqueueing ~700 launches each from Python
    hip_graphs = False
    GRAPH_WARMUP = 2          # eager steps before capture: they size the buffer pools and the per-stream scratch

    grad_exchange = None      # utils.dist_util.GradExchange of the gradient arena (set by the trainer when world_size > 1): backward_hip reports
                              # finished parameter ranges to it through `_grads_final_from`, the exchange then overlaps the rest of backward

    def _grads_final_from(self, param) -> None:
        """All gradients of `param` and of every parameter registered after it are final (their last accumulating launch is enqueued)."""
        ex = self.grad_exchange
        if ex is None or not ex.active or (ex.flat.is_cuda and torch.cuda.is_current_stream_capturing()):     # (a hipGraph capture records kernels only: the exchange
            return                                                                       #  then runs as a whole after the replay, GradExchange.finish)
        offs = self.__dict__.get("_offset_of")
        if offs is None or offs.get("_arena_ptr") != self._grad_arena.data_ptr():
            offs = self._offset_of = {id(p): o for p, o in zip(self._plist, self._offsets)}
            offs["_arena_ptr"] = self._grad_arena.data_ptr()
        ex.ready_from(offs[id(param)])

    def _init_arena_state(self) -> None:
        self._gs = None
        self._pool_epoch = 0
        self._arena = None
        self._grad_arena = None
        self._plan = None
        self._ws: dict = {}
        self._saved = None
        self._trigger = None

    def _frozen_param_names(self) -> set:
        """Names of parameters that never train (SPAN's exported `eval_conv.*` kernels).  They live in a side buffer so that the trainable
        arena holds exactly what the optimizer is given: the fused sweep (`tr_adamw_ema`: AdamW + EMA lerp + zero_grad) needs ONE fully
        covered buffer.  Declared by name, not read from `requires_grad`, so that the EMA twin (every parameter frozen) keeps the layout of
        the online network."""
        return set()

    def _flatten(self) -> None:
        named = list(self.named_parameters())
        frozen_names = self._frozen_param_names()
        train = [p for n, p in named if n not in frozen_names]
        frozen = [p for n, p in named if n in frozen_names]
        ps = self._plist = train + frozen
        dev = ps[0].device
        offs, total = [], 0
        for p in train:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        arena = torch.zeros(total, dtype=torch.float32, device=dev)
        garena = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, o in zip(train, offs):
            if p.dtype != torch.float32:
                raise TypeError("master weights stay float32; select bf16 compute with `compute_dtype`")
            arena[o:o + p.numel()].copy_(p.data.reshape(-1))
            if p.grad is not None:
                garena[o:o + p.numel()].copy_(p.grad.reshape(-1))
            p.data = arena[o:o + p.numel()].view(p.shape)
            p.grad = garena[o:o + p.numel()].view(p.shape) if p.requires_grad else None
        if frozen:
            fo, ft = [], 0
            for p in frozen:
                fo.append(ft)
                ft += (p.numel() + 3) // 4 * 4
            side = torch.zeros(ft, dtype=torch.float32, device=dev)
            for p, o in zip(frozen, fo):
                side[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = side[o:o + p.numel()].view(p.shape)
                p.grad = None
            self._frozen_arena = side
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
        new._gs, new._pool_epoch = None, 0
        return new

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._arena = None  # storage moved: rebuild lazily
        return out

    def _ensure_arena(self) -> None:
        # the parameter list is walked once per flatten: `list(self.parameters())` on every call was 8 ms of host time per ESRGAN step
        ps = self.__dict__.get("_plist") if self._arena is not None else None
        if ps is None:
            ps = list(self.parameters())
        if self._arena is None or ps[0].data_ptr() != self._arena.data_ptr() or ps[0].device != self._arena.device:
            self._flatten()
        elif ps[0].requires_grad and (ps[0].grad is None or ps[len(self._offsets) - 1].grad is None or ps[0].grad.data_ptr() != self._grad_arena.data_ptr()):
            # optimizer.zero_grad(set_to_none=True) dropped the views: re-attach, zeroed
            self._grad_arena.zero_()
            for p, o in zip(ps, self._offsets):     # the trainable prefix of _plist
                p.grad = self._grad_arena[o:o + p.numel()].view(p.shape) if p.requires_grad else None

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
            self._pool_epoch = getattr(self, "_pool_epoch", 0) + 1     # captured graphs hold raw pointers into the pools: a new buffer retires them
            return t

        return take

    # ------------------------------------------------------------------ hipGraph replay of the training step ---------------------------
    # Every operand of forward_hip / backward_hip lives at a fixed address after the first step (pooled activations, weight slabs, the
    # parameter and gradient arenas), every entry point of the C ABI is capturable (no allocation, no host sync), so a training step of a
    # network is two graph launches.  Measured on MI355X (tools/try_graph.py, ESRGAN 23 blocks, batch 8, 64 -> 256): host time per
    # forward + backward 22.4 ms -> 0.4 ms, outputs and gradients bit-identical to the eager launches.
    # A graph is retired when anything it points at may have moved: another input shape / compute dtype, a re-flattened arena (.to()), a
    # pool buffer allocated since (validation at another tile size), or per-step operands that are not static (DropPath keep masks).
    def _graphs_on(self) -> bool:
        import os

        # stochastic depth: only networks whose keep masks are drawn on the device into fixed buffers inside forward_hip (SwinIR: the draw is
        # captured with the step and torch advances the generator's offset on every replay) may replay; host-drawn masks are per-step operands
        dp = max(float(getattr(self, "drop_path", 0.0) or 0.0), float(getattr(self, "drop_path_rate", 0.0) or 0.0))
        return bool(self.hip_graphs or os.environ.get("TRAINNER_HIP_GRAPHS") == "1") and (dp == 0.0 or getattr(self, "graph_safe_drop_path", False))

    def train_forward(self, x: torch.Tensor) -> torch.Tensor:
        """forward_hip(x, train=True), replayed from a hipGraph once the step has run GRAPH_WARMUP times with this geometry."""
        if not self._graphs_on():
            self._gs = None
            return self.forward_hip(x, train=True)
        self._ensure_arena()
        key = (tuple(x.shape), x.dtype, self.compute_dtype, self._arena.data_ptr())
        gs = self._gs
        if gs is None or gs["key"] != key or (gs.get("gf") is not None and gs["epoch"] != self._pool_epoch):
            gs = self._gs = dict(key=key, calls=0, gf=None, gb=None)
        gs["calls"] += 1
        if gs["calls"] <= self.GRAPH_WARMUP or x.dtype != torch.float32 or not x.is_contiguous():
            gs["mode"] = "eager"
            return self.forward_hip(x, train=True)
        if gs["gf"] is None:
            gs["x"] = x.clone()
            gf = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gf):
                gs["y"] = self.forward_hip(gs["x"], train=True)
            gs["gf"], gs["saved"], gs["epoch"] = gf, self._saved, self._pool_epoch
            gs["keep"] = list(self._ws.values())       # the graph's operands (pool lists / workspace objects) stay allocated as long as it lives
        else:
            gs["x"].copy_(x)
        gs["gf"].replay()
        self._saved = gs["saved"]
        gs["mode"] = "graph"
        return gs["y"]

    def train_backward(self, grad_out: torch.Tensor) -> None:
        gs = self._gs
        if gs is None or gs.get("mode") != "graph":
            return self.backward_hip(grad_out)
        if self._saved is None:
            raise RuntimeError("train_backward needs a preceding train_forward")
        grad_out = grad_out.contiguous().float()
        if gs["gb"] is None:
            gs["gy"] = grad_out.clone()
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=gs["gf"].pool()):
                self.backward_hip(gs["gy"])
            if self._pool_epoch != gs["epoch"]:       # the capture itself had to allocate (backward pools not warm): retire and fall back
                self._saved = gs["saved"]
                self._gs = None
                return self.backward_hip(grad_out)
            gs["gb"] = gb
        else:
            gs["gy"].copy_(grad_out)
        gs["gb"].replay()
        self._saved = None
