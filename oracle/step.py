"""Oracle restatement of one SRModel.optimize_parameters micro-step (test infrastructure).

Follows /root/reference/traiNNer/models/sr_model.py:452-828 for the generator-only branch that
BASELINE.json configs 1-2 exercise (no net_d):
  :478      output = net_g(lq)
  :500-626  every loss gets (output, gt)
  :639-663  scalar loss:  l_g_total += loss*|w| / accum_iters ; dict loss: += v*|w| * accum_iters (sic)
  :665      non-finite -> RuntimeError("Training failed: NaN/Inf found in loss.")
  :681      backward
  :696-705  grad_norm_g = || [ ||p.grad||_2 for p ] ||_2
  :707-711  optional clip_grad_norm_(threshold)  (default threshold 1.0)
  :713-719  optimizer step, zero_grad
  :810-816  log_dict[k] += value * n_samples
  :826-828  EMA update
Optimizer: torch.optim.AdamW semantics (lr, betas, eps 1e-8, decoupled weight decay), the
reference's registered `AdamW` (/root/reference/traiNNer/optimizers/__init__.py:30-47).
EMA: ema-pytorch (absent from this container; formula restated from its published source, values
*parity unpinned*): step counter starts at 0; update(): s = step; step += 1; if s <= update_after_step:
copy params; on the first later update copy once more (`initted`), then decay = clamp(1 - (1 + (step - update_after_step - 1))**(-power), 0, beta),
ema = lerp(ema, p, 1 - decay).  ctor kwargs at sr_model.py:226-235.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch


class AdamWState:
    def __init__(self, params: "OrderedDict[str, torch.Tensor]", lr=2e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.0):
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.t = 0
        self.m = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())
        self.v = OrderedDict((k, torch.zeros_like(v)) for k, v in params.items())

    @torch.no_grad()
    def step(self, params, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        for k, p in params.items():
            g = grads[k]
            p.mul_(1 - self.lr * self.wd)
            self.m[k].lerp_(g, 1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


class EMAState:
    def __init__(self, params, beta=0.999, power=10.0, update_after_step=0):
        self.beta, self.power, self.uas = beta, power, update_after_step
        self.step = 0
        self.initted = False
        self.shadow = OrderedDict((k, v.detach().clone()) for k, v in params.items())

    def current_decay(self) -> float:
        epoch = max(self.step - self.uas - 1, 0)
        if epoch <= 0:
            return 0.0
        return min(max(1 - (1 + epoch) ** (-self.power), 0.0), self.beta)

    @torch.no_grad()
    def update(self, params):
        s = self.step
        self.step += 1
        if s <= self.uas or not self.initted:
            for k, v in params.items():
                self.shadow[k].copy_(v)
            if s <= self.uas:
                return
            self.initted = True
        d = self.current_decay()
        for k, v in params.items():
            self.shadow[k].lerp_(v, 1 - d)


def grad_norm(grads) -> torch.Tensor:
    return torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2) for g in grads.values()]))


@torch.no_grad()
def clip_grads_(grads, max_norm: float):
    total = grad_norm(grads)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def generator_step(params, forward_fn, loss_fns, lq, gt, opt: AdamWState, ema: EMAState | None,
                   accum_iters: int = 1, apply_gradient: bool = True, grad_clip: float | None = None):
    """One micro-step.  `params`: name -> leaf tensor (requires_grad);  `forward_fn(params, lq)`;
    `loss_fns`: label -> (callable(output, gt) -> scalar | dict, loss_weight).
    Returns (output, loss_dict) with the reference's log keys."""
    n = gt.shape[0]
    out = forward_fn(params, lq)
    total = torch.zeros((), dtype=torch.float32)
    loss_dict = OrderedDict()
    for label, (fn, w) in loss_fns.items():
        val = fn(out, gt)
        aw = abs(w)
        if isinstance(val, dict):
            for sub, v in val.items():
                if v > 0:
                    wv = v * aw
                    total = total + wv * accum_iters
                    loss_dict[f"{label}_{sub}"] = wv
        else:
            wv = val * aw
            total = total + wv / accum_iters
            loss_dict[label] = wv
    if not torch.isfinite(total):
        raise RuntimeError("Training failed: NaN/Inf found in loss.")
    loss_dict["l_g_total"] = total
    names = list(params.keys())
    gs = torch.autograd.grad(total, [params[k] for k in names])
    grads = OrderedDict()
    for k, g in zip(names, gs):
        prev = getattr(params[k], "_acc_grad", None)
        grads[k] = g if prev is None else prev + g
    if apply_gradient:
        loss_dict["grad_norm_g"] = grad_norm(grads)
        if grad_clip:
            clip_grads_(grads, grad_clip)
            loss_dict["grad_clip_threshold"] = grad_clip
        opt.step(OrderedDict((k, params[k].data) for k in names), grads)
        for k in names:
            params[k]._acc_grad = None
        if ema is not None:
            ema.update(OrderedDict((k, params[k].data) for k in names))
    else:
        for k in names:
            params[k]._acc_grad = grads[k]
    log = OrderedDict((k, (v.detach().float() if torch.is_tensor(v) else v) * n) for k, v in loss_dict.items())
    return out.detach(), loss_dict, log, grads
