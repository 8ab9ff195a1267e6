"""Precision policies for the oracle (test infrastructure, see oracle/__init__.py).

`fp32`      : the reference CPU path as its own test runs it (use_amp False,
              /root/reference/tests/test_models/test_sr_model.py:31-32).
`bf16_ops` / `f16_ops` : same op order, but every convolution rounds its two MFMA operands (activation and
              weight; in backward also the incoming gradient) to bfloat16 and accumulates in fp32,
              which is the arithmetic contract of the HIP bf16 path (DESIGN.md section "numerics").
              Everything outside the contraction (bias, LeakyReLU, residual adds, loss, AdamW)
              stays fp32.  The same holds for the attention matmuls (q k^T and P v): operands bf16, scores /
              softmax / accumulation fp32.  This is *more* precise than the reference's own AMP path
              (autocast also rounds conv outputs and residual adds to bf16,
              /root/reference/traiNNer/models/sr_model.py:474-478).
"""
from __future__ import annotations

import torch
from torch.nn import functional as F


def _r(t: torch.Tensor, dt: torch.dtype = torch.bfloat16) -> torch.Tensor:
    return t.to(dt).to(torch.float32)


class _ConvBf16Ops(torch.autograd.Function):
    """conv2d whose contraction operands are bf16-rounded in fwd, dgrad and wgrad."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, dt=torch.bfloat16):
        xr, wr = _r(x, dt), _r(w, dt)
        ctx.save_for_backward(xr, wr)
        ctx.stride, ctx.padding, ctx.has_bias, ctx.dt = stride, padding, b is not None, dt
        return F.conv2d(xr, wr, b, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, gy):
        xr, wr = ctx.saved_tensors
        gr = _r(gy, ctx.dt)
        gx = torch.nn.grad.conv2d_input(xr.shape, wr, gr, stride=ctx.stride, padding=ctx.padding)
        gw = torch.nn.grad.conv2d_weight(xr, wr.shape, gr, stride=ctx.stride, padding=ctx.padding)
        gb = gr.sum(dim=(0, 2, 3)) if ctx.has_bias else None
        return gx, gw, gb, None, None, None


class _MatmulBf16Ops(torch.autograd.Function):
    """a @ b with both operands bf16-rounded (fp32 accumulate); the backward contractions round the incoming gradient too
    (window attention on MFMA: q, k, v, P, dO, dS are the bf16 operands, trainner-redux_amd/csrc/attn_mfma.hip)."""

    @staticmethod
    def forward(ctx, a, b, dt=torch.bfloat16):
        ar, br = _r(a, dt), _r(b, dt)
        ctx.save_for_backward(ar, br)
        ctx.dt = dt
        return ar @ br

    @staticmethod
    def backward(ctx, g):
        ar, br = ctx.saved_tensors
        gr = _r(g, ctx.dt)
        return gr @ br.transpose(-2, -1), ar.transpose(-2, -1) @ gr, None


class Policy:
    def __init__(self, name: str) -> None:
        if name not in ("fp32", "bf16_ops", "f16_ops"):
            raise ValueError(name)
        self.name = name
        self.dt = torch.float16 if name == "f16_ops" else torch.bfloat16

    def conv2d(self, x, w, b=None, stride=1, padding=1):
        if self.name == "fp32":
            return F.conv2d(x, w, b, stride=stride, padding=padding)
        return _ConvBf16Ops.apply(x, w, b, stride, padding, self.dt)

    def matmul(self, a, b):
        if self.name == "fp32":
            return a @ b
        return _MatmulBf16Ops.apply(a, b, self.dt)


FP32 = Policy("fp32")
BF16_OPS = Policy("bf16_ops")
F16_OPS = Policy("f16_ops")     # same contract with fp16 operands: the reference's default AMP dtype (use_amp without amp_bf16)
