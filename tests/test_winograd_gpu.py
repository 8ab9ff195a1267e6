"""Fused Winograd F(2x2, 3x3) fp32 conv (csrc/winograd.hip) against plain PyTorch fp32 conv2d on the CPU and against the direct MFMA conv:
same function, different fp32 summation order.  Tolerance 2e-6 rel-L2 (Winograd's transforms cost about one more bit than the direct sum)."""
import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("n,h,w,ci,co", [(1, 8, 16, 8, 64), (2, 10, 18, 16, 64), (1, 38, 50, 24, 128), (2, 64, 64, 64, 64), (1, 32, 32, 256, 128)])
def test_winograd_forward_bias_relu_mask(cuda, n, h, w, ci, co):
    from trainner_redux_amd import _lib as L, ops

    g = torch.Generator().manual_seed(h * 100 + ci)
    x = torch.randn(n, ci, h, w, generator=g)
    wt = torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)
    bs = torch.randn(co, generator=g)
    m = torch.randn(n, co, h, w, generator=g)
    ref = F.conv2d(x, wt, bs, padding=1)
    xb = torch.randn(n, h, w, ci + 8)
    xb[..., 8:] = x.permute(0, 2, 3, 1)
    xb = xb.to(cuda)
    ob = torch.zeros(n, h, w, co + 4, device=cuda)
    U = ops.pack_winograd(wt.to(cuda))
    kw = dict(n=n, h=h, w=w, cin=ci, cout=co, compute=L.F32_WINOGRAD, inp=ops.view(xb, 8, ci), wpacked=U)
    ops.conv3x3(out=ops.view(ob, 4, co), bias=bs.to(cuda), **kw)
    got = ob[..., 4:].cpu().permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < 2e-6, rel_l2(got, ref)
    assert ob[..., :4].abs().sum() == 0
    ops.conv3x3(out=ops.view(ob, 4, co), bias=bs.to(cuda), act=L.ACT_RELU, **kw)
    assert rel_l2(ob[..., 4:].cpu().permute(0, 3, 1, 2), ref.relu()) < 2e-6
    mb = m.permute(0, 2, 3, 1).contiguous().to(cuda)
    ops.conv3x3(out=ops.view(ob, 4, co), mask=ops.view(mb), mask_slope=0.0, **kw)
    assert rel_l2(ob[..., 4:].cpu().permute(0, 3, 1, 2), F.conv2d(x, wt, None, padding=1) * (m > 0)) < 2e-6


def test_winograd_input_gradient_slab(cuda):
    """transpose=True packs the conv whose output is d/d(input): equals autograd's conv2d input gradient."""
    from trainner_redux_amd import _lib as L, ops

    g = torch.Generator().manual_seed(5)
    n, h, w, ci, co = 2, 20, 28, 64, 128
    x = torch.randn(n, ci, h, w, generator=g, requires_grad=True)
    wt = torch.randn(co, ci, 3, 3, generator=g) / 24
    gy = torch.randn(n, co, h, w, generator=g)
    F.conv2d(x, wt, None, padding=1).backward(gy)
    gb = gy.permute(0, 2, 3, 1).contiguous().to(cuda)
    ob = torch.zeros(n, h, w, ci, device=cuda)
    ops.conv3x3(n=n, h=h, w=w, cin=co, cout=ci, compute=L.F32_WINOGRAD, inp=ops.view(gb), out=ops.view(ob), wpacked=ops.pack_winograd(wt.to(cuda), transpose=True))
    assert rel_l2(ob.cpu().permute(0, 3, 1, 2), x.grad) < 2e-6


def test_winograd_rejects_what_it_does_not_build(cuda):
    from trainner_redux_amd import _lib as L, ops

    x = torch.zeros(1, 9, 8, 8, device=cuda)
    with pytest.raises(RuntimeError, match="even output dims"):
        ops.conv3x3(n=1, h=9, w=8, cin=8, cout=64, compute=L.F32_WINOGRAD, inp=ops.view(x), out=ops.view(torch.zeros(1, 9, 8, 64, device=cuda)), wpacked=x)
    with pytest.raises(RuntimeError, match="multiple of"):
        ops.conv3x3(n=1, h=8, w=8, cin=8, cout=32, compute=L.F32_WINOGRAD, inp=ops.view(torch.zeros(1, 8, 8, 8, device=cuda)),
                    out=ops.view(torch.zeros(1, 8, 8, 32, device=cuda)), wpacked=x)
