"""Seeded random sweeps of the non-conv kernels against plain PyTorch: depthwise conv (forward, flipped = input gradient, weight gradient), BatchNorm2d
(+ LeakyReLU, training and eval, backward), the optimizer sweep and its norm on sizes that are not multiples of any vector width, the pixel losses.
Same idea as tests/test_conv_fuzz_gpu.py: shapes nobody listed by hand (odd extents, channel slices at offsets, canaries around every output)."""
import math
import random

import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _slice_buf(t_nchw, dtype, cuda, g, slack, off):
    """NCHW cpu tensor -> NHWC device buffer with `slack` extra channels (random garbage), the tensor at channel offset `off`."""
    n, c, h, w = t_nchw.shape
    buf = torch.randn(n, h, w, c + slack, generator=g).to(dtype)
    buf[..., off:off + c] = t_nchw.permute(0, 2, 3, 1).to(dtype)
    return buf.to(cuda).contiguous()


@pytest.mark.parametrize("seed", range(40))
def test_dwconv_random_shape(cuda, seed):
    """`tr_dwconv` / `tr_dwconv_wgrad` (MoSR's depthwise 7x7, spandrel GatedCNNBlock.conv): kernel sizes 3 / 5 / 7, channel counts in steps of 4, fp32 and
    16-bit views (the LDS-staged kernel), slices of wider buffers."""
    from trainner_redux_amd import ops

    r = random.Random(41000 + seed)
    k = r.choice([3, 5, 7, 7])
    c = r.choice([4, 8, 12, 24, 32, 36, 48, 64, 72, 96])
    n, h, w = r.randint(1, 3), r.randint(1, 50), r.randint(1, 70)
    dtype = r.choice([torch.float32, torch.bfloat16, torch.float16])
    slack = r.choice([0, 4, 8, 12]); off = r.choice([0, 4, 8]) if slack >= 8 else (r.choice([0, 4]) if slack else 0)
    g = torch.Generator().manual_seed(43000 + seed)
    x = torch.randn(n, c, h, w, generator=g).to(dtype).float()
    wt = torch.randn(c, 1, k, k, generator=g) * 0.2
    bs = torch.randn(c, generator=g)
    gy = torch.randn(n, c, h, w, generator=g).to(dtype).float()
    xr, wr, br = x.clone().requires_grad_(True), wt.clone().requires_grad_(True), bs.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, br, padding=k // 2, groups=c)
    y_ref.backward(gy)
    what = (seed, k, c, n, h, w, dtype, slack, off)
    xb = _slice_buf(x, dtype, cuda, g, slack, off)
    xv = ops.view(xb, off, c)
    yb = torch.full((n, h, w, c + 8), 3.0, dtype=dtype, device=cuda)
    ops.dwconv(xv, ops.view(yb, 4, c), wt.to(cuda), bs.to(cuda), n, h, w, c)
    tol = 3e-6 if dtype == torch.float32 else (5e-3 if dtype == torch.bfloat16 else 7e-4)
    got = yb.float().cpu()
    assert rel_l2(got[..., 4:4 + c].permute(0, 3, 1, 2), y_ref.detach()) < tol, (what, rel_l2(got[..., 4:4 + c].permute(0, 3, 1, 2), y_ref.detach()))
    assert torch.all(got[..., :4] == 3.0) and torch.all(got[..., 4 + c:] == 3.0), (what, "bytes around the output slice were written")
    gb = _slice_buf(gy, dtype, cuda, g, 0, 0)
    dxb = torch.full((n, h, w, c), 9.0, dtype=torch.float32, device=cuda)
    ops.dwconv(ops.view(gb), ops.view(dxb), wt.to(cuda), None, n, h, w, c, flip=True)
    assert rel_l2(dxb.cpu().permute(0, 3, 1, 2), xr.grad) < 3e-6, (what, rel_l2(dxb.cpu().permute(0, 3, 1, 2), xr.grad))
    dw = torch.full((c, 1, k, k), 0.5, device=cuda)
    db = torch.full((c,), -0.25, device=cuda)
    ws = torch.empty(ops.dwconv_wgrad_workspace_bytes(n, h, w, c, k) // 4 + 4, device=cuda)
    ops.dwconv_wgrad(xv, ops.view(gb), dw, db, n, h, w, c, ws, accumulate=True)
    assert rel_l2(dw.cpu() - 0.5, wr.grad) < 2e-5 and rel_l2(db.cpu() + 0.25, br.grad) < 2e-5, (what, rel_l2(dw.cpu() - 0.5, wr.grad), rel_l2(db.cpu() + 0.25, br.grad))


@pytest.mark.parametrize("seed", range(30))
def test_batchnorm_random_shape(cuda, seed):
    """`tr_batchnorm_fwd` / `_bwd` (VGGStyleDiscriminator: conv -> BatchNorm2d -> LeakyReLU(0.2), discriminator_arch.py:31-57) against torch's BatchNorm2d in
    training mode (batch statistics, running statistics moved by momentum with the unbiased variance), eval mode, and autograd's backward through
    LeakyReLU(BN(x)) -- random pixel counts and channel counts (multiples of 4), fp32 / 16-bit activation views."""
    from trainner_redux_amd import _lib as L, ops

    r = random.Random(45000 + seed)
    c = r.choice([4, 8, 16, 32, 64, 100, 128, 256, 512])
    n, h, w = r.randint(1, 4), r.randint(1, 40), r.randint(1, 40)
    if n * h * w < 2:
        h = 2
    dtype = r.choice([torch.float32, torch.bfloat16, torch.float16])
    act = r.choice(["lrelu", "none"])
    g = torch.Generator().manual_seed(47000 + seed)
    x = (torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3).to(dtype).float()
    gam, bet = 1 + 0.2 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)
    rm0, rv0 = 0.1 * torch.randn(c, generator=g), 1 + 0.2 * torch.rand(c, generator=g)
    gy = torch.randn(n, c, h, w, generator=g).to(dtype).float()
    bn = torch.nn.BatchNorm2d(c)
    with torch.no_grad():
        bn.weight.copy_(gam); bn.bias.copy_(bet); bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
    xr = x.clone().requires_grad_(True)
    bn.train()
    y_ref = bn(xr)
    if act == "lrelu":
        y_ref = F.leaky_relu(y_ref, 0.2)
    y_ref.backward(gy)
    npix = n * h * w
    what = (seed, c, n, h, w, dtype, act)
    xd = x.permute(0, 2, 3, 1).contiguous().to(cuda, dtype)
    yd = torch.full((n, h, w, c + 4), 3.0, dtype=dtype, device=cuda)
    rm, rv = rm0.to(cuda), rv0.to(cuda)
    sm, sr = torch.empty(c, device=cuda), torch.empty(c, device=cuda)
    ws = ops.batchnorm_workspace(c, cuda)
    a = L.ACT_LRELU if act == "lrelu" else L.ACT_NONE
    ops.batchnorm_fwd(ops.view(xd), ops.view(yd, 0, c), npix, c, gam.to(cuda), bet.to(cuda), rm, rv, sm, sr, ws, training=True, act=a, slope=0.2)
    tol = 3e-6 if dtype == torch.float32 else (5e-3 if dtype == torch.bfloat16 else 7e-4)
    got = yd.float().cpu()
    assert rel_l2(got[..., :c].permute(0, 3, 1, 2), y_ref.detach()) < tol, (what, rel_l2(got[..., :c].permute(0, 3, 1, 2), y_ref.detach()))
    assert torch.all(got[..., c:] == 3.0), (what, "bytes behind the output were written")
    assert rel_l2(rm.cpu(), bn.running_mean) < 1e-5 and rel_l2(rv.cpu(), bn.running_var) < 1e-5, (what, rel_l2(rm.cpu(), bn.running_mean), rel_l2(rv.cpu(), bn.running_var))
    gyd = gy.permute(0, 2, 3, 1).contiguous().to(cuda, dtype)
    dx = torch.full((n, h, w, c), 9.0, dtype=torch.float32, device=cuda)
    dgam, dbet = torch.full((c,), 2.0, device=cuda), torch.full((c,), 2.0, device=cuda)
    ops.batchnorm_bwd(ops.view(gyd), ops.view(xd), ops.view(yd, 0, c) if act == "lrelu" else ops.NULL_VIEW, ops.view(dx), npix, c, gam.to(cuda), sm, sr, dgam, dbet, ws,
                      slope=0.2, accumulate=True)
    # the activation's derivative is taken from the sign of the STORED y: in 16-bit views a y that rounds to +-0 may sit on the other side than the fp32 reference's
    btol = 5e-6 if dtype == torch.float32 else 2e-2
    assert rel_l2(dx.cpu().permute(0, 3, 1, 2), xr.grad) < btol, (what, rel_l2(dx.cpu().permute(0, 3, 1, 2), xr.grad))
    assert rel_l2(dgam.cpu() - 2.0, bn.weight.grad) < btol * 2 and rel_l2(dbet.cpu() - 2.0, bn.bias.grad) < btol * 2, what
    # eval mode: the (updated) running statistics normalise
    bn.eval()
    with torch.no_grad():
        y_eval = bn(x)
        if act == "lrelu":
            y_eval = F.leaky_relu(y_eval, 0.2)
    ye = torch.empty(n, h, w, c, dtype=dtype, device=cuda)
    ops.batchnorm_fwd(ops.view(xd), ops.view(ye), npix, c, gam.to(cuda), bet.to(cuda), rm, rv, None, None, ws, training=False, act=a, slope=0.2)
    assert rel_l2(ye.float().cpu().permute(0, 3, 1, 2), y_eval) < tol, what


@pytest.mark.parametrize("seed", range(24))
def test_optimizer_sweep_and_norm_on_odd_sizes(cuda, seed):
    """`tr_sumsq` and `tr_adamw_ema` on arenas whose length is a multiple of 4 only (the arena's granularity), against `torch.optim.AdamW`, with clipping by the
    device-resident norm, the EMA lerp and zero_grad folded in."""
    from trainner_redux_amd import ops

    r = random.Random(49000 + seed)
    numel = 4 * r.choice([1, 3, 17, 255, 256, 257, 1023, 4099, 65537, 262145, 1000003])
    clip = r.choice([0.0, 0.5])
    wd = r.choice([0.0, 0.01])
    g = torch.Generator().manual_seed(51000 + seed)
    p0 = torch.randn(numel, generator=g)
    gr = torch.randn(numel, generator=g) * 0.1
    ema0 = torch.randn(numel, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=2e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    p, m, v, ema = p0.to(cuda), torch.zeros(numel, device=cuda), torch.zeros(numel, device=cuda), ema0.to(cuda)
    norm = torch.zeros(1, device=cuda)
    for step in (1, 2, 3):
        gs = gr * step
        ref.grad = gs.clone()
        if clip:
            torch.nn.utils.clip_grad_norm_([ref], clip)
        opt.step()
        gd = gs.to(cuda)
        ops.sumsq_norm(gd, norm)
        assert abs(float(norm) - float(gs.double().norm())) < 1e-5 * float(gs.double().norm()), (seed, numel, float(norm), float(gs.double().norm()))
        ops.adamw_ema(p, gd, m, v, ema, lr=2e-4, beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=wd, step=step, ema_decay=0.99, clip=clip, grad_norm=norm, zero_grad=True)
        ema0 = ema0 * 0.99 + ref.detach() * 0.01
        assert torch.all(gd == 0), (seed, numel, "zero_grad missed elements")
    assert rel_l2(p.cpu(), ref.detach()) < 2e-6, (seed, numel, rel_l2(p.cpu(), ref.detach()))
    assert rel_l2(ema.cpu(), ema0) < 2e-6, (seed, numel, rel_l2(ema.cpu(), ema0))
