"""Parity of every C-ABI kernel (called through the ctypes binding, i.e. through the C ABI) against
plain PyTorch fp32 CPU references of the same op.  Tolerances:
  * fp32 compute (v_mfma_f32_32x32x2_f32 is an exact fp32 fma chain): rel-L2 <= 2e-6
  * bf16 compute on bf16-representable operands, fp32 output: rel-L2 <= 2e-6 (only the summation
    order differs); bf16 output adds one RNE rounding: max elementwise rel <= 2^-8, rel-L2 <= 1e-3
    (the tolerance BASELINE.json north_star states for bf16 activations).
"""
import math

import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def nhwc(t):  # NCHW -> NHWC contiguous
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def ct(L, name):
    """compute-dtype name -> (C-ABI code, torch dtype, operand rounding)"""
    if name == "f32":
        return L.F32, torch.float32, (lambda t: t)
    if name == "bf16":
        return L.BF16, torch.bfloat16, bf
    return L.F16, torch.float16, (lambda t: t.to(torch.float16).to(torch.float32))


def pack(ops, L, w, cin, cout, compute, dev, transpose=False):
    buf = torch.zeros(ops.packed_bytes(cin, cout, compute), dtype=torch.uint8, device=dev)
    plan = ops.PackPlan(dev)
    plan.add(w, buf, cin, cout, compute, transpose=transpose)
    plan.run()
    return buf


@pytest.mark.parametrize("compute_name", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 19, 37, 64, 32), (1, 8, 8, 96, 32), (1, 33, 64, 192, 64), (1, 16, 16, 8, 64), (2, 12, 40, 64, 3),
                                   (1, 9, 31, 32, 128)])
def test_conv3x3_forward(cuda, compute_name, shape):
    import trainner_redux_amd  # noqa: F401
    from trainner_redux_amd import _lib as L, ops

    n, h, w, cin, cout = shape
    compute, dt, rnd = ct(L, compute_name)
    g = torch.Generator().manual_seed(1234 + cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    x, wt = rnd(x), rnd(wt)
    ref = F.leaky_relu(F.conv2d(x, wt, b, padding=1), 0.2)
    xd = nhwc(x).to(cuda, dt)
    wp = pack(ops, L, wt.to(cuda), cin, cout, compute, cuda)
    out = torch.empty(n, h, w, cout, dtype=torch.float32, device=cuda)
    bd = b.to(cuda)  # NOTE: ops.view()/data_ptr() do not keep tensors alive -- every device operand is a named variable
    ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=compute, inp=ops.view(xd), out=ops.view(out), wpacked=wp, bias=bd,
                act=L.ACT_LRELU, slope=0.2)
    got = nchw(out.cpu())
    assert rel_l2(got, ref) < 2e-6, rel_l2(got, ref)


def test_conv3x3_bf16_output_rounding(cuda):
    from trainner_redux_amd import _lib as L, ops

    n, h, w, cin, cout = 1, 24, 48, 64, 64
    g = torch.Generator().manual_seed(7)
    x = bf(torch.randn(n, cin, h, w, generator=g))
    wt = bf(torch.randn(cout, cin, 3, 3, generator=g) / 24)
    ref = F.conv2d(x, wt, None, padding=1)
    wp = pack(ops, L, wt.to(cuda), cin, cout, L.BF16, cuda)
    out = torch.empty(n, h, w, cout, dtype=torch.bfloat16, device=cuda)
    xd = nhwc(x).to(cuda, torch.bfloat16)
    ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.BF16, inp=ops.view(xd), out=ops.view(out), wpacked=wp)
    got = nchw(out.float().cpu())
    # one RNE rounding to bf16 is by itself 2^-9/sqrt(3)-ish ~ 1.7e-3 rel-L2 against the unrounded fp32 value ...
    assert rel_l2(got, ref) < 2.5e-3
    assert ((got - ref).abs() <= ref.abs() * 2 ** -8 + 1e-6).all()
    # ... so the meaningful check is against the oracle rounded the same way: <= 1e-3 (in fact ~1e-5: rare 1-ulp flips)
    assert rel_l2(got, bf(ref)) < 1e-3
    # bit-exact against rounding the fp32 reference wherever the fp32 sums agree to within half an ulp
    assert (got == bf(ref)).float().mean() > 0.99


@pytest.mark.parametrize("compute_name", ["f32", "bf16"])
def test_conv3x3_dense_slices_residuals_and_dual_output(cuda, compute_name):
    """In-place dense-block concat (channel slices of one 192-channel buffer), `x5*0.2 + x` then
    `out*0.2 + x` (scunet_aaf6aa_arch.py:334-336,355-360), dual fp32 + compute-dtype stores."""
    from trainner_redux_amd import _lib as L, ops

    compute = L.F32 if compute_name == "f32" else L.BF16
    dt = torch.float32 if compute == L.F32 else torch.bfloat16
    n, h, w = 2, 20, 36
    g = torch.Generator().manual_seed(3)
    buf = torch.randn(n, h, w, 192, generator=g)
    if compute == L.BF16:
        buf = bf(buf)
    w2 = torch.randn(32, 96, 3, 3, generator=g) / 30
    w5 = torch.randn(64, 192, 3, 3, generator=g) / 40
    b2 = torch.randn(32, generator=g)
    b5 = torch.randn(64, generator=g)
    if compute == L.BF16:
        w2, w5 = bf(w2), bf(w5)
    r1 = torch.randn(n, h, w, 64, generator=g)
    r2 = torch.randn(n, h, w, 64, generator=g)
    x_nchw = nchw(buf)
    ref2 = F.leaky_relu(F.conv2d(x_nchw[:, :96], w2, b2, padding=1), 0.2)
    bd = buf.to(cuda, dt)
    wp2, wp5 = pack(ops, L, w2.to(cuda), 96, 32, compute, cuda), pack(ops, L, w5.to(cuda), 192, 64, compute, cuda)
    b2d, b5d, r1d, r2d = b2.to(cuda), b5.to(cuda), r1.to(cuda), r2.to(cuda)
    ops.conv3x3(n=n, h=h, w=w, cin=96, cout=32, compute=compute, inp=ops.view(bd, 0, 96), out=ops.view(bd, 96, 32),
                wpacked=wp2, bias=b2d, act=L.ACT_LRELU)
    got2 = nchw(bd[..., 96:128].float().cpu())
    tol = 2e-6 if compute == L.F32 else 1e-3
    assert rel_l2(got2, ref2 if compute == L.F32 else bf(ref2)) < tol
    # untouched channels stay untouched
    assert torch.equal(bd[..., :96].float().cpu(), buf[..., :96]) and torch.equal(bd[..., 128:].float().cpu(), buf[..., 128:])
    full = nchw(bd.float().cpu())
    ref5 = (F.conv2d(full, w5, b5, padding=1) * 0.2 + nchw(r1)) * 0.2 + nchw(r2)
    o32 = torch.empty(n, h, w, 64, dtype=torch.float32, device=cuda)
    o2 = torch.zeros(n, h, w, 192, dtype=dt, device=cuda)
    ops.conv3x3(n=n, h=h, w=w, cin=192, cout=64, compute=compute, inp=ops.view(bd), out=ops.view(o32), out2=ops.view(o2, 0, 64),
                out2_scale=0.5, wpacked=wp5, bias=b5d, res1=ops.view(r1d), scale1=0.2, res2=ops.view(r2d), scale2=0.2)
    assert rel_l2(nchw(o32.cpu()), ref5) < 2e-6
    assert rel_l2(nchw(o2[..., :64].float().cpu()), ref5 * 0.5 if compute == L.F32 else bf(ref5 * 0.5)) < tol
    assert o2[..., 64:].abs().max().item() == 0


@pytest.mark.parametrize("compute_name", ["f32", "bf16"])
def test_conv3x3_nearest2x_and_replicate(cuda, compute_name):
    from trainner_redux_amd import _lib as L, ops

    compute = L.F32 if compute_name == "f32" else L.BF16
    dt = torch.float32 if compute == L.F32 else torch.bfloat16
    n, h, w, c = 1, 10, 18, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, c, 3, 3, generator=g) / 24
    if compute == L.BF16:
        x, wt = bf(x), bf(wt)
    wp = pack(ops, L, wt.to(cuda), c, c, compute, cuda)
    xd = nhwc(x).to(cuda, dt)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, None, padding=1)
    out = torch.empty(n, 2 * h, 2 * w, c, dtype=torch.float32, device=cuda)
    ops.conv3x3(n=n, h=2 * h, w=2 * w, cin=c, cout=c, compute=compute, inp=ops.view(xd), out=ops.view(out), wpacked=wp, in_mode=L.IN_NEAREST2X)
    assert rel_l2(nchw(out.cpu()), ref) < 2e-6
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), wt, None)
    out = torch.empty(n, h, w, c, dtype=torch.float32, device=cuda)
    ops.conv3x3(n=n, h=h, w=w, cin=c, cout=c, compute=compute, inp=ops.view(xd), out=ops.view(out), wpacked=wp, pad_mode=L.PAD_REPLICATE)
    assert rel_l2(nchw(out.cpu()), ref) < 2e-6


@pytest.mark.parametrize("compute_name", ["f32", "bf16"])
def test_conv3x3_input_gradient_via_transposed_pack_and_mask(cuda, compute_name):
    """dgrad = the same kernel on flipped/transposed weights; LeakyReLU derivative applied as an epilogue mask."""
    from trainner_redux_amd import _lib as L, ops

    compute = L.F32 if compute_name == "f32" else L.BF16
    dt = torch.float32 if compute == L.F32 else torch.bfloat16
    n, h, w, cin, cout = 2, 13, 35, 96, 32
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 30
    gy = torch.randn(n, cout, h, w, generator=g)
    pre = torch.randn(n, cin, h, w, generator=g)  # stands for the saved activation the mask is taken from
    if compute == L.BF16:
        wt, gy, pre = bf(wt), bf(gy), bf(pre)
    ref = torch.nn.grad.conv2d_input(x.shape, wt, gy, padding=1) * torch.where(pre > 0, 1.0, 0.2)
    wp = pack(ops, L, wt.to(cuda), cout, cin, compute, cuda, transpose=True)
    out = torch.empty(n, h, w, cin, dtype=torch.float32, device=cuda)
    gyd, pred = nhwc(gy).to(cuda, dt), nhwc(pre).to(cuda, dt)
    ops.conv3x3(n=n, h=h, w=w, cin=cout, cout=cin, compute=compute, inp=ops.view(gyd), out=ops.view(out), wpacked=wp,
                mask=ops.view(pred), mask_slope=0.2)
    assert rel_l2(nchw(out.cpu()), ref) < 2e-6


@pytest.mark.parametrize("compute_name", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("impl", [0, 1, 2], ids=["auto", "reg", "ring"])
@pytest.mark.parametrize("shape", [(2, 19, 37, 64, 32), (1, 8, 8, 96, 32), (1, 33, 70, 192, 64), (1, 16, 16, 8, 64), (2, 12, 40, 64, 8),
                                   (1, 20, 40, 64, 96), (2, 16, 32, 96, 160), (1, 24, 32, 192, 96)])      # cout = 64 k + 32: half a last output group
def test_wgrad3x3(cuda, compute_name, shape, impl):
    from trainner_redux_amd import _lib as L, ops

    if impl and compute_name == "f32":
        pytest.skip("one kernel family for fp32")
    compute, dt, rnd = ct(L, compute_name)
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(21 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    gy = torch.randn(n, cout, h, w, generator=g)
    x, gy = rnd(x), rnd(gy)
    ref_w = torch.nn.grad.conv2d_weight(x, (cout, cin, 3, 3), gy, padding=1)
    ref_b = gy.sum(dim=(0, 2, 3))
    ws = torch.empty(ops.wgrad_workspace_bytes(n, h, w, cin, cout) // 4, dtype=torch.float32, device=cuda)
    dw = torch.full((cout, cin, 3, 3), 7.0, device=cuda)
    db = torch.full((cout,), 7.0, device=cuda)
    xd, gyd = nhwc(x).to(cuda, dt), nhwc(gy).to(cuda, dt)
    kw = dict(n=n, h=h, w=w, cin=cin, cout=cout, compute=compute, inp=ops.view(xd), gy=ops.view(gyd), dw=dw, db=db, workspace=ws)
    with ops.tuned(wgrad_impl=impl):
        ops.wgrad3x3(**kw)
        w1, b1 = dw.cpu().clone(), db.cpu().clone()
        ops.wgrad3x3(**kw, accumulate=True)
    tol = 3e-6 if compute == L.F32 else 3e-6
    assert rel_l2(w1, ref_w) < tol, rel_l2(w1, ref_w)
    assert rel_l2(b1, ref_b) < 3e-6
    assert rel_l2(dw.cpu(), 2 * ref_w) < tol
    assert rel_l2(db.cpu(), 2 * ref_b) < 3e-6


@pytest.mark.parametrize("compute_name", ["bf16", "f16"])
@pytest.mark.parametrize("geom", [(2, 40, 72), (1, 17, 33)])
def test_wgrad3x3_multi_dense_block(cuda, compute_name, geom):
    """tr_wgrad3x3_multi: the five weight / bias gradients of a dense block (spandrel ESRGAN ResidualDenseBlock_5C: conv k reads
    cat(x, x1..x_{k-1}), here channel prefixes of one buffer, and its own slice of the block's gradient buffer) in one launch of the ring
    kernel, against torch's conv2d_weight on the same rounded operands; accumulation on top of existing gradients; and the fallback
    (five separate launches) on the same inputs."""
    from trainner_redux_amd import _lib as L, ops

    compute, dt, rnd = ct(L, compute_name)
    n, h, w = geom
    nf, gc = 64, 32
    g = torch.Generator().manual_seed(5 + h)
    x = rnd(torch.randn(n, nf + 4 * gc, h, w, generator=g))
    gy = rnd(torch.randn(n, nf + 4 * gc, h, w, generator=g) / 8)
    xd, gyd = nhwc(x).to(cuda, dt), nhwc(gy).to(cuda, dt)
    shapes = [(nf + gc * (c - 1), gc if c < 5 else nf, 0 if c == 5 else nf + gc * (4 - c)) for c in range(1, 6)]
    need = max(ops.wgrad_workspace_bytes(n, h, w, ci, co) for ci, co, _ in shapes)
    for fused in (1, 0):
        ops.tune_set("wgrad_impl", 2)      # these sizes are below the ring kernel's automatic threshold
        ops.tune_set("wgrad_multi", fused)
        dws = [torch.full((co, ci, 3, 3), 0.5, device=cuda) for ci, co, _ in shapes]
        dbs = [torch.full((co,), 0.25, device=cuda) for ci, co, _ in shapes]
        wss = [torch.empty(need // 4 + 16, dtype=torch.float32, device=cuda) for _ in shapes]
        convs = [dict(cin=ci, cout=co, inp=ops.view(xd, 0, ci), gy=ops.view(gyd, go, co), dw=dws[k], db=dbs[k], workspace=wss[k])
                 for k, (ci, co, go) in enumerate(shapes)]
        try:
            ops.wgrad3x3_multi(convs, n=n, h=h, w=w, compute=compute, accumulate=True)
        finally:
            ops.tune_set("wgrad_impl", 0)
            ops.tune_set("wgrad_multi", 1)
        for k, (ci, co, go) in enumerate(shapes):
            ref_w = torch.nn.grad.conv2d_weight(x[:, :ci], (co, ci, 3, 3), gy[:, go:go + co], padding=1)
            ref_b = gy[:, go:go + co].sum(dim=(0, 2, 3))
            assert rel_l2(dws[k].cpu() - 0.5, ref_w) < 3e-6, (fused, k)
            assert rel_l2(dbs[k].cpu() - 0.25, ref_b) < 3e-6, (fused, k)


def test_wgrad3x3_is_bitwise_reproducible(cuda):
    from trainner_redux_amd import _lib as L, ops

    n, h, w, cin, cout = 2, 40, 72, 64, 64
    g = torch.Generator().manual_seed(2)
    x = nhwc(torch.randn(n, cin, h, w, generator=g)).to(cuda, torch.bfloat16)
    gy = nhwc(torch.randn(n, cout, h, w, generator=g)).to(cuda, torch.bfloat16)
    ws = torch.empty(ops.wgrad_workspace_bytes(n, h, w, cin, cout) // 4, dtype=torch.float32, device=cuda)
    outs = []
    for _ in range(3):
        dw = torch.empty(cout, cin, 3, 3, device=cuda)
        ops.wgrad3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.BF16, inp=ops.view(x), gy=ops.view(gy), dw=dw, db=None, workspace=ws)
        outs.append(dw.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("kind_name", ["l1", "charbonnier", "mse"])
def test_pixel_loss_fused_with_pool_backward(cuda, kind_name):
    """tr_pixel_loss_pool_bwd: the criterion on a tapped fp32 activation that also feeds max_pool2d(relu(.), 2, 2)
    (perceptual_fp16_loss.py:12-50: VGG19's conv -> ReLU -> MaxPool with the pre-ReLU feature tapped), gradient = loss term + the pool's
    backward through the ReLU in one pass, against autograd (ties and non-positive windows included)."""
    from trainner_redux_amd import _lib as L, ops

    kind = {"l1": L.LOSS_L1, "mse": L.LOSS_MSE, "charbonnier": L.LOSS_CHARBONNIER}[kind_name]
    n, h, w, c = 2, 12, 20, 16
    g = torch.Generator().manual_seed(31)
    act = torch.randn(n, c, h, w, generator=g)
    act[0, 0, 0, :4] = 0.7          # ties inside a window: the first maximal element takes the gradient
    act[0, 1, :2, :2] = -1.0        # a window with no positive value: nothing flows through the ReLU
    gt = torch.randn(n, c, h, w, generator=g)
    gout = torch.randn(n, c, h // 2, w // 2, generator=g)
    a = act.clone().requires_grad_(True)
    eps = 1e-6
    d = a - gt
    crit = d.abs().mean() if kind == L.LOSS_L1 else (d * d).mean() if kind == L.LOSS_MSE else torch.sqrt(d * d + eps).mean()
    lw = 0.37
    (lw * crit + (F.max_pool2d(F.relu(a), 2, 2) * gout).sum()).backward()
    ad, gtd, god = nhwc(act).to(cuda), nhwc(gt).to(cuda), nhwc(gout).to(cuda)
    grad = torch.full((n, h, w, c), 9.0, device=cuda)
    val = torch.zeros(1, device=cuda)
    tot = torch.zeros(1, device=cuda)
    ops.pixel_loss(kind, ad, gtd, val, value_weight=lw, total_out=tot, total_weight=lw, grad=grad, grad_weight=lw, eps=eps, pool_gout=god)
    assert abs(val.item() - lw * crit.item()) < 1e-6 * max(1.0, abs(crit.item())) and abs(tot.item() - val.item()) < 1e-7
    assert rel_l2(nchw(grad.cpu()), a.grad) < 1e-6


def test_layout_and_pool(cuda):
    from trainner_redux_amd import _lib as L, ops

    g = torch.Generator().manual_seed(9)
    img = torch.rand(2, 3, 10, 14, generator=g)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    for dt in (torch.float32, torch.bfloat16):
        dst = torch.full((2, 10, 14, 8), 5.0, dtype=dt, device=cuda)
        imgd = img.to(cuda)
        ops.nchw_to_nhwc(imgd, dst, mean, std)
        ref = (img - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
        got = dst.float().cpu()
        assert got[..., 3:].abs().max() == 0
        assert rel_l2(nchw(got[..., :3]), ref) < (1e-6 if dt == torch.float32 else 3e-3)
    back = torch.empty(2, 3, 10, 14, device=cuda)
    d32 = torch.zeros(2, 10, 14, 4, device=cuda)
    ops.nchw_to_nhwc(imgd, d32)
    ops.nhwc_to_nchw(ops.view(d32), back)
    assert torch.equal(back.cpu(), img)
    # maxpool (+ fused relu) forward / backward against ATen
    x = torch.randn(2, 16, 12, 20, generator=g)
    x[0, 0, 0, :4] = 0.0  # ties
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(F.relu(xr), 2, 2)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = nhwc(x).to(cuda)
    yd = torch.empty(2, 6, 10, 16, device=cuda)
    ops.maxpool2x2(ops.view(xd), ops.view(yd), 2, 12, 20, True)
    assert torch.equal(nchw(yd.cpu()), y.detach())
    gxd = torch.empty_like(xd)
    gyd = nhwc(gy).to(cuda)
    ops.maxpool2x2_bwd(ops.view(xd), ops.view(gyd), ops.view(gxd), 2, 12, 20, True)
    assert torch.equal(nchw(gxd.cpu()), xr.grad)


@pytest.mark.parametrize("kind", ["l1", "mse", "charbonnier"])
@pytest.mark.parametrize("numel_shape", [(2, 3, 33, 47), (1, 3, 64, 64), (1, 1, 1, 3)])
def test_pixel_loss(cuda, kind, numel_shape):
    from oracle import losses as OL
    from trainner_redux_amd import _lib as L, ops

    g = torch.Generator().manual_seed(4)
    pred = torch.rand(numel_shape, generator=g).requires_grad_(True)
    tgt = torch.rand(numel_shape, generator=g)
    with torch.no_grad():
        tgt.view(-1)[0] = pred.view(-1)[0]  # exact-zero difference: sign(0) = 0
    fn = {"l1": OL.l1, "mse": OL.mse, "charbonnier": OL.charbonnier}[kind]
    code = {"l1": L.LOSS_L1, "mse": L.LOSS_MSE, "charbonnier": L.LOSS_CHARBONNIER}[kind]
    ref = fn(pred, tgt) * 0.7
    ref.backward()
    out = torch.zeros(1, device=cuda)
    grad = torch.empty(numel_shape, device=cuda)
    pd, td = pred.detach().to(cuda), tgt.to(cuda)
    tot = torch.full((1,), 10.0, device=cuda)
    ops.pixel_loss(code, pd, td, out, value_weight=0.7, grad=grad, total_out=tot, total_weight=0.35)
    assert abs(tot.item() - 10.0 - 0.5 * ref.item()) <= 1e-5
    assert abs(out.item() - ref.item()) <= 2e-6 * abs(ref.item()) + 1e-9
    assert rel_l2(grad.cpu(), pred.grad) < 2e-6
    ops.pixel_loss(code, pd, td, out, value_weight=0.7, grad=grad, grad_weight=1.4, accumulate_grad=True)
    assert abs(out.item() - ref.item()) <= 2e-6 * abs(ref.item()) + 1e-9
    assert rel_l2(grad.cpu(), 3 * pred.grad) < 2e-6
    # reference invariant (tests/test_losses/test_losses.py:114-128): black vs black is ~0
    z = torch.zeros(numel_shape, device=cuda)
    ops.pixel_loss(code, z, z, out)
    assert out.item() <= 1e-5


def test_adamw_ema_clip_against_oracle(cuda):
    from collections import OrderedDict

    from oracle import step as OS
    from trainner_redux_amd import ops

    g = torch.Generator().manual_seed(8)
    nel = 100003
    p0 = torch.randn(nel, generator=g)
    params = OrderedDict(p=p0.clone())
    st = OS.AdamWState(params, lr=2e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01)
    ema = OS.EMAState(params, beta=0.999, power=0.75)
    p = p0.to(cuda)
    m = torch.zeros(nel, device=cuda)
    v = torch.zeros(nel, device=cuda)
    e = p.clone()
    gn = torch.zeros(1, device=cuda)
    ema_step = 0
    for it in range(1, 6):
        grad = torch.randn(nel, generator=g) * (3.0 if it % 2 else 0.001)
        gref = OrderedDict(p=grad.clone())
        total = OS.clip_grads_(gref, 1.0)
        st.step(params, gref)
        ema.update(params)
        gd = grad.to(cuda)
        ops.sumsq_norm(gd, gn)
        assert abs(gn.item() - total.item()) < 1e-5 * total.item()
        # ema decay for this step, host side (same schedule as the oracle)
        s = ema_step
        ema_step += 1
        if s <= 1:  # s == 0: warm-up copy; s == 1: the one-off `initted` copy
            decay = 0.0
        else:
            ep = max(ema_step - 0 - 1, 0)
            decay = 0.0 if ep <= 0 else min(max(1 - (1 + ep) ** -0.75, 0.0), 0.999)
        ops.adamw_ema(p, gd, m, v, e, lr=2e-4, beta1=0.9, beta2=0.99, eps=1e-8, weight_decay=0.01, step=it, ema_decay=decay, clip=1.0,
                      grad_norm=gn, zero_grad=True)
        assert gd.abs().max().item() == 0
        assert rel_l2(p.cpu(), params["p"]) < 1e-6
        assert rel_l2(e.cpu(), ema.shadow["p"]) < 1e-6


@pytest.mark.parametrize("compute_name", ["f32", "bf16"])
def test_stem_input_gradient_full_mode_and_replicate_fold(cuda, compute_name):
    """Input gradient of a replicate-padded 3-channel stem conv (VGG conv1_1): 'full'-mode MFMA conv on the padded domain +
    border fold + 1/std, against autograd; and the VALU reference kernel tr_conv3x3_dgrad_image against the same."""
    from trainner_redux_amd import _lib as L, ops

    compute = L.F32 if compute_name == "f32" else L.BF16
    dt = torch.float32 if compute == L.F32 else torch.bfloat16
    n, h, w, cout = 2, 14, 38, 64
    g = torch.Generator().manual_seed(13)
    img = torch.rand(n, 3, h, w, generator=g).requires_grad_(True)
    wt = torch.randn(cout, 3, 3, 3, generator=g) / 5
    gy = torch.randn(n, cout, h, w, generator=g)
    std = (0.229, 0.224, 0.225)
    if compute == L.BF16:
        wt, gy = bf(wt), bf(gy)
    xin = img / torch.tensor(std).view(1, 3, 1, 1)
    F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="replicate"), wt).backward(gy)
    wd, gyd = wt.to(cuda), nhwc(gy).to(cuda, dt)
    slab = torch.zeros(ops.packed_bytes(cout, 3, compute), dtype=torch.uint8, device=cuda)
    plan = ops.PackPlan(cuda)
    plan.add(wd, slab, cout, 3, compute, transpose=True)
    plan.run()
    gp = torch.empty(n, h + 2, w + 2, 4, device=cuda)
    ops.conv3x3(n=n, h=h + 2, w=w + 2, cin=cout, cout=3, compute=compute, inp=ops.view(gyd), out=ops.view(gp), wpacked=slab, in_mode=L.IN_FULL)
    out = torch.full((n, 3, h, w), 2.0, device=cuda)
    ops.fold_replicate_nchw(ops.view(gp), out, ch_scale=[1 / s for s in std], accumulate=True)
    assert rel_l2(out.cpu() - 2.0, img.grad) < 3e-6
    out2 = torch.empty(n, 3, h, w, device=cuda)
    ops.conv3x3_dgrad_image(ops.view(gyd), wd, n, h, w, out2, pad_mode=L.PAD_REPLICATE, ch_scale=[1 / s for s in std])
    assert rel_l2(out2.cpu(), img.grad) < 3e-6


@pytest.mark.parametrize("pad", ["replicate", "zeros"])
@pytest.mark.parametrize("shape", [(2, 14, 38, 64, 0), (1, 8, 32, 64, 0), (1, 1, 5, 64, 0), (3, 40, 65, 64, 16), (1, 9, 12, 32, 0)])
def test_stem_input_gradient_tiled_kernel(cuda, pad, shape):
    """tr_conv3x3_dgrad_image (input gradient of VGG19 conv1_1, perceptual_fp16_loss.py:272-274,322 under autograd): cout = 64 fp32 takes the
    LDS-tiled kernel (partial tiles, one-row images, border folding of replicate padding, a channel slice of a wider buffer, accumulate);
    cout = 32 takes the per-pixel reference kernel.  fp32 FMA chains in another order than autograd's: 3e-6."""
    from trainner_redux_amd import _lib as L, ops

    n, h, w, cout, coff = shape
    g = torch.Generator().manual_seed(n * 100 + h + w)
    img = torch.rand(n, 3, h, w, generator=g).requires_grad_(True)
    wt = torch.randn(cout, 3, 3, 3, generator=g) / 5
    gy = torch.randn(n, cout, h, w, generator=g)
    std = (0.229, 0.224, 0.225)
    xin = img / torch.tensor(std).view(1, 3, 1, 1)
    xp = F.pad(xin, (1, 1, 1, 1), mode="replicate") if pad == "replicate" else F.pad(xin, (1, 1, 1, 1))
    F.conv2d(xp, wt).backward(gy)
    buf = torch.randn(n, h, w, cout + coff + 4, generator=g).to(cuda)
    buf[..., coff:coff + cout] = nhwc(gy).to(cuda)
    out = torch.full((n, 3, h, w), 2.0, device=cuda)
    ops.conv3x3_dgrad_image(ops.view(buf, coff, cout), wt.to(cuda), n, h, w, out, pad_mode=L.PAD_REPLICATE if pad == "replicate" else L.PAD_ZEROS,
                            ch_scale=[1 / s for s in std], accumulate=True)
    assert rel_l2(out.cpu() - 2.0, img.grad) < 3e-6


@pytest.mark.parametrize("pad", ["replicate", "zeros"])
@pytest.mark.parametrize("shape", [(2, 14, 38, 64, True), (1, 9, 7, 64, False), (1, 33, 20, 24, True), (1, 8, 16, 128, True)])
def test_stem_conv_from_nchw_image(cuda, pad, shape):
    """tr_conv3x3_stem (VGG19 conv1_1, perceptual_fp16_loss.py:272-274,322): normalise + pad + 3x3 conv + bias + ReLU straight from the NCHW
    image into a channel slice of an NHWC buffer, against torch in fp32 (fp32 FMA chains in another order: 1e-6)."""
    from trainner_redux_amd import _lib as L, ops

    n, h, w, cout, relu = shape
    g = torch.Generator().manual_seed(17)
    img = torch.rand(n, 3, h, w, generator=g)
    wt = torch.randn(cout, 3, 3, 3, generator=g) / 5
    b = torch.randn(cout, generator=g)
    mean, std = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
    xin = (img - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    ref = F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="replicate" if pad == "replicate" else "constant"), wt, b)
    if relu:
        ref = ref.relu()
    buf = torch.full((n, h, w, cout + 8), 7.0, device=cuda)
    ops.conv3x3_stem(img.to(cuda), wt.to(cuda), b.to(cuda), ops.view(buf, 4, cout), mean=mean, std=std,
                     pad_mode=L.PAD_REPLICATE if pad == "replicate" else L.PAD_ZEROS, act=L.ACT_RELU if relu else L.ACT_NONE)
    got = buf[..., 4:4 + cout].permute(0, 3, 1, 2).cpu()
    assert rel_l2(got, ref) < 2e-6
    assert torch.all(buf[..., :4] == 7.0) and torch.all(buf[..., 4 + cout:] == 7.0)      # the slice's neighbours are untouched


@pytest.mark.parametrize("linear_impl", [0, 1])
@pytest.mark.parametrize("compute_name", ["f32", "bf16", "f16"])
@pytest.mark.parametrize("dims", [(2, 16, 32, 192, 180), (1, 8, 32, 64, 60), (2, 24, 32, 384, 180), (2, 18, 32, 192, 540), (1, 8, 32, 192, 256),
                                  (1, 6, 32, 192, 300), (2, 16, 32, 192, 360), (1, 9, 32, 64, 120), (1, 8, 48, 64, 60), (2, 5, 32, 64, 24)])
def test_linear_as_1x1_conv_forward_backward(cuda, compute_name, dims, linear_impl):
    """nn.Linear over tokens (drct_arch.py:64-66,166-168) = the MFMA kernel with ksize 1 on an n x h x 32 token grid:
    forward with exact GELU (+ pre-activation side output), input gradient with the GELU-derivative mask, weight/bias gradient.
    linear_impl 0: the LDS-DMA staged token GEMM (16-bit operands, the default), 1: the register-staged one."""
    from trainner_redux_amd import _lib as L, ops

    if linear_impl == 1 and compute_name == "f32":
        pytest.skip("fp32 has one token-GEMM kernel (covered by linear_impl 0)")
    ops.tune_set("linear_impl", linear_impl)
    try:
        _linear_case(cuda, compute_name, dims, linear_impl)
    finally:
        ops.tune_set("linear_impl", 0)


def _linear_case(cuda, compute_name, dims, linear_impl):
    from trainner_redux_amd import _lib as L, ops

    compute, dt, rnd = ct(L, compute_name)
    n, h, w, kp, nout = dims            # kp: padded input width (multiple of 32), k = kp - 12 real inputs
    k = kp - 12 if kp > 64 else kp - 4
    g = torch.Generator().manual_seed(kp + nout)
    x = torch.randn(n * h * w, k, generator=g)
    wt = torch.randn(nout, k, generator=g) / math.sqrt(k)
    b = torch.randn(nout, generator=g)
    keep = torch.tensor([1.25, 0.0][:n])
    x, wt = rnd(x), rnd(wt)
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    pre = F.linear(xr, wr, br)
    y = F.gelu(pre) * keep.repeat_interleave(h * w).view(-1, 1)
    gy = torch.randn(y.shape, generator=g)
    gy = rnd(gy)
    y.backward(gy)
    xd = torch.zeros(n, h, w, kp, dtype=dt, device=cuda)
    xd[..., :k] = x.view(n, h, w, k).to(cuda, dt)
    nop = (nout + 31) // 32 * 32
    wd, bd, keepd = wt.to(cuda), b.to(cuda), keep.to(cuda)
    slab = torch.zeros(ops.packed_bytes(kp, nout, compute, 1), dtype=torch.uint8, device=cuda)
    slab_t = torch.zeros(ops.packed_bytes(nop, k, compute, 1), dtype=torch.uint8, device=cuda)
    plan = ops.PackPlan(cuda)
    plan.add(wd, slab, kp, nout, compute)
    plan.add(wd, slab_t, nop, k, compute, transpose=True)
    plan.run()
    out = torch.zeros(n, h, w, nop, dtype=torch.float32, device=cuda)
    pre_d = torch.zeros(n, h, w, nop, dtype=dt, device=cuda)
    ops.conv3x3(n=n, h=h, w=w, cin=kp, cout=nout, compute=compute, inp=ops.view(xd), out=ops.view(out), out2=ops.view(pre_d), wpacked=slab, bias=bd,
                act=L.ACT_GELU, ksize=1, sample_scale=keepd)
    if w == 32 and nout > 32:          # the token-GEMM tiling (64-wide output groups on a 32-wide token grid)
        want = "linear_dma" if (linear_impl == 0 and compute_name != "f32" and kp >= 64) else "linear"
        assert ops.last_conv_family() == want, (ops.last_conv_family(), want)
    assert rel_l2(out[..., :nout].reshape(-1, nout).cpu(), y.detach()) < 3e-6
    assert nop == nout or out[..., nout:].abs().max().item() == 0
    tolp = 3e-6 if compute == L.F32 else 3e-3
    assert rel_l2(pre_d[..., :nout].float().reshape(-1, nout).cpu(), pre.detach()) < tolp
    # input gradient: dX = (gy * keep * gelu'(pre)) @ W ; the mask/keep are applied by a preceding axpby-like op in the net, here on the host
    gpre = (gy * keep.repeat_interleave(h * w).view(-1, 1) * torch.autograd.grad(F.gelu(pre.detach().requires_grad_(True)).sum(),
                                                                                  [])[0] if False else None)
    pre_det = pre.detach().clone().requires_grad_(True)
    F.gelu(pre_det).sum().backward()
    gpre = gy * keep.repeat_interleave(h * w).view(-1, 1) * pre_det.grad
    gpre = rnd(gpre)
    gpd = torch.zeros(n, h, w, nop, dtype=dt, device=cuda)
    gpd[..., :nout] = gpre.view(n, h, w, nout).to(cuda, dt)
    gx = torch.zeros(n, h, w, kp, dtype=torch.float32, device=cuda)
    ops.conv3x3(n=n, h=h, w=w, cin=nop, cout=k, compute=compute, inp=ops.view(gpd), out=ops.view(gx), wpacked=slab_t, ksize=1)
    ref_gx = gpre @ wt
    assert rel_l2(gx[..., :k].reshape(-1, k).cpu(), ref_gx) < 3e-6
    # weight / bias gradient
    ws = torch.empty(ops.wgrad_workspace_bytes(n, h, w, k, nout, 1) // 4 + 16, dtype=torch.float32, device=cuda)
    dw = torch.full((nout, k), 3.0, device=cuda)
    db = torch.full((nout,), 3.0, device=cuda)
    ops.wgrad3x3(n=n, h=h, w=w, cin=k, cout=nout, compute=compute, inp=ops.view(xd), gy=ops.view(gpd), dw=dw, db=db, workspace=ws, ksize=1)
    assert rel_l2(dw.cpu(), gpre.t() @ x) < 5e-6
    assert rel_l2(db.cpu(), gpre.sum(0)) < 5e-6
    # GELU-derivative mask in the epilogue (mask_mode 1): gx2 = (gy @ W2^T-like) * gelu'(mask)
    mk = torch.randn(n, h, w, kp, generator=g)
    mkd = mk.to(cuda, dt)
    gx2 = torch.zeros_like(gx)
    ops.conv3x3(n=n, h=h, w=w, cin=nop, cout=k, compute=compute, inp=ops.view(gpd), out=ops.view(gx2), wpacked=slab_t, ksize=1,
                mask=ops.view(mkd, 0, kp), mask_mode=1)
    mref = mkd.float().cpu()[..., :k].reshape(-1, k).requires_grad_(True)
    F.gelu(mref).sum().backward()
    assert rel_l2(gx2[..., :k].reshape(-1, k).cpu(), ref_gx * mref.grad) < 3e-6


@pytest.mark.parametrize("shape", [(2, 19, 37, 64, 64), (1, 16, 48, 8, 64), (1, 33, 40, 128, 32), (1, 12, 32, 256, 128)])
def test_conv3x3_split_bf16_fast_matmul(cuda, shape):
    """compute = TR_BF16X3 (the reference's `fast_matmul` trade, done without TF32): fp32 views, every product as hi*hi + hi*lo + lo*hi in
    bf16 MFMAs.  Against the fp32 reference: rel-L2 <= 2e-5 (a bf16 conv of the same data sits at ~3e-3, TF32 at ~5e-4)."""
    from trainner_redux_amd import _lib as L, ops

    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(77 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9)
    b = torch.randn(cout, generator=g)
    ref = F.relu(F.conv2d(x, wt, b, padding=1))
    xd, bd = nhwc(x).to(cuda), b.to(cuda)
    wp = pack(ops, L, wt.to(cuda), cin, cout, L.BF16X3, cuda)
    out = torch.empty(n, h, w, cout, dtype=torch.float32, device=cuda)
    ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.BF16X3, inp=ops.view(xd), out=ops.view(out), wpacked=wp, bias=bd, act=L.ACT_RELU)
    e = rel_l2(nchw(out.cpu()), ref)
    e_bf = rel_l2(F.relu(F.conv2d(bf(x), bf(wt), b, padding=1)), ref)
    print(f"split-bf16 {shape}: rel-L2 {e:.2e} (plain bf16 operands: {e_bf:.2e})")
    assert e < 2e-5, e
    # input gradient through the transposed slab in the same mode
    gy = torch.randn(n, cout, h, w, generator=g)
    wpt = pack(ops, L, wt.to(cuda), cout, cin, L.BF16X3, cuda, transpose=True)
    gyd = nhwc(gy).to(cuda)
    gx = torch.empty(n, h, w, cin, dtype=torch.float32, device=cuda)
    ops.conv3x3(n=n, h=h, w=w, cin=cout, cout=cin, compute=L.BF16X3, inp=ops.view(gyd), out=ops.view(gx), wpacked=wpt)
    assert rel_l2(nchw(gx.cpu()), torch.nn.grad.conv2d_input(x.shape, wt, gy, padding=1)) < 2e-5


@pytest.mark.parametrize("shape", [(1, 16, 32, 64, 64), (2, 20, 36, 64, 128), (1, 8, 8, 128, 64), (1, 44, 68, 40, 64), (3, 4, 4, 256, 64), (3, 176, 352, 64, 64)])
def test_winograd43_f32_forward_and_input_gradient(cuda, shape):
    """tr_conv3x3_winograd43_f32 (fused Winograd F(4x4, 3x3), fp32 MFMA): forward with bias + ReLU into a channel slice, and the input-gradient
    form (transposed slab + ReLU-derivative mask), against torch in fp64.  fp32 rounding of the 6 x 6 transforms: ~2e-6 rel-L2 per layer
    (measured 1e-6 ... 2.4e-6; the F(2x2, 3x3) kernel: 3e-7), ragged sizes (multiples of 4) included."""
    import math

    from trainner_redux_amd import _lib as L, ops

    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(23)
    x = torch.relu(torch.randn(n, cin, h, w, generator=g, dtype=torch.float64))
    wt = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) * math.sqrt(2 / (cin * 9))
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    ref = torch.relu(F.conv2d(x, wt, b, padding=1))
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(cuda)
    wd = wt.float().to(cuda)
    out = torch.full((n, h, w, cout + 4), 3.0, device=cuda)
    U = ops.pack_winograd43(wd)
    ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD43, inp=ops.view(xd), out=ops.view(out, 4, cout), wpacked=U, bias=b.float().to(cuda),
                act=L.ACT_RELU)
    assert rel_l2(out[..., 4:].permute(0, 3, 1, 2).cpu(), ref) < 5e-6
    assert torch.all(out[..., :4] == 3.0)
    # pooled second output (conv -> ReLU -> MaxPool pairs of VGG19): pre-ReLU activation stored, max_pool2d(relu(.)) next to it, in a channel slice
    pre = torch.full((n, h, w, cout), 3.0, device=cuda)
    pooled = torch.full((n, h // 2, w // 2, cout + 4), 5.0, device=cuda)
    ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD43, inp=ops.view(xd), out=ops.view(pre), wpacked=U, bias=b.float().to(cuda),
                out2=ops.view(pooled, 4, cout))
    assert rel_l2(pre.permute(0, 3, 1, 2).cpu(), F.conv2d(x, wt, b, padding=1)) < 5e-6
    assert rel_l2(pooled[..., 4:].permute(0, 3, 1, 2).cpu(), F.max_pool2d(ref, 2, 2)) < 5e-6
    assert torch.all(pooled[..., :4] == 5.0)
    if cin % 64 == 0:
        gy = torch.randn(n, cout, h, w, generator=g, dtype=torch.float64)
        m = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64)
        refg = F.conv_transpose2d(gy, wt, padding=1) * (m > 0)
        Ut = ops.pack_winograd43(wd, transpose=True)
        gyd = gy.float().permute(0, 2, 3, 1).contiguous().to(cuda)
        md = m.float().permute(0, 2, 3, 1).contiguous().to(cuda)
        gin = torch.empty(n, h, w, cin, device=cuda)
        ops.conv3x3(n=n, h=h, w=w, cin=cout, cout=cin, compute=L.F32_WINOGRAD43, inp=ops.view(gyd), out=ops.view(gin), wpacked=Ut, mask=ops.view(md), mask_slope=0.0)
        assert rel_l2(gin.permute(0, 3, 1, 2).cpu(), refg) < 5e-6


@pytest.mark.parametrize("linear_impl", [0, 1])
@pytest.mark.parametrize("compute_name", ["bf16", "f16"])
@pytest.mark.parametrize("dims", [(2, 16, 192, 180), (1, 9, 192, 540), (2, 8, 384, 180), (1, 12, 192, 360), (2, 6, 64, 60), (1, 8, 64, 120)])
def test_linear_straight_line_epilogue_forms(cuda, compute_name, dims, linear_impl):
    """The five compile-time epilogue forms of the token GEMM (conv3x3.hip `lin_epilogue_fast`; what SwinIR's Linears use, drct_arch.py:52-75,166-223):
    plain -> 16-bit, plain -> fp32, bias + keep * v + fp32 residual -> fp32 (+ 16-bit copy), bias + exact GELU -> 16-bit with the pre-activation as second
    output, GELU-derivative mask -> 16-bit.  Against torch on the CPU and against the generic epilogue (`linear_epilogue` = 1) on the same operands."""
    from trainner_redux_amd import _lib as L, ops

    compute, dt, rnd = ct(L, compute_name)
    n, h, kp, nout = dims
    w = 32
    k = kp - 12 if kp > 64 else kp - 4
    nop = (nout + 31) // 32 * 32
    g = torch.Generator().manual_seed(kp + nout + h)
    x = rnd(torch.randn(n * h * w, k, generator=g))
    wt = rnd(torch.randn(nout, k, generator=g) / math.sqrt(k))
    b = torch.randn(nout, generator=g)
    keep = torch.tensor([1.25, 0.0][:n])
    res = torch.randn(n * h * w, nout, generator=g)
    mk = rnd(torch.randn(n * h * w, nout, generator=g))
    lin = F.linear(x, wt, b)
    lin_nb = F.linear(x, wt)
    kk = keep.repeat_interleave(h * w).view(-1, 1)
    mref = mk.clone().requires_grad_(True)
    F.gelu(mref).sum().backward()
    refs = {"plain_cd": lin, "plain_f32": lin_nb, "res_f32": lin * kk * 1.0 + res, "gelu_cd": F.gelu(lin), "maskg_cd": lin_nb * mref.grad}
    xd = torch.zeros(n, h, w, kp, dtype=dt, device=cuda)
    xd[..., :k] = x.view(n, h, w, k).to(cuda, dt)
    slab = torch.zeros(ops.packed_bytes(kp, nout, compute, 1), dtype=torch.uint8, device=cuda)
    plan = ops.PackPlan(cuda)
    plan.add(wt.to(cuda), slab, kp, nout, compute)
    plan.run()
    bd, keepd = b.to(cuda), keep.to(cuda)
    resd = torch.zeros(n, h, w, nop, device=cuda)
    resd[..., :nout] = res.view(n, h, w, nout).to(cuda)
    mkd = torch.zeros(n, h, w, nop, dtype=dt, device=cuda)
    mkd[..., :nout] = mk.view(n, h, w, nout).to(cuda, dt)

    def run(form, generic):
        out16 = torch.full((n, h, w, nop), 3.0, dtype=dt, device=cuda)
        out32 = torch.full((n, h, w, nop), 3.0, dtype=torch.float32, device=cuda)
        o2 = torch.full((n, h, w, nop), 5.0, dtype=dt, device=cuda)
        K = dict(n=n, h=h, w=w, cin=kp, cout=nout, compute=compute, inp=ops.view(xd), wpacked=slab, ksize=1)
        with ops.tuned(linear_impl=linear_impl, linear_epilogue=int(generic)):
            if form == "plain_cd":
                ops.conv3x3(out=ops.view(out16), bias=bd, **K)
            elif form == "plain_f32":
                ops.conv3x3(out=ops.view(out32), **K)
            elif form == "res_f32":
                ops.conv3x3(out=ops.view(out32), out2=ops.view(o2), out2_scale=0.5, bias=bd, res1=ops.view(resd), sample_scale=keepd, **K)
            elif form == "gelu_cd":
                ops.conv3x3(out=ops.view(out16), out2=ops.view(o2), bias=bd, act=L.ACT_GELU, **K)
            else:
                ops.conv3x3(out=ops.view(out16), mask=ops.view(mkd), mask_mode=1, **K)
        assert ops.last_conv_family() == ("linear_dma" if (linear_impl == 0 and kp >= 64) else "linear")
        return out16.float().cpu(), out32.cpu(), o2.float().cpu()

    for form, ref in refs.items():
        f16o, f32o, f2 = run(form, generic=False)
        g16o, g32o, g2 = run(form, generic=True)
        ref = ref.detach().view(n, h, w, nout)
        if form in ("plain_f32", "res_f32"):
            got, gen = f32o, g32o
            assert rel_l2(got[..., :nout], ref) < 3e-6, (form, rel_l2(got[..., :nout], ref))
        else:
            got, gen = f16o, g16o
            e = rel_l2(got[..., :nout], rnd(ref))
            assert e < 1e-3 and (got[..., :nout] == rnd(ref)).float().mean() > 0.98, (form, e)
        assert torch.equal(got[..., nout:], gen[..., nout:])                 # the padding channels are left alone by both
        assert rel_l2(got[..., :nout], gen[..., :nout]) < (3e-6 if got is f32o else 1e-3), form
        if form == "res_f32":
            assert rel_l2(f2[..., :nout], rnd(ref * 0.5)) < 1e-3 and torch.equal(f2[..., nout:], g2[..., nout:])
        if form == "gelu_cd":
            assert rel_l2(f2[..., :nout], rnd(lin.view(n, h, w, nout))) < 1e-3        # the pre-activation the backward's GELU derivative is taken from
