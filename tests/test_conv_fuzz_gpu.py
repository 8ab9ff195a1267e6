"""Seeded random sweep of `tr_conv3x3` against `F.conv2d` (CPU, fp32, on the same rounded operands) under the AUTOMATIC kernel selection -- whatever family
(register-staged, LDS-DMA ring with its fast / generic epilogue forms, token GEMM) the library picks for a shape must agree with the plain convolution.

Why: every hand-written case list so far shared the hot path's habits (cout a multiple of the output-group width, slices of 192-channel buffers).  The
round-3 bug -- the 64-wide ring instances stored the absent second tile of a half-filled last group into the next pixel when cout = 64 k + 32 -- sat in
exactly the shapes nobody had listed.  This sweep draws shapes nobody chose: channel counts 8 .. 256 in steps of 8, ragged h / w, input and output views
that are slices of wider buffers at random offsets (the bytes around the output slice must survive), and each epilogue form of the header's contract
(include/trainner_hip.h: bias, activation, per-sample scale, one or two residuals, sign mask, second output).

Tolerances as tests/test_conv_ring_gpu.py: fp32 outputs rel-L2 <= 3e-6; 16-bit outputs <= 1e-3 against the reference rounded the same way with >= 99 % of the
elements identical."""
import math
import random

import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu

CH = [8, 16, 24, 32, 40, 48, 64, 72, 96, 104, 128, 160, 192, 224, 256]
FORMS = ["plain", "bias_lrelu", "relu", "mask", "res1", "res2_out2", "scale_res1", "mask_out2"]


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _case(seed):
    r = random.Random(seed)
    compute = r.choice(["bf16", "bf16", "f16", "f32"])
    n, h, w = r.randint(1, 3), r.randint(3, 44), r.randint(3, 76)
    cin, cout = r.choice(CH), r.choice(CH)
    if compute == "f32":                                  # fp32 MFMA is 16x slower and the CPU reference dominates: keep the products small
        cin, cout = min(cin, 96), min(cout, 96)
    form = r.choice(FORMS)
    in_slack, out_slack = r.choice([0, 0, 8, 32]), r.choice([0, 0, 8, 32, 64])
    in_off = r.choice([0, 8]) if in_slack >= 8 else 0
    out_off = r.choice([0, 8, 16, 32]) if out_slack >= 32 else (r.choice([0, 8]) if out_slack >= 8 else 0)
    out_off = min(out_off, out_slack)
    return dict(seed=seed, compute=compute, n=n, h=h, w=w, cin=cin, cout=cout, form=form, in_slack=in_slack, in_off=in_off, out_slack=out_slack, out_off=out_off)


@pytest.mark.parametrize("seed", range(160))
def test_conv3x3_random_shape_matches_conv2d(cuda, seed):
    from trainner_redux_amd import _lib as L, ops

    c = _case(seed)
    compute, dt = {"bf16": (L.BF16, torch.bfloat16), "f16": (L.F16, torch.float16), "f32": (L.F32, torch.float32)}[c["compute"]]
    rnd = (lambda t: t.to(dt).to(torch.float32))
    n, h, w, cin, cout, form = c["n"], c["h"], c["w"], c["cin"], c["cout"], c["form"]
    g = torch.Generator().manual_seed(1000 + seed)
    x = rnd(torch.randn(n, cin, h, w, generator=g))
    wt = rnd(torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    b = torch.randn(cout, generator=g)
    pre = rnd(torch.randn(n, cout, h, w, generator=g))
    r1, r2 = torch.randn(n, cout, h, w, generator=g), torch.randn(n, cout, h, w, generator=g)
    ss = torch.rand(n, generator=g) + 0.5

    v = F.conv2d(x, wt, None if form in ("plain", "mask", "mask_out2") else b, padding=1)
    kw, out32, second = {}, False, False
    if form == "bias_lrelu":
        v = F.leaky_relu(v, 0.2); kw = dict(act=L.ACT_LRELU, slope=0.2)
    elif form == "relu":
        v = F.relu(v); kw = dict(act=L.ACT_RELU)
    elif form in ("mask", "mask_out2"):
        v = v * torch.where(pre > 0, 1.0, 0.2); second = form == "mask_out2"
    elif form == "res1":
        v = v * 0.2 + r1; out32 = True
    elif form == "res2_out2":
        v = (v * 0.2 + r1) * 0.2 + 0.7 * r2; out32, second = True, True
    elif form == "scale_res1":
        v = v * ss.view(n, 1, 1, 1) + r1; out32 = True
    ref = v

    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()        # noqa: E731
    xin = torch.randn(n, h, w, cin + c["in_slack"], generator=g)
    xin[..., c["in_off"]:c["in_off"] + cin] = nhwc(x)
    xd = xin.to(cuda, dt)
    wbuf = torch.zeros(ops.packed_bytes(cin, cout, compute), dtype=torch.uint8, device=cuda)
    plan = ops.PackPlan(cuda); plan.add(wt.to(cuda), wbuf, cin, cout, compute); plan.run()
    odt = torch.float32 if (out32 or compute == L.F32) else dt
    out = torch.full((n, h, w, cout + c["out_slack"]), 3.0, dtype=odt, device=cuda)
    out2 = torch.full((n, h, w, cout + 8), 5.0, dtype=dt, device=cuda)
    bias = None if form in ("plain", "mask", "mask_out2") else b.to(cuda)
    if form in ("mask", "mask_out2"):
        kw = dict(mask=ops.view(nhwc(pre).to(cuda, dt)), mask_slope=0.2)
    if form in ("res1", "res2_out2", "scale_res1"):
        r1d = nhwc(r1).to(cuda)
        kw = dict(res1=ops.view(r1d), scale1=1.0 if form == "scale_res1" else 0.2)
    if form == "res2_out2":
        r2d = nhwc(r2).to(cuda)
        kw.update(res2=ops.view(r2d), scale2=0.2, res2_scale=0.7)
    if form == "scale_res1":
        kw.update(sample_scale=ss.to(cuda))
    if second:
        kw.update(out2=ops.view(out2, 0, cout), out2_scale=0.5)
    keep = [v_ for v_ in kw.values() if not isinstance(v_, (int, float))]      # views hold raw pointers: keep their tensors alive   # noqa: F841
    ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=compute, inp=ops.view(xd, c["in_off"], cin), out=ops.view(out, c["out_off"], cout), wpacked=wbuf,
                bias=bias, **kw)
    fam, inst = ops.last_conv_family(), ops.tune_get("last_conv_instance")
    got = out.float().cpu()
    o0 = c["out_off"]
    what = (c, fam, inst)

    def check(got_nchw, ref_t, dtype_out):
        if dtype_out == torch.float32:
            e = rel_l2(got_nchw, ref_t)
            assert e < (3e-6 if compute != L.F32 else 3e-6), (what, e)
        else:
            refr = ref_t.to(dtype_out).to(torch.float32)
            e, same = rel_l2(got_nchw, refr), (got_nchw == refr).float().mean().item()
            assert e < 1e-3 and same > 0.99, (what, e, same)

    check(got[..., o0:o0 + cout].permute(0, 3, 1, 2), ref, odt)
    assert torch.all(got[..., :o0] == 3.0) and torch.all(got[..., o0 + cout:] == 3.0), (what, "bytes around the output slice were written")
    if second:
        g2 = out2.float().cpu()
        check(g2[..., :cout].permute(0, 3, 1, 2), ref * 0.5, dt)
        assert torch.all(g2[..., cout:] == 5.0), (what, "bytes behind the second output were written")


LIN_FORMS = ["plain_cd", "plain_f32", "res_f32", "gelu_cd", "gelu_mask_cd", "scale_res_f32"]


@pytest.mark.parametrize("seed", range(60))
def test_token_gemm_random_shape_matches_linear(cuda, seed):
    """ksize 1 on a 32-wide token grid (nn.Linear over tokens, drct_arch.py:64-66,166-168): the LDS-DMA token GEMM with its straight-line epilogue forms,
    the register-staged one and the conv-tiled fallback, whichever the library picks, against `x @ W^T + b` (+ exact-erf GELU / GELU' / residual / DropPath scale)."""
    from trainner_redux_amd import _lib as L, ops

    r = random.Random(5000 + seed)
    cname = r.choice(["bf16", "bf16", "f16", "f32"])
    compute, dt = {"bf16": (L.BF16, torch.bfloat16), "f16": (L.F16, torch.float16), "f32": (L.F32, torch.float32)}[cname]
    rnd = (lambda t: t.to(dt).to(torch.float32))
    n, h, w = r.randint(1, 3), r.randint(1, 40), 32
    cin = r.choice([32, 64, 96, 128, 160, 192, 256, 384, 544])
    cout = r.choice([32, 48, 60, 64, 96, 120, 180, 192, 256, 360, 540])
    form = r.choice(LIN_FORMS)
    cs_in = cin + r.choice([0, 0, 32])
    cop = (cout + 31) // 32 * 32 + r.choice([0, 32])
    g = torch.Generator().manual_seed(7000 + seed)
    x = rnd(torch.randn(n, h, w, cin, generator=g))
    wt = rnd(torch.randn(cout, cin, generator=g) / math.sqrt(cin))
    b = torch.randn(cout, generator=g)
    pre_m = rnd(torch.randn(n, h, w, cout, generator=g))
    res = torch.randn(n, h, w, cout, generator=g)
    ss = torch.rand(n, generator=g) + 0.5
    v = x @ wt.t() + b
    pre = v
    out32 = form in ("plain_f32", "res_f32", "scale_res_f32") or compute == L.F32
    kw = {}
    if form == "res_f32":
        v = v + res
    elif form == "scale_res_f32":
        v = v * ss.view(n, 1, 1, 1) + res
    elif form == "gelu_cd":
        v = F.gelu(v)
    elif form == "gelu_mask_cd":
        m = pre_m.double()
        v = v * (0.5 * (1 + torch.erf(m / math.sqrt(2))) + m * torch.exp(-0.5 * m * m) / math.sqrt(2 * math.pi)).float()
    xin = torch.randn(n, h, w, cs_in, generator=g)
    xin[..., :cin] = x
    xd = xin.to(cuda, dt)
    wbuf = torch.zeros(ops.packed_bytes(cin, cout, compute, 1), dtype=torch.uint8, device=cuda)
    plan = ops.PackPlan(cuda); plan.add(wt.to(cuda), wbuf, cin, cout, compute); plan.run()
    odt = torch.float32 if out32 else dt
    out = torch.full((n, h, w, cop), 3.0, dtype=odt, device=cuda)
    out2 = torch.full((n, h, w, cop), 5.0, dtype=dt, device=cuda)
    resd, md, ssd = res.to(cuda), pre_m.to(cuda, dt), ss.to(cuda)
    if form in ("res_f32", "scale_res_f32"):
        kw.update(res1=ops.view(resd))
    if form == "scale_res_f32":
        kw.update(sample_scale=ssd)
    if form == "gelu_cd":
        kw.update(act=L.ACT_GELU, out2=ops.view(out2, 0, cout))
    if form == "gelu_mask_cd":
        kw.update(mask=ops.view(md), mask_mode=1)
    ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=compute, inp=ops.view(xd, 0, cin), out=ops.view(out, 0, cout), wpacked=wbuf, bias=b.to(cuda), ksize=1, **kw)
    what = (seed, cname, n, h, cin, cout, form, ops.last_conv_family())
    got = out.float().cpu()

    def check(got_t, ref_t, dtype_out):
        if dtype_out == torch.float32:
            e = rel_l2(got_t, ref_t)
            assert e < 3e-6, (what, e)
        else:
            refr = ref_t.to(dtype_out).to(torch.float32)
            e, same = rel_l2(got_t, refr), (got_t == refr).float().mean().item()
            assert e < 1e-3 and same > 0.985, (what, e, same)        # the 16-bit forms' GELU is the rcp / exp form (4e-7 absolute): a few more 1-ulp flips

    check(got[..., :cout], v, odt)
    assert torch.all(got[..., cout:] == 3.0) or torch.all(got[..., (cout + 3) // 4 * 4:] == 3.0), (what, "bytes behind the output were written")
    if form == "gelu_cd":
        check(out2.float().cpu()[..., :cout], pre, dt)


@pytest.mark.parametrize("seed", range(70))
def test_wgrad_random_shape_matches_conv2d_weight(cuda, seed):
    """`tr_wgrad3x3` (3x3 and ksize 1) on random shapes and channel slices under the automatic selection (register-staged, ring, split-K token kernel; 4- or
    16-byte second stage): dW against `torch.nn.grad.conv2d_weight`, db against the plain sum, written once and accumulated once."""
    from trainner_redux_amd import _lib as L, ops

    r = random.Random(9000 + seed)
    cname = r.choice(["bf16", "bf16", "f16", "f32"])
    compute, dt = {"bf16": (L.BF16, torch.bfloat16), "f16": (L.F16, torch.float16), "f32": (L.F32, torch.float32)}[cname]
    rnd = (lambda t: t.to(dt).to(torch.float32))
    ks = r.choice([3, 3, 1])
    n = r.randint(1, 3)
    h, w = (r.randint(1, 24), 32) if ks == 1 else (r.randint(3, 36), r.randint(3, 70))
    cin, cout = r.choice(CH + [180, 360]), r.choice(CH + [180, 544])
    if cname == "f32":
        cin, cout = min(cin, 96), min(cout, 96)
    if ks == 1 and cin == 180:
        cin = 192
    g = torch.Generator().manual_seed(11000 + seed)
    x = rnd(torch.randn(n, cin, h, w, generator=g))
    gy = rnd(torch.randn(n, cout, h, w, generator=g))
    ref_w = torch.nn.grad.conv2d_weight(x, (cout, cin, ks, ks), gy, padding=ks // 2)
    ref_b = gy.sum(dim=(0, 2, 3))
    si, so = r.choice([0, 8, 32]), r.choice([0, 8, 32])
    oi, oo = (r.choice([0, 8]) if si else 0), (r.choice([0, 8]) if so else 0)
    cpi, cpo = (cin + 7) // 8 * 8, (cout + 7) // 8 * 8
    xin = torch.randn(n, h, w, cpi + si, generator=g); xin[..., oi:oi + cin] = x.permute(0, 2, 3, 1)
    gin = torch.randn(n, h, w, cpo + so, generator=g); gin[..., oo:oo + cout] = gy.permute(0, 2, 3, 1)
    if cpi != cin:
        xin[..., oi + cin:oi + cpi] = 0                   # the view is 8-channel granular: the pad channels belong to it
    if cpo != cout:
        gin[..., oo + cout:oo + cpo] = 0
    xd, gd = xin.to(cuda, dt), gin.to(cuda, dt)
    ws = torch.empty(ops.wgrad_workspace_bytes(n, h, w, cin, cout, ks) // 4 + 16, dtype=torch.float32, device=cuda)
    dw = torch.full((cout, cin, ks, ks) if ks == 3 else (cout, cin), 7.0, device=cuda)
    db = torch.full((cout,), 7.0, device=cuda)
    kw = dict(n=n, h=h, w=w, cin=cin, cout=cout, compute=compute, inp=ops.view(xd, oi, cpi), gy=ops.view(gd, oo, cpo), dw=dw, db=db, workspace=ws, ksize=ks)
    ops.wgrad3x3(**kw)
    w1, b1 = dw.cpu().clone(), db.cpu().clone()
    ops.wgrad3x3(**kw, accumulate=True)
    what = (seed, cname, ks, n, h, w, cin, cout, si, oi, so, oo)
    assert rel_l2(w1.view_as(ref_w), ref_w) < 4e-6, (what, rel_l2(w1.view_as(ref_w), ref_w))
    assert rel_l2(b1, ref_b) < 4e-6, (what, rel_l2(b1, ref_b))
    assert rel_l2(dw.cpu().view_as(ref_w), 2 * ref_w) < 4e-6 and rel_l2(db.cpu(), 2 * ref_b) < 4e-6, what


@pytest.mark.parametrize("seed", range(36))
def test_conv3x3_many_tiles_per_workgroup_ring_vs_register_staged(cuda, seed):
    """The sweeps above run a tile or two per workgroup.  The persistent kernels' cross-tile machinery (the halo stream running into the next tile during
    the epilogue, LDS strips aliasing a consumed ring stage, pipelined epilogue units hooked into the next tile's MFMAs) only runs when a workgroup walks
    SEVERAL tiles: 2^18 .. 2^19 pixels here (2 .. 8 tiles per workgroup).  No CPU reference at that size: the ring kernel is compared with the
    register-staged kernel, which the small-shape sweeps pin to F.conv2d -- same products, same K order, so fp32 outputs agree to 2e-6 and 16-bit
    outputs are identical on >= 99.8 % of the elements.  Ragged right / bottom edges, random channel slices, every epilogue form."""
    from trainner_redux_amd import _lib as L, ops

    r = random.Random(13000 + seed)
    cname = r.choice(["bf16", "f16"])
    compute, dt = (L.BF16, torch.bfloat16) if cname == "bf16" else (L.F16, torch.float16)
    n = r.choice([2, 4, 8])
    h = r.choice([128, 200, 250, 256]); w = (1 << r.choice([18, 19])) // (n * h) + r.choice([0, 3, 17])
    cin = r.choice([32, 64, 96, 128, 160, 192]); cout = r.choice([32, 64, 96, 128, 160, 48, 24])
    form = r.choice([f for f in FORMS if f != "scale_res1"])          # a per-sample scale is served by the register-staged kernel only
    g = torch.Generator(device="cuda").manual_seed(17000 + seed)
    x = torch.randn(n, h, w, cin + 32, device=cuda, generator=g).to(dt)
    wt = (torch.randn(cout, cin, 3, 3, device=cuda, generator=g) / math.sqrt(cin * 9))
    b = torch.randn(cout, device=cuda, generator=g)
    pre = torch.randn(n, h, w, cout, device=cuda, generator=g).to(dt)
    r1, r2 = torch.randn(n, h, w, cout, device=cuda, generator=g), torch.randn(n, h, w, cout, device=cuda, generator=g)
    ss = torch.rand(n, device=cuda, generator=g) + 0.5
    wbuf = torch.zeros(ops.packed_bytes(cin, cout, compute), dtype=torch.uint8, device=cuda)
    plan = ops.PackPlan(cuda); plan.add(wt, wbuf, cin, cout, compute); plan.run()
    out32 = form in ("res1", "res2_out2", "scale_res1")
    second = form in ("res2_out2", "mask_out2")
    kw = {}
    if form == "bias_lrelu": kw = dict(act=L.ACT_LRELU, slope=0.2)
    if form == "relu": kw = dict(act=L.ACT_RELU)
    if form in ("mask", "mask_out2"): kw = dict(mask=ops.view(pre), mask_slope=0.2)
    if out32: kw = dict(res1=ops.view(r1), scale1=1.0 if form == "scale_res1" else 0.2)
    if form == "res2_out2": kw.update(res2=ops.view(r2), scale2=0.2, res2_scale=0.7)
    if form == "scale_res1": kw.update(sample_scale=ss)
    bias = None if form in ("plain", "mask", "mask_out2") else b
    res = {}
    for impl in (1, 2):
        out = torch.full((n, h, w, cout + 32), 3.0, dtype=torch.float32 if out32 else dt, device=cuda)
        out2 = torch.full((n, h, w, cout + 8), 5.0, dtype=dt, device=cuda)
        k2 = dict(kw, out2=ops.view(out2, 0, cout), out2_scale=0.5) if second else kw
        with ops.tuned(conv_impl=impl):
            ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=compute, inp=ops.view(x, 8, cin), out=ops.view(out, 16, cout), wpacked=wbuf, bias=bias, **k2)
        res[impl] = (out, out2, ops.last_conv_family(), ops.tune_get("last_conv_instance"))
    (o1, p1, f1, _), (o2, p2, f2, inst) = res[1], res[2]
    what = (seed, cname, n, h, w, cin, cout, form, f1, f2, inst)
    assert f1 == "reg", what
    if f2 != "ring":
        pytest.skip(f"not ring-eligible: {what}")
    assert torch.all(o2[..., :16] == 3.0) and torch.all(o2[..., 16 + cout:] == 3.0) and torch.all(p2[..., cout:] == 5.0), (what, "bytes around the outputs were written")
    a, bb = o2[..., 16:16 + cout].float(), o1[..., 16:16 + cout].float()
    if out32:
        assert rel_l2(a, bb) < 2e-6, (what, rel_l2(a, bb))
    else:
        same = (a == bb).float().mean().item()
        assert same > 0.998 and rel_l2(a, bb) < 1e-3, (what, same, rel_l2(a, bb))
    if second:
        same2 = (p2[..., :cout] == p1[..., :cout]).float().mean().item()
        assert same2 > 0.998, (what, same2)


@pytest.mark.parametrize("seed", range(40))
def test_layernorm_random_shape_matches_torch(cuda, seed):
    """`tr_layernorm` / `tr_layernorm_bwd` (16-byte and 4-byte kernels, whichever the shape allows) on random token counts (not multiples of the 8 tokens a wave
    takes), channel counts 4 .. 256, padded channel strides, 16-bit or fp32 gradient / output views, with and without the folded residual."""
    from trainner_redux_amd import ops

    r = random.Random(21000 + seed)
    c = r.choice([4, 8, 20, 36, 48, 60, 64, 90, 96, 128, 180, 192, 200, 256, 30, 45])
    cp = c + r.choice([0, 0, 4, 12, 32]) if c % 4 == 0 else c + r.choice([0, 3])
    ntok = r.choice([1, 5, 8, 31, 64, 257, 1000, 4099, 20000])
    ydt = r.choice([torch.float32, torch.bfloat16, torch.float16])
    gdt = r.choice([torch.float32, torch.bfloat16])
    with_res = r.random() < 0.6
    eps = r.choice([1e-5, 1e-6])
    g = torch.Generator().manual_seed(23000 + seed)
    x = torch.randn(ntok, c, generator=g) * 2 + 0.5
    gam, bet = 1 + 0.1 * torch.randn(c, generator=g), 0.1 * torch.randn(c, generator=g)
    gy = torch.randn(ntok, c, generator=g).to(gdt).float()
    res = torch.randn(ntok, c, generator=g)
    xr, gr, br = x.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    y = F.layer_norm(xr, (c,), gr, br, eps)
    y.backward(gy)
    xd = torch.randn(ntok, cp, generator=g).to(cuda)
    xd[:, :c] = x.to(cuda)
    yd = torch.full((ntok, cp), 7.0, dtype=ydt, device=cuda)
    mean, rstd = torch.empty(ntok, device=cuda), torch.empty(ntok, device=cuda)
    gd, bd = gam.to(cuda), bet.to(cuda)
    ops.layernorm(ops.view(xd, 0, c), ops.view(yd, 0, c), ntok, c, gd, bd, eps, mean, rstd)
    what = (seed, c, cp, ntok, ydt, gdt, with_res)
    tol = 3e-6 if ydt == torch.float32 else (4e-3 if ydt == torch.bfloat16 else 6e-4)
    assert rel_l2(yd[:, :c].float().cpu(), y.detach()) < tol, (what, rel_l2(yd[:, :c].float().cpu(), y.detach()))
    assert torch.all(yd[:, c:] == 7.0), (what, "bytes behind the output view were written")
    assert rel_l2(mean.cpu(), x.mean(1)) < 1e-5 or x.mean(1).abs().max() < 1e-3
    gyd = torch.zeros(ntok, cp, dtype=gdt, device=cuda)
    gyd[:, :c] = gy.to(cuda, gdt)
    resd = torch.randn(ntok, cp, generator=g).to(cuda)
    resd[:, :c] = res.to(cuda)
    dx = torch.full((ntok, cp), 9.0, device=cuda)
    dgam, dbet = torch.full((c,), 2.0, device=cuda), torch.full((c,), 2.0, device=cuda)
    ws = torch.empty(ops.layernorm_bwd_workspace_bytes(c) // 4, device=cuda)
    # second output: the compute-dtype copy of dx scaled per sample (the DropPath keep mask of the branch that reads it), drct_arch.py:572-573 backward
    mode2 = r.choice(["none", "copy", "scaled"])
    pps = r.choice([1, 7, 64, 1000])
    d2dt = r.choice([torch.bfloat16, torch.float16, torch.float32])
    dx2 = torch.full((ntok, cp), 11.0, dtype=d2dt, device=cuda)
    ssc = (torch.rand((ntok + pps - 1) // pps, generator=g) + 0.5)
    kw2 = {} if mode2 == "none" else (dict(dx2=ops.view(dx2, 0, c)) if mode2 == "copy" else dict(dx2=ops.view(dx2, 0, c), sample_scale=ssc.to(cuda), pix_per_sample=pps))
    keep_alive = kw2.get("sample_scale")      # noqa: F841
    ops.layernorm_bwd(ops.view(gyd, 0, c), ops.view(xd, 0, c), ntok, c, gd, mean, rstd, ops.view(resd, 0, c) if with_res else ops.NULL_VIEW, ops.view(dx, 0, c), dgam, dbet,
                      ws, **kw2)
    ref_dx = xr.grad + (res if with_res else 0)
    if mode2 != "none":
        ref2 = ref_dx * (ssc.repeat_interleave(pps)[:ntok].view(ntok, 1) if mode2 == "scaled" else 1.0)
        tol2 = 5e-6 if d2dt == torch.float32 else (4e-3 if d2dt == torch.bfloat16 else 6e-4)
        assert rel_l2(dx2[:, :c].float().cpu(), ref2) < tol2, (what, mode2, pps, d2dt, rel_l2(dx2[:, :c].float().cpu(), ref2))
        assert torch.all(dx2[:, c:] == 11.0), (what, "bytes behind dx2 were written")
    assert rel_l2(dx[:, :c].cpu(), ref_dx) < 5e-6, (what, rel_l2(dx[:, :c].cpu(), ref_dx))
    assert torch.all(dx[:, c:] == 9.0), (what, "bytes behind dx were written")
    assert rel_l2(dgam.cpu() - 2.0, gr.grad) < 2e-5 and rel_l2(dbet.cpu() - 2.0, br.grad) < 2e-5, (what, rel_l2(dgam.cpu() - 2.0, gr.grad), rel_l2(dbet.cpu() - 2.0, br.grad))


@pytest.mark.parametrize("seed", range(24))
def test_winograd43_random_shape_matches_conv2d(cuda, seed):
    """The fp32 Winograd F(4x4,3x3) kernel of the VGG head on random eligible shapes (h, w multiples of 4, cin > 32 in steps of 8, cout in steps of 64), several
    tiles per persistent workgroup in some, all three epilogue forms (bias + ReLU, bias + ReLU + fused 2x2 max-pool output, input-gradient form with the
    ReLU-derivative mask), against fp64 `F.conv2d` / `conv_transpose2d`: 1e-5 (measured 2e-6 per layer)."""
    from trainner_redux_amd import _lib as L, ops

    r = random.Random(25000 + seed)
    n = r.randint(1, 3)
    h, w = 4 * r.randint(1, 40), 4 * r.randint(1, 60)
    cin, cout = r.choice([40, 64, 72, 128, 192, 256]), r.choice([64, 128, 192, 256])
    form = r.choice(["relu", "pool", "mask"])
    if not ops.winograd43_eligible(cin, cout, h, w):
        pytest.skip("not eligible")
    g = torch.Generator().manual_seed(27000 + seed)
    x = torch.relu(torch.randn(n, cin, h, w, generator=g, dtype=torch.float64))
    wt = torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) * math.sqrt(2 / (cin * 9))
    b = torch.randn(cout, generator=g, dtype=torch.float64)
    xd = x.float().permute(0, 2, 3, 1).contiguous().to(cuda)
    what = (seed, n, h, w, cin, cout, form)
    if form in ("relu", "pool"):
        ref = torch.relu(F.conv2d(x, wt, b, padding=1))
        out = torch.full((n, h, w, cout + 4), 3.0, device=cuda)
        pooled = torch.full((n, h // 2, w // 2, cout + 4), 5.0, device=cuda)
        U = ops.pack_winograd43(wt.float().to(cuda))
        kw = dict(out2=ops.view(pooled, 0, cout)) if form == "pool" else {}
        ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD43, inp=ops.view(xd), out=ops.view(out, 4, cout), wpacked=U, bias=b.float().to(cuda),
                    act=L.ACT_RELU, **kw)
        assert rel_l2(out[..., 4:].permute(0, 3, 1, 2).cpu(), ref) < 1e-5, what
        assert torch.all(out[..., :4] == 3.0), what
        if form == "pool":
            assert rel_l2(pooled[..., :cout].permute(0, 3, 1, 2).cpu(), F.max_pool2d(ref, 2, 2)) < 1e-5 and torch.all(pooled[..., cout:] == 5.0), what
    else:
        if cin % 64:
            pytest.skip("the input-gradient form writes cin channels: 64-wide groups")
        gy = torch.randn(n, cout, h, w, generator=g, dtype=torch.float64)
        m = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64)
        ref = F.conv_transpose2d(gy, wt, padding=1) * (m > 0)
        if not ops.winograd43_eligible(cout, cin, h, w):
            pytest.skip("not eligible")
        Ut = ops.pack_winograd43(wt.float().to(cuda), transpose=True)
        gin = torch.full((n, h, w, cin), 3.0, device=cuda)
        gyd = gy.float().permute(0, 2, 3, 1).contiguous().to(cuda)
        md = m.float().permute(0, 2, 3, 1).contiguous().to(cuda)
        ops.conv3x3(n=n, h=h, w=w, cin=cout, cout=cin, compute=L.F32_WINOGRAD43, inp=ops.view(gyd), out=ops.view(gin), wpacked=Ut, mask=ops.view(md), mask_slope=0.0)
        assert rel_l2(gin.permute(0, 3, 1, 2).cpu(), ref) < 1e-5, what


@pytest.mark.parametrize("seed", range(60))
def test_conv_input_modes_random_shape(cuda, seed):
    """The input modes of `tr_conv3x3` on random shapes: nearest-2x fused into the load (nn.Upsample -> conv, scunet_aaf6aa_arch.py:371-377), replicate padding
    (VGG conv1_1, perceptual_fp16_loss.py:262-274), the 'full' correlation over an (h-2) x (w-2) input and the padding-free 'valid' form over (h+2) x (w+2)
    (SPAN's Conv3XC and its input gradient, span_arch.py:134-141,199-204), ksize 3 and 1 where the mode allows it."""
    from trainner_redux_amd import _lib as L, ops

    r = random.Random(31000 + seed)
    cname = r.choice(["bf16", "f16", "f32"])
    compute, dt = {"bf16": (L.BF16, torch.bfloat16), "f16": (L.F16, torch.float16), "f32": (L.F32, torch.float32)}[cname]
    rnd = (lambda t: t.to(dt).to(torch.float32))
    mode = r.choice(["nearest2x", "replicate", "full", "valid", "valid_k1", "full_k1"])
    n = r.randint(1, 3)
    h, w = r.randint(3, 40), r.randint(3, 70)
    if mode == "nearest2x":
        h, w = 2 * r.randint(2, 20), 2 * r.randint(2, 35)
    cin, cout = r.choice([8, 16, 32, 48, 56, 64, 96, 104, 128]), r.choice([8, 16, 32, 48, 56, 64, 96, 104, 128])
    if cname == "f32":
        cin, cout = min(cin, 64), min(cout, 64)
    ks = 1 if mode.endswith("_k1") else 3
    lrelu = r.random() < 0.5
    g = torch.Generator().manual_seed(33000 + seed)
    ih, iw = {"nearest2x": (h // 2, w // 2), "replicate": (h, w), "full": (h - 2, w - 2), "valid": (h + 2, w + 2), "valid_k1": (h + 2, w + 2), "full_k1": (h - 2, w - 2)}[mode]
    x = rnd(torch.randn(n, cin, ih, iw, generator=g))
    wt = rnd(torch.randn(cout, cin, ks, ks, generator=g) / math.sqrt(cin * ks * ks))
    b = torch.randn(cout, generator=g)
    if mode == "nearest2x":
        ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), wt, b, padding=1)
    elif mode == "replicate":
        ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), wt, b)
    elif mode == "full":
        ref = F.conv2d(x, wt, b, padding=2)
    elif mode == "valid":
        ref = F.conv2d(x, wt, b)
    elif mode == "valid_k1":
        ref = F.conv2d(x[:, :, 1:-1, 1:-1], wt, b)
    else:   # full_k1: out(y, x) = W in(y - 1, x - 1), zeros outside -- the bias lands on the border too (Conv3XC's first 1x1 over the padded grid)
        ref = F.conv2d(F.pad(x, (1, 1, 1, 1)), wt, b)
    if lrelu:
        ref = F.leaky_relu(ref, 0.2)
    xd = x.permute(0, 2, 3, 1).contiguous().to(cuda, dt)
    wbuf = torch.zeros(ops.packed_bytes(cin, cout, compute, ks), dtype=torch.uint8, device=cuda)
    plan = ops.PackPlan(cuda); plan.add(wt.to(cuda) if ks == 3 else wt.reshape(cout, cin).to(cuda), wbuf, cin, cout, compute); plan.run()
    out = torch.full((n, h, w, cout + 8), 3.0, dtype=dt, device=cuda)
    im = {"nearest2x": L.IN_NEAREST2X, "replicate": L.IN_DIRECT, "full": L.IN_FULL, "valid": L.IN_VALID, "valid_k1": L.IN_VALID, "full_k1": L.IN_FULL}[mode]
    ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=compute, inp=ops.view(xd), out=ops.view(out, 0, cout), wpacked=wbuf, bias=b.to(cuda), in_mode=im,
                pad_mode=L.PAD_REPLICATE if mode == "replicate" else L.PAD_ZEROS, ksize=ks, **(dict(act=L.ACT_LRELU, slope=0.2) if lrelu else {}))
    what = (seed, cname, mode, n, h, w, cin, cout, lrelu, ops.last_conv_family())
    got = out.float().cpu()
    assert tuple(ref.shape) == (n, cout, h, w), (what, ref.shape)
    if dt == torch.float32:
        assert rel_l2(got[..., :cout].permute(0, 3, 1, 2), ref) < 3e-6, (what, rel_l2(got[..., :cout].permute(0, 3, 1, 2), ref))
    else:
        refr = ref.to(dt).float()
        e, same = rel_l2(got[..., :cout].permute(0, 3, 1, 2), refr), (got[..., :cout].permute(0, 3, 1, 2) == refr).float().mean().item()
        assert e < 1e-3 and same > 0.99, (what, e, same)
    assert torch.all(got[..., cout:] == 3.0), (what, "bytes behind the output were written")
