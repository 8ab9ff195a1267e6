"""MoSR on the HIP kernels: the new kernels (depthwise k x k conv forward / input gradient / weight gradient, Mish gate, Mish epilogue and
derivative) against plain PyTorch fp32 on the CPU, and the whole network against the oracle (oracle/mosr.py; spandrel's module is absent, so
the oracle is pinned by the reference's parameter-count checksums only: "parity unpinned" for values, see its header).

Tolerances: fp32 compute: kernels 2e-6 rel-L2, network output <= 1e-5, every parameter gradient <= 2e-4; bf16: output <= 1e-2 vs the
bf16-policy oracle, gradient cosine >= 0.99 (end-to-end rounding-flip regime, see tests/test_esrgan_gpu.py)."""
from collections import OrderedDict

import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def _nhwc(t, dtype, cuda, cstride=None, coff=0):
    """NCHW cpu tensor -> (buffer NHWC on the GPU with `cstride` channels, view of the slice holding t)."""
    from trainner_redux_amd import ops

    n, c, h, w = t.shape
    cs = cstride or c
    buf = torch.randn(n, h, w, cs).to(dtype)            # the other channels hold garbage that must not leak
    buf[..., coff:coff + c] = t.permute(0, 2, 3, 1).to(dtype)
    buf = buf.to(cuda).contiguous()
    return buf, ops.view(buf, coff, c)


def _back(buf, coff, c):
    return buf[..., coff:coff + c].float().cpu().permute(0, 3, 1, 2)


@pytest.mark.parametrize("k,c,h,w", [(7, 64, 64, 64), (7, 24, 19, 37), (5, 8, 5, 3), (3, 36, 33, 70)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dwconv_forward_dgrad_wgrad(cuda, k, c, h, w, dtype):
    from trainner_redux_amd import ops

    g = torch.Generator().manual_seed(k * 1000 + c)
    n = 2
    x = torch.randn(n, c, h, w, generator=g).to(dtype).float()
    wt = torch.randn(c, 1, k, k, generator=g) * 0.2
    bs = torch.randn(c, generator=g)
    gy = torch.randn(n, c, h, w, generator=g).to(dtype).float()
    xr = x.clone().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    br = bs.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, br, padding=k // 2, groups=c)
    y_ref.backward(gy)
    xb, xv = _nhwc(x, dtype, cuda, cstride=c + 12, coff=8)
    yb = torch.zeros(n, h, w, c + 4, dtype=dtype, device=cuda)
    ops.dwconv(xv, ops.view(yb, 4, c), wt.to(cuda), bs.to(cuda), n, h, w, c)
    tol = 2e-6 if dtype == torch.float32 else 4e-3       # bf16: the stored output is rounded
    assert rel_l2(_back(yb, 4, c), y_ref.detach()) < tol
    assert yb[..., :4].abs().sum() == 0                  # neighbours of the slice untouched
    gb, gv = _nhwc(gy, dtype, cuda)
    dxb = torch.zeros(n, h, w, c, dtype=torch.float32, device=cuda)
    ops.dwconv(gv, ops.view(dxb), wt.to(cuda), None, n, h, w, c, flip=True)
    assert rel_l2(_back(dxb, 0, c), xr.grad) < 2e-6
    dw = torch.full((c, 1, k, k), 0.5, device=cuda)
    db = torch.full((c,), -0.25, device=cuda)
    ws = torch.empty(ops.dwconv_wgrad_workspace_bytes(n, h, w, c, k) // 4, device=cuda)
    ops.dwconv_wgrad(xv, gv, dw, db, n, h, w, c, ws, accumulate=True)
    assert rel_l2(dw.cpu() - 0.5, wr.grad) < 1e-5 and rel_l2(db.cpu() + 0.25, br.grad) < 1e-5
    dw2, db2 = torch.empty_like(dw), torch.empty_like(db)
    ops.dwconv_wgrad(xv, gv, dw2, db2, n, h, w, c, ws, accumulate=False)
    ops.dwconv_wgrad(xv, gv, dw, db, n, h, w, c, ws, accumulate=False)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)      # fixed-order reduction: bitwise reproducible


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mish_gate_and_derivative_sweeps(cuda, dtype):
    from trainner_redux_amd import _lib as L, ops

    gen = torch.Generator().manual_seed(3)
    n, h, w, ca, cb = 2, 9, 13, 8, 16
    g = (torch.randn(n, ca + cb, h, w, generator=gen) * 3).to(dtype).float()
    g[0, 0, 0, :4] = torch.tensor([25.0, -30.0, 0.0, 19.5])            # softplus threshold / tails
    a = torch.randn(n, ca, h, w, generator=gen).to(dtype).float()
    b = torch.randn(n, cb, h, w, generator=gen).to(dtype).float()
    d = torch.randn(n, ca + cb, h, w, generator=gen).to(dtype).float()
    gr, ar, br = g.clone().requires_grad_(True), a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out_ref = F.mish(gr) * torch.cat((ar, br), 1)
    out_ref.backward(d)
    Fb, gv = _nhwc(torch.cat((g, a), 1), dtype, cuda)                   # [g | a] share a buffer like fc1's output, b lives apart
    T = n * h * w
    bb, bv = _nhwc(b, dtype, cuda)
    ob = torch.zeros(n, h, w, ca + cb, dtype=torch.float32, device=cuda)
    ops.gate_mish(ops.view(Fb, 0, ca + cb), ops.view(Fb, ca + cb, ca), bv, ops.view(ob), T)
    assert rel_l2(_back(ob, 0, ca + cb), out_ref.detach()) < 2e-6
    db_, dv = _nhwc(d, dtype, cuda)
    dF = torch.zeros(n, h, w, 2 * ca + cb, dtype=torch.float32, device=cuda)
    dB = torch.zeros(n, h, w, cb, dtype=torch.float32, device=cuda)
    ops.gate_mish_bwd(dv, ops.view(Fb, 0, ca + cb), ops.view(Fb, ca + cb, ca), bv, ops.view(dF, 0, ca + cb), ops.view(dF, ca + cb, ca), ops.view(dB), T)
    assert rel_l2(_back(dF, 0, ca + cb), gr.grad) < 5e-6 and rel_l2(_back(dF, ca + cb, ca), ar.grad) < 2e-6 and rel_l2(_back(dB, 0, cb), br.grad) < 2e-6
    # act_grad: d * keep[sample] * mish'(pre)
    pre = g.clone().requires_grad_(True)
    keep = torch.tensor([0.0, 1.25])
    (F.mish(pre) * keep.view(-1, 1, 1, 1)).backward(d)
    o2 = torch.zeros(n, h, w, ca + cb, dtype=torch.float32, device=cuda)
    ops.act_grad(L.ACT_MISH, dv, ops.view(Fb, 0, ca + cb), ops.view(o2), T, sample_scale=keep.to(cuda), pix_per_sample=h * w)
    assert rel_l2(_back(o2, 0, ca + cb), pre.grad) < 5e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_epilogue_mish_and_mask(cuda, dtype):
    """conv -> Mish -> keep scale -> + residual, pre-activation side output; and the Mish derivative mask on a gradient conv."""
    from trainner_redux_amd import _lib as L, ops

    gen = torch.Generator().manual_seed(11)
    n, h, w, ci, co = 2, 11, 21, 16, 24
    cd = ops.dt_code(dtype)
    x = torch.randn(n, ci, h, w, generator=gen).to(dtype).float()
    wt = (torch.randn(co, ci, 3, 3, generator=gen) * 0.3)
    bs = torch.randn(co, generator=gen)
    res = torch.randn(n, co, h, w, generator=gen)
    keep = torch.tensor([2.0, 0.0])
    wq = wt.to(dtype).float()
    pre_ref = F.conv2d(x, wq, bs, padding=1)
    y_ref = F.mish(pre_ref) * keep.view(-1, 1, 1, 1) + res
    xb, xv = _nhwc(x, dtype, cuda)
    rb, rv = _nhwc(res, torch.float32, cuda)
    slab = torch.zeros(ops.packed_bytes(ci, co, cd), dtype=torch.uint8, device=cuda)
    plan = ops.PackPlan(cuda); plan.add(wt.to(cuda), slab, ci, co, cd); plan.run()
    yb = torch.zeros(n, h, w, co, dtype=torch.float32, device=cuda)
    pb = torch.zeros(n, h, w, co, dtype=torch.float32, device=cuda)
    ops.conv3x3(n=n, h=h, w=w, cin=ci, cout=co, compute=cd, inp=xv, out=ops.view(yb), out2=ops.view(pb), act=L.ACT_MISH, res1=rv, wpacked=slab,
                bias=bs.to(cuda), sample_scale=keep.to(cuda))
    tol = 2e-6 if dtype == torch.float32 else 1e-5
    assert rel_l2(_back(pb, 0, co), pre_ref) < tol and rel_l2(_back(yb, 0, co), y_ref) < tol
    m = torch.randn(n, co, h, w, generator=gen) * 3
    mb, mv = _nhwc(m, torch.float32, cuda)
    mr = m.clone().requires_grad_(True)
    F.mish(mr).backward(pre_ref)
    ops.conv3x3(n=n, h=h, w=w, cin=ci, cout=co, compute=cd, inp=xv, out=ops.view(yb), wpacked=slab, bias=bs.to(cuda), mask=mv, mask_mode=3)
    assert rel_l2(_back(yb, 0, co), mr.grad) < 5e-6 + tol


def _hip_net(cuda, hp, sd, dtype, name="mosr"):
    from trainner_redux_amd.archs import build_network

    net = build_network({"type": name, "scale": hp["scale"], "n_block": hp["n_block"], "dim": hp["dim"], "kernel_size": hp["kernel_size"],
                         "expansion_ratio": hp["expansion_ratio"], "conv_ratio": hp["conv_ratio"]})
    assert list(net.state_dict().keys()) == list(sd.keys())
    net.load_state_dict(sd, strict=True)
    net = net.to(cuda).train()
    net.compute_dtype = dtype
    return net


def _oracle_run(hp, sd, x, gy, policy, keep):
    from oracle import mosr as OM

    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    y = OM.forward(params, x, hp, policy, keep_masks=keep)
    grads = dict(zip(params, torch.autograd.grad(y, list(params.values()), gy)))
    return y.detach(), grads


@pytest.mark.parametrize("cfg", [dict(scale=2, n_block=2, dim=48), dict(scale=4, n_block=3, dim=64, kernel_size=5, conv_ratio=0.5),
                                 dict(scale=3, n_block=1, dim=32, kernel_size=3, expansion_ratio=2.0)])
def test_mosr_fp32_matches_oracle(cuda, cfg):
    from oracle import mosr as OM
    from oracle.precision import FP32

    hp = OM.hparams(**cfg)
    sd = OM.init_params(hp, 1, std=0.08, bias_std=0.05)
    x = torch.rand(2, 3, 20, 27, generator=torch.Generator().manual_seed(2))
    keep = {1: torch.tensor([1.25, 0.0]), hp["n_block"]: torch.tensor([1.0, 2.0])}
    gy = torch.randn(2, 3, 20 * hp["scale"], 27 * hp["scale"], generator=torch.Generator().manual_seed(3))
    y_ref, g_ref = _oracle_run(hp, sd, x, gy, FP32, keep)
    net = _hip_net(cuda, hp, sd, torch.float32)
    y = net.forward_hip(x.to(cuda), train=True, keep_masks={k: v.to(cuda) for k, v in keep.items()})
    e_y = rel_l2(y.cpu(), y_ref)
    net.backward_hip(gy.to(cuda))
    errs = {k: rel_l2(p.grad.cpu(), g_ref[k]) for k, p in net.named_parameters()}
    worst = max(errs, key=errs.get)
    print(f"mosr fp32 {cfg}: fwd {e_y:.2e}, worst grad {worst} {errs[worst]:.2e}")
    assert y.shape == y_ref.shape and e_y < 1e-5, e_y
    assert errs[worst] < 2e-4, (worst, errs[worst])
    with torch.no_grad():                                   # inference path: same kernels, nothing saved
        assert rel_l2(net.eval()(x.to(cuda)).cpu(), OM.forward(sd, x, hp)) < 1e-5


@pytest.mark.parametrize("dim", [48, 64])      # 48: mosr_t (hid 72: generic conv epilogues); 64: mosr (hid 96 = 64 + 32: the 64-wide ring instances with half a last group)
@pytest.mark.parametrize("dtype,policy", [(torch.bfloat16, "BF16_OPS"), (torch.float16, "F16_OPS")])
def test_mosr_16bit_vs_policy_oracle(cuda, dtype, policy, dim):
    from oracle import mosr as OM
    from oracle import precision as OP

    hp = OM.hparams(scale=2, n_block=3, dim=dim)
    sd = OM.init_params(hp, 4, std=0.06, bias_std=0.02)
    x = torch.rand(2, 3, 24, 32, generator=torch.Generator().manual_seed(5))
    gy = torch.randn(2, 3, 48, 64, generator=torch.Generator().manual_seed(6)) / (2 * 3 * 48 * 64)
    y_ref, g_ref = _oracle_run(hp, sd, x, gy, getattr(OP, policy), None)
    net = _hip_net(cuda, hp, sd, dtype)
    y = net.forward_hip(x.to(cuda), train=True)
    e_y = rel_l2(y.cpu(), y_ref)
    net.backward_hip(gy.to(cuda))
    cs = {k: cosine(p.grad.cpu(), g_ref[k]) for k, p in net.named_parameters()}
    worst = min(cs, key=cs.get)
    print(f"mosr {dtype}: fwd {e_y:.2e}, worst grad cosine {worst} {cs[worst]:.4f}")
    assert e_y < 1e-2 and cs[worst] > 0.99, (e_y, worst, cs[worst])


def test_mosr_module_surface_and_full_size(cuda):
    """autograd bridge, registry names / checksums, DropPath on, mosr 2x at 64 -> 128 (BASELINE configs[4] tile), batch 4."""
    from trainner_redux_amd.archs import build_network

    net = build_network({"type": "mosr", "scale": 2, "drop_path": 0.1}).to(cuda).train()
    assert sum(p.numel() for p in net.parameters()) == 4_266_828          # docs/source/benchmarks.md:226
    x = torch.rand(4, 3, 64, 64, device=cuda)
    y = net(x)
    assert y.shape == (4, 3, 128, 128) and y.requires_grad
    (y * y).mean().backward()
    for k, p in net.named_parameters():
        assert torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0, k
    net_t = build_network({"type": "mosr_t", "scale": 4}).to(cuda).eval()
    assert sum(p.numel() for p in net_t.parameters()) == 609_888          # docs/source/benchmarks.md:29
    assert net_t(torch.rand(1, 3, 20, 28, device=cuda)).shape == (1, 3, 80, 112)
