"""fp16 compute + loss scaling (the reference's default AMP mode: use_amp without amp_bf16, GradScaler at
sr_model.py:188-191,681-719).  Kernels use v_mfma_f32_32x32x16_f16 with fp32 accumulation; the oracle policy `f16_ops` rounds the same
operands to fp16.  Tolerances as for bf16 but tighter where fp16's 10-bit mantissa allows: forward <= 3e-3 vs the same-policy oracle."""
from collections import OrderedDict

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def test_esrgan_fp16_vs_policy_oracle(cuda):
    from oracle import esrgan as OE, losses as OL
    from oracle.precision import F16_OPS
    from trainner_redux_amd.archs import build_network

    hp = OE.esrgan_hparams(scale=4, num_blocks=2)
    sd = OE.init_state_dict(hp, 0)
    net = build_network({"type": "esrgan", "scale": 4, "num_blocks": 2})
    net.load_state_dict(sd)
    net = net.to(cuda).train()
    net.compute_dtype = torch.float16
    g = torch.Generator().manual_seed(2)
    x, gt = torch.rand(2, 3, 24, 40, generator=g), torch.rand(2, 3, 96, 160, generator=g)
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    out_ref = OE.forward(params, x, hp, F16_OPS)
    loss = OL.l1(out_ref, gt)
    # loss scaling on BOTH sides: d(loss)/d(out) ~ 1e-6 underflows fp16 operands (that is what GradScaler exists for)
    grads = torch.autograd.grad(loss * 1024.0, [out_ref, *params.values()])
    out = net.forward_hip(x.to(cuda), train=True)
    e = rel_l2(out.cpu(), out_ref.detach())
    net.backward_hip(grads[0].to(cuda))
    cs = {k: cosine(p.grad.cpu(), gr) for (k, p), gr in zip(net.state_dict(keep_vars=True).items(), grads[1:])}
    worst = min(cs, key=cs.get)
    print(f"esrgan fp16: fwd {e:.2e}; worst grad cosine {worst} {cs[worst]:.4f}")
    assert e < 3e-3 and cs[worst] > 0.99


def test_swinir_and_span_fp16_forward(cuda):
    from oracle import span as OSP, swinir as OW
    from oracle.precision import F16_OPS
    from trainner_redux_amd.archs import build_network

    hp = OW.hparams("swinir_s", scale=4, depths=[2, 2], num_heads=[6, 6])
    sd = OW.init_state_dict(hp, 0)
    net = build_network({"type": "swinir_s", "scale": 4, "depths": [2, 2], "num_heads": [6, 6]})
    net.load_state_dict(sd, strict=False)
    net = net.to(cuda).train()
    net.compute_dtype = torch.float16
    x = torch.rand(2, 3, 16, 24, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ref = OW.forward(sd, x, hp, F16_OPS)
    out = net.forward_hip(x.to(cuda), train=True, keep_masks={})
    e_sw = rel_l2(out.cpu(), ref)
    net.backward_hip(torch.randn_like(out) * 64.0)          # MFMA attention backward in fp16 runs and stays finite
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())
    import os
    e = torch.load(os.path.join(os.path.dirname(__file__), "golden", "span.pt"), weights_only=True)["fc12_x4_norm"]
    kw = e["kwargs"]
    hp2 = OSP.hparams(feature_channels=kw["feature_channels"], scale=kw["upscale"], norm=kw["norm"], img_range=kw["img_range"])
    net2 = build_network({"type": "span", "feature_channels": 12, "scale": 4, "norm": True, "img_range": 255.0})
    net2.load_state_dict(e["state"])
    net2 = net2.to(cuda).eval()
    net2.compute_dtype = torch.float16
    with torch.no_grad():
        ref2 = OSP.forward({k: v for k, v in e["state"].items()}, e["x"], hp2, F16_OPS, train=True)
        out2 = net2(e["x"].to(cuda))
    e_sp = rel_l2(out2.cpu(), ref2)
    print(f"fp16 forward vs f16-policy oracle: swinir_s {e_sw:.2e}, span {e_sp:.2e}")
    assert e_sw < 3e-3 and e_sp < 3e-3


def test_fp16_step_with_loss_scaling_and_overflow_skip(cuda):
    """use_amp without amp_bf16: the step runs with scale 65536 and lands where the f16-policy oracle step lands; an overflowing
    step (scale forced to 2^60) is skipped: weights and Adam moments untouched, scale halved, scheduler not advanced."""
    from oracle import esrgan as OE, losses as OL, step as OS
    from oracle.precision import F16_OPS
    from tests.test_sr_model_gpu import _data, _opt
    from trainner_redux_amd.models import SRModel

    opt = _opt([dict(type="l1loss", loss_weight=1.0)], use_amp=True, blocks=1)
    opt.amp_bf16 = False
    model = SRModel(opt)
    assert model.compute_dtype == torch.float16 and model.scaler_g.enabled and model.scaler_g.scale == 65536.0
    hp = OE.esrgan_hparams(scale=4, num_blocks=1)
    sd = OE.init_state_dict(hp, 0)
    model.net_g.load_state_dict(sd)
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    ost = OS.AdamWState(OrderedDict((k, v.data) for k, v in params.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    lq, gt = _data(seed=7)
    _, ld, _, _ = OS.generator_step(params, lambda p, x: OE.forward(p, x, hp, F16_OPS), OrderedDict(l_g_l1=(OL.l1, 1.0)), lq, gt, ost, None)
    # the oracle's weights come from a SCALED loss as well (unscaled fp16 gradients sit in the subnormals); a first Adam step is
    # invariant to the scale up to eps
    params2 = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    ost2 = OS.AdamWState(OrderedDict((k, v.data) for k, v in params2.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    OS.generator_step(params2, lambda p, x: OE.forward(p, x, hp, F16_OPS), OrderedDict(l_g_l1=(lambda o, t: OL.l1(o, t) * 65536.0, 1.0)), lq, gt, ost2, None)
    params = params2
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(1, 0, True)
    log = model.get_current_log()
    assert abs(log["l_g_l1"] - float(ld["l_g_l1"])) < 2e-3 * float(ld["l_g_l1"])
    assert abs(log["grad_norm_g"] - float(ld["grad_norm_g"])) < 0.1 * float(ld["grad_norm_g"])       # logged AFTER unscale_ (oracle value: unscaled, subnormal-noisy)
    moved = max(rel_l2(p.detach().cpu() - sd[k], params[k].data - sd[k]) for k, p in model.net_g.state_dict().items() if p.ndim == 4)
    print("fp16 step: update direction rel-L2 vs f16-policy oracle (first AdamW step = lr * sign-like):", moved)
    assert moved < 0.35          # sign flips of near-zero gradients dominate a first Adam step (see test_sr_model_gpu.py)
    # ---- overflow: skip
    before = {k: v.clone() for k, v in model.net_g.state_dict().items()}
    model.scaler_g.scale = 2.0 ** 60
    model.reset_current_log()
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(2, 0, True)
    assert model.optimizers_skipped[0] and model.scaler_g.scale == 2.0 ** 59
    for k, v in model.net_g.state_dict().items():
        assert torch.equal(v, before[k]), k
    g_arena = model.net_g.flat_parameters()[1]
    assert float(g_arena.abs().sum()) == 0.0           # the overflowed gradients are dropped, not accumulated


def test_span_2x_fp16_loss_scaled_training_step(cuda):
    """BASELINE.json configs[4]: SPAN 2x (52 feature channels, the `span` preset of archs/span_arch.py:326-346), fp16 AMP with loss scaling, through
    `SRModel.optimize_parameters`: the step with scale 65536 lands where the f16-policy oracle's step lands (loss value, unscaled gradient norm,
    update direction of the first AdamW step), the scaler keeps its scale on a clean step, and a second step keeps everything finite."""
    from oracle import losses as OL, span as OSP, step as OS
    from oracle.precision import F16_OPS
    from tests.test_sr_model_gpu import _opt
    from trainner_redux_amd.models import SRModel

    opt = _opt([dict(type="l1loss", loss_weight=1.0)], use_amp=True, network_g=dict(type="span", norm=False))
    opt.amp_bf16 = False
    opt.scale = 2
    model = SRModel(opt)
    assert model.compute_dtype == torch.float16 and model.scaler_g.enabled and model.net_g.compute_dtype == torch.float16
    hp = OSP.hparams(feature_channels=52, scale=2, norm=False)
    g0 = torch.Generator().manual_seed(3)
    sd = OrderedDict((k, torch.randn(shp, generator=g0) * (0.05 if len(shp) == 1 else (1.0 / (shp[1] * shp[2] * shp[3]) ** 0.5)))
                     for k, shp in OSP.param_table(hp).items() if "eval_conv" not in k)
    assert sum(v.numel() for v in sd.values()) + sum(p.numel() for k, p in model.net_g.named_parameters() if "eval_conv" in k) == 2_605_080      # docs/source/benchmarks.md:209
    res = model.net_g.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all("eval_conv" in k or k == "no_norm" for k in res.missing_keys), res
    g = torch.Generator().manual_seed(12)
    gt = torch.rand(4, 3, 64, 64, generator=g)
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.5, mode="bicubic", antialias=True).clamp(0, 1)
    fwd = lambda p, x: OSP.forward(p, x, hp, F16_OPS, train=True)  # noqa: E731
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    ost = OS.AdamWState(OrderedDict((k, v.data) for k, v in params.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    _, ld, _, _ = OS.generator_step(params, fwd, OrderedDict(l_g_l1=(OL.l1, 1.0)), lq, gt, ost, None)
    # the update direction from a SCALED loss, as the scaler produces it (unscaled fp16 gradients sit in the subnormals); Adam's first step is scale-invariant
    params2 = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    ost2 = OS.AdamWState(OrderedDict((k, v.data) for k, v in params2.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    _, ld2, _, _ = OS.generator_step(params2, fwd, OrderedDict(l_g_l1=(lambda o, t: OL.l1(o, t) * 65536.0, 1.0)), lq, gt, ost2, None)
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(1, 0, True)
    log = model.get_current_log()
    assert not model.optimizers_skipped[0] and model.scaler_g.scale == 65536.0
    assert abs(log["l_g_l1"] - float(ld["l_g_l1"])) < 2e-3 * float(ld["l_g_l1"]), (log["l_g_l1"], float(ld["l_g_l1"]))
    gn_ref = float(ld2["grad_norm_g"]) / 65536.0
    assert abs(log["grad_norm_g"] - gn_ref) < 0.05 * gn_ref, (log["grad_norm_g"], gn_ref)      # logged after unscale_ (sr_model.py:688-705)
    sdm = model.net_g.state_dict()
    big = [k for k, v in sd.items() if v.ndim == 4 and v.shape[-1] == 3]                       # the 3x3 kernels carry most of the update
    moved = max(rel_l2(sdm[k].detach().cpu() - sd[k], params2[k].data - sd[k]) for k in big)
    cos = min(cosine(sdm[k].detach().cpu() - sd[k], params2[k].data - sd[k]) for k in big)
    # the rounding-robust comparison: AdamW's first moment after step 1 = 0.1 x the UNSCALED gradient (inv_scale is folded into the sweep); the update
    # itself is lr * g / (|g| + eps), i.e. sign-like, and flips wherever a gradient element sits at fp16 rounding level (printed, not asserted tightly)
    og = model.optimizers[0]
    m_hip = {k: og.state[p]["exp_avg"].detach().cpu() for k, p in model.net_g.named_parameters() if p.requires_grad}
    m_cos = min(cosine(m_hip[k] * 65536.0, ost2.m[k]) for k in big)
    m_rel = max(rel_l2(m_hip[k] * 65536.0, ost2.m[k]) for k in big)
    print(f"span 2x fp16 step: loss {log['l_g_l1']:.6f} (oracle {float(ld['l_g_l1']):.6f}); first moments vs f16-policy oracle: min cosine {m_cos:.5f}, max rel-L2 {m_rel:.3f}; "
          f"update direction rel-L2 {moved:.3f}, min cosine {cos:.4f}")
    assert m_cos > 0.99 and m_rel < 0.15, (m_cos, m_rel)
    assert moved < 0.6 and cos > 0.8, (moved, cos)
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(2, 0, True)
    log2 = model.get_current_log()
    assert all(v == v and abs(v) < 1e6 for v in log2.values()) and torch.isfinite(model.net_g.flat_parameters()[0]).all()
