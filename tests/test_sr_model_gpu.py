"""`SRModel.optimize_parameters` end to end on the GPU against the oracle's step (oracle/step.py, oracle/discriminator.py).

fp32 compute: the weights after one full step (forward, all heads, backward, AdamW, EMA) must agree with the oracle to 1e-5;
the log keys are the reference's (tests/test_models/test_sr_model.py:63-80)."""
import os
from collections import OrderedDict

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _opt(losses, accum=1, network_d=None, optim_d=None, use_amp=False, grad_clip=False, blocks=2, network_g=None):
    from trainner_redux_amd.utils.options import parse_options_dict

    train = dict(total_iter=100, ema_decay=0.999, ema_power=0.75, grad_clip=grad_clip,
                 optim_g=dict(type="AdamW", lr=2e-4, weight_decay=0, betas=[0.9, 0.99]), losses=losses)
    if optim_d:
        train["optim_d"] = optim_d
    d = dict(name="t", scale=4, num_gpu=1, path={}, use_amp=use_amp, amp_bf16=True, network_g=network_g or dict(type="esrgan", num_blocks=blocks),
             datasets=dict(train=dict(name="x", type="pairedimagedataset", accum_iter=accum)), train=train)
    if network_d:
        d["network_d"] = network_d
    return parse_options_dict(d)


def _data(seed=1024, n=2, s=16):
    g = torch.Generator().manual_seed(seed)
    gt = torch.rand(n, 3, 4 * s, 4 * s, generator=g)
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", antialias=True).clamp(0, 1)
    return lq, gt


def test_l1_perceptual_step_with_accumulation_matches_oracle(cuda):
    import warnings

    from oracle import esrgan as OE, losses as OL, step as OS, vgg as OV
    from trainner_redux_amd.models import SRModel

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = SRModel(_opt([dict(type="l1loss", loss_weight=1.0), dict(type="perceptualloss", loss_weight=0.01)], accum=2, grad_clip=True))
    hp = OE.esrgan_hparams(scale=4, num_blocks=2)
    sd = OE.init_state_dict(hp, 0)
    vsd = OV.synthetic_state_dict(7)
    model.net_g.load_state_dict(sd)
    model.losses["l_g_perceptual"].vgg.load_torchvision_state_dict(vsd)
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    ost = OS.AdamWState(OrderedDict((k, v.data) for k, v in params.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    oema = OS.EMAState(OrderedDict((k, v.data) for k, v in params.items()), beta=0.999, power=0.75)
    fns = OrderedDict(l_g_l1=(OL.l1, 1.0), l_g_perceptual=(lambda o, t: OV.perceptual_loss(vsd, o, t), 0.01))
    fwd = lambda p, x: OE.forward(p, x, hp)  # noqa: E731
    logs = []
    for it, apply in ((1, False), (2, True)):          # accum_iter = 2: second micro-step applies the gradient
        lq, gt = _data(seed=it)
        _, ld, _, _ = OS.generator_step(params, fwd, fns, lq, gt, ost, oema, accum_iters=2, apply_gradient=apply, grad_clip=1.0)
        logs.append(ld)
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it, it - 1, apply)
    log = model.get_current_log()
    expect = {"l_g_l1", "l_g_total", "grad_norm_g"} | {f"l_g_perceptual_{k}" for k in OV.CONV_LAYER_WEIGHTS}
    assert expect <= set(log), set(log)
    # logged values are sample-weighted means over both micro-steps (base_model.py:186-200)
    for k in ("l_g_l1", "l_g_total", "l_g_perceptual_conv3_4"):
        ref = sum(float(l[k]) for l in logs) / 2
        assert abs(log[k] - ref) < 2e-5 * abs(ref), (k, log[k], ref)
    # dict losses are multiplied by accum_iters in the total, scalar ones divided (sr_model.py:654-663)
    l0 = logs[0]
    tot = float(l0["l_g_l1"]) / 2 + sum(float(v) for k, v in l0.items() if k.startswith("l_g_perceptual")) * 2
    assert abs(tot - float(l0["l_g_total"])) < 1e-6
    e_w = max(rel_l2(p.detach().cpu(), params[k].data) for k, p in model.net_g.state_dict().items())
    e_ema = max(rel_l2(p.detach().cpu(), oema.shadow[k]) for k, p in model.net_g_ema.ema_model.state_dict().items())
    print("weights after accumulated step", e_w, "ema", e_ema, "grad_norm", log["grad_norm_g"], float(logs[1]["grad_norm_g"]))
    # AdamW's first update is lr * g / (|g| + eps): a gradient element within fp32 noise of zero can flip its +-lr step, so the
    # weights agree to (flipped fraction)^0.5 * lr / |w| rather than to rounding (L1-only: 2e-7 in smoke(); with the VGG head ~1e-4)
    assert e_w < 5e-4 and e_ema < 5e-4
    assert abs(log["grad_norm_g"] * 2 - float(logs[1]["grad_norm_g"])) < 1e-4 * float(logs[1]["grad_norm_g"])  # logged only on the applying step


def test_gan_step_matches_oracle(cuda):
    """L1 + vanilla GAN with UNetDiscriminatorSN: G and D weights after one step, u/v after the three power iterations."""
    from oracle import discriminator as OD, esrgan as OE, losses as OL, step as OS
    from trainner_redux_amd.models import SRModel

    wg = 0.1
    torch.manual_seed(0)             # the discriminator's initial weights and u / v come from torch's global generator
    model = SRModel(_opt([dict(type="l1loss", loss_weight=1.0), dict(type="ganloss", gan_type="vanilla", loss_weight=wg)],
                         network_d=dict(type="unetdiscriminatorsn", num_in_ch=3, num_feat=16, skip_connection=True),
                         optim_d=dict(type="AdamW", lr=1e-4, weight_decay=0, betas=[0.9, 0.99]), blocks=1))
    hp = OE.esrgan_hparams(scale=4, num_blocks=1)
    sd = OE.init_state_dict(hp, 3)
    model.net_g.load_state_dict(sd)
    dsd = {k: v.detach().cpu().clone() for k, v in model.net_d.state_dict().items()}
    lq, gt = _data(seed=5, n=2, s=8)
    gp = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    dkeys = [k for k in dsd if not k.endswith(("weight_u", "weight_v"))]
    dp = OrderedDict((k, dsd[k].clone().requires_grad_(True)) for k in dkeys)
    state = dict(dsd)

    def dnet(x):
        cur = {**state, **dp}
        y, uv = OD.forward(cur, x, training=True)
        for name, (u, v) in uv.items():
            state[f"{name}.weight_u"], state[f"{name}.weight_v"] = u, v
        return y

    out = OE.forward(gp, lq, hp)
    l1 = OL.l1(out, gt)
    lgan = OL.gan(dnet(out), True, is_disc=False)
    total = l1 * 1.0 + lgan * wg
    gg = torch.autograd.grad(total, list(gp.values()))
    ogs = OS.AdamWState(OrderedDict((k, v.data) for k, v in gp.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    ogs.step(OrderedDict((k, v.data) for k, v in gp.items()), OrderedDict(zip(gp.keys(), gg)))
    ldr = OL.gan(dnet(gt), True, is_disc=True)
    ldf = OL.gan(dnet(out.detach()), False, is_disc=True)
    dg = torch.autograd.grad((ldr + ldf) * wg, list(dp.values()))
    ods = OS.AdamWState(OrderedDict((k, v.data) for k, v in dp.items()), lr=1e-4, betas=(0.9, 0.99), weight_decay=0.0)
    ods.step(OrderedDict((k, v.data) for k, v in dp.items()), OrderedDict(zip(dp.keys(), dg)))

    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(1, 0, True)
    log = model.get_current_log()
    assert {"l_g_l1", "l_g_gan", "l_g_total", "grad_norm_g", "l_d_real", "l_d_fake", "grad_norm_d"} <= set(log), set(log)
    for k, ref in (("l_g_l1", l1), ("l_g_gan", lgan * wg), ("l_g_total", total), ("l_d_real", ldr), ("l_d_fake", ldf)):
        assert abs(log[k] - float(ref)) < 2e-5 * abs(float(ref)), (k, log[k], float(ref))
    e_g = max(rel_l2(p.detach().cpu(), gp[k].data) for k, p in model.net_g.state_dict().items())
    dnow = model.net_d.state_dict()
    e_d = max(rel_l2(dnow[k].cpu(), dp[k].data) for k in dkeys)
    e_uv = max(rel_l2(dnow[k].cpu(), state[k]) for k in dsd if k.endswith(("weight_u", "weight_v")))
    # the rounding-tight comparison of the D update is its first moment (0.1 x gradient); the weights after a FIRST AdamW step move by lr * g / (|g| + eps),
    # i.e. by +-lr wherever |g| >> eps: a gradient element at fp32 rounding level can flip its step (seen once in ~10 unseeded runs: 7e-5 on the D weights)
    od = model.optimizers[1]
    m_d = {k: od.state[p]["exp_avg"].cpu() for k, p in model.net_d.named_parameters() if p in od.state}
    e_md = max(rel_l2(m_d[k], ods.m[k]) for k in dkeys if k in m_d)
    print("G weights", e_g, "D weights", e_d, "D first moments", e_md, "u/v", e_uv)
    assert e_g < 1e-5 and e_md < 1e-5 and e_d < 5e-4 and e_uv < 1e-5


def _gan_model(losses, seed=0, adaptive_d=False):
    from oracle import esrgan as OE
    from trainner_redux_amd.models import SRModel

    torch.manual_seed(seed)
    opt = _opt(losses, network_d=dict(type="unetdiscriminatorsn", num_in_ch=3, num_feat=16, skip_connection=True),
               optim_d=dict(type="AdamW", lr=1e-4, weight_decay=0, betas=[0.9, 0.99]), blocks=1)
    opt.train.adaptive_d = adaptive_d
    opt.train.scheduler = dict(type="MultiStepLR", milestones=[50], gamma=0.5)
    model = SRModel(opt)
    model.net_g.load_state_dict(OE.init_state_dict(OE.esrgan_hparams(scale=4, num_blocks=1), 3))
    return model


def test_iteration_scheduled_weights_reach_the_heads_like_plain_weights(cuda):
    """IterativeLossWrapper on the HIP path (losses/__init__.py:79-110, sr_model.py:523-663,763-768): a step at iteration 5 of
    [L1 0.5 ramping 0.5 -> 1.5 over 10, vanilla GAN 1.0 ramping 1.0 -> 0.2 over 10, MSE not started yet] must equal the step of the plain configuration
    [L1 0.5 * 1.0, GAN 1.0 * 0.6] -- G side weight = abs(loss_weight) * schedule(iter), D side weight = schedule(iter) -- and the inactive
    head logs 0 and moves nothing."""
    sched = [dict(type="l1loss", loss_weight=0.5, start_iter=0, target_iter=10, target_weight=1.5),
             dict(type="ganloss", gan_type="vanilla", loss_weight=1.0, start_iter=0, target_iter=10, target_weight=0.2, schedule_type="linear"),
             dict(type="mseloss", loss_weight=1.0, start_iter=100, warn_on_unused=False)]
    plain = [dict(type="l1loss", loss_weight=0.5), dict(type="ganloss", gan_type="vanilla", loss_weight=0.6)]
    a, b = _gan_model(sched), _gan_model(plain)
    assert type(a.losses["l_g_gan"]).__name__ == "IterativeLossWrapper" and a.losses["l_g_gan"].get_current_weight(5) == pytest.approx(0.6)
    lq, gt = _data(seed=5, n=2, s=8)
    for m in (a, b):
        m.feed_data({"lq": lq, "gt": gt})
        m.optimize_parameters(5, 0, True)
    la, lb = a.get_current_log(), b.get_current_log()
    assert la["l_g_mse"] == 0.0 and set(la) - {"l_g_mse"} == set(lb), (set(la), set(lb))
    for k in lb:
        assert abs(la[k] - lb[k]) <= 1e-6 * abs(lb[k]), (k, la[k], lb[k])
    for na, nb in ((a.net_g, b.net_g), (a.net_d, b.net_d)):
        sa, sb = na.state_dict(), nb.state_dict()
        e = max(rel_l2(sa[k].float().cpu(), sb[k].float().cpu()) for k in sb)
        assert e < 1e-6, e
    # the ramp moves on: at iteration 10 the GAN weight is the target on both sides
    assert a.losses["l_g_gan"].get_current_weight(10) == pytest.approx(0.2) and a.losses["l_g_l1"].get_current_weight(10) == 1.5


def test_adaptive_d_skips_the_discriminator_update_when_the_gan_ema_rises(cuda):
    """sr_model.py:558-569: ema' = 0.999 ema + 0.001 l_g_gan (value before loss_weight); ema' > 1.02 ema skips the D update of the step.  The EMA
    starts at 0 (base_model.py:55), so the first step always skips; with a large EMA the update runs."""
    wg = 0.1
    m = _gan_model([dict(type="l1loss", loss_weight=1.0), dict(type="ganloss", gan_type="vanilla", loss_weight=wg)], adaptive_d=True)
    lq, gt = _data(seed=5, n=2, s=8)
    d0 = {k: v.detach().clone() for k, v in m.net_d.state_dict().items() if not k.endswith(("weight_u", "weight_v"))}
    m.feed_data({"lq": lq, "gt": gt})
    m.optimize_parameters(1, 0, True)
    log = m.get_current_log()
    assert "l_d_real" not in log and "grad_norm_d" not in log and m.optimizers_skipped == [False, True]
    l_gan = log["l_g_gan"] / wg
    assert float(m.l_g_gan_ema) == pytest.approx(0.001 * l_gan, rel=1e-5)
    now = m.net_d.state_dict()
    assert all(torch.equal(now[k], v) for k, v in d0.items())
    m.update_learning_rate(1)
    assert m.schedulers[1].last_epoch == 0 and m.schedulers[0].last_epoch == 1            # the skipped optimizer's scheduler waits (base_model.py:378)

    m.reset_current_log()
    m.l_g_gan_ema.fill_(100.0)
    m.feed_data({"lq": lq, "gt": gt})
    m.optimize_parameters(2, 0, True)
    log = m.get_current_log()
    assert {"l_d_real", "l_d_fake", "grad_norm_d"} <= set(log) and m.optimizers_skipped == [False, False]
    assert float(m.l_g_gan_ema) == pytest.approx(99.9 + 0.001 * log["l_g_gan"] / wg, rel=1e-6)
    now = m.net_d.state_dict()
    assert any(not torch.equal(now[k], v) for k, v in d0.items())


def test_bf16_step_and_foreign_loss_head(cuda):
    """bf16 compute (use_amp + amp_bf16) with a stock-PyTorch loss head registered through the open plugin surface."""
    from torch import nn

    from trainner_redux_amd.models import SRModel
    from trainner_redux_amd.utils.registry import LOSS_REGISTRY

    if "stocktvloss" not in LOSS_REGISTRY:
        @LOSS_REGISTRY.register()
        class StockTVLoss(nn.Module):   # a plain nn.Module, as any of the reference's 25+ other losses would be
            def __init__(self, loss_weight: float):
                super().__init__()
                self.loss_weight = loss_weight

            def forward(self, x, gt):
                return (x[..., 1:] - x[..., :-1]).abs().mean() + (x[..., 1:, :] - x[..., :-1, :]).abs().mean()

    model = SRModel(_opt([dict(type="charbonnierloss", loss_weight=1.0), dict(type="stocktvloss", loss_weight=0.05)], use_amp=True))
    assert model.net_g.compute_dtype == torch.bfloat16
    lq, gt = _data(seed=9)
    w0 = model.net_g.flat_parameters()[0].clone()
    for it in range(1, 4):
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it, 0, True)
    log = model.get_current_log()
    assert {"l_g_charbonnier", "l_g_stocktv", "l_g_total", "grad_norm_g"} <= set(log)
    assert all(v == v and abs(v) < 1e6 for v in log.values())
    assert abs(log["l_g_total"] - (log["l_g_charbonnier"] + log["l_g_stocktv"])) < 1e-5
    assert (model.net_g.flat_parameters()[0] - w0).abs().max().item() > 0
    assert model.output.shape == gt.shape
    model.test()
    assert model.output.shape == gt.shape


def test_swinir_step_matches_oracle(cuda):
    """SwinIR generator behind the same trainer: one L1 + Charbonnier step (fp32, DropPath off) against the oracle's step."""
    from oracle import losses as OL, step as OS, swinir as OW
    from trainner_redux_amd.models import SRModel

    over = dict(scale=4, depths=[2, 2], num_heads=[6, 6], drop_path_rate=0.0)
    model = SRModel(_opt([dict(type="l1loss", loss_weight=1.0), dict(type="charbonnierloss", loss_weight=0.5)],
                         network_g=dict(type="swinir_s", **over)))
    hp = OW.hparams("swinir_s", **over)
    sd = OW.init_state_dict(hp, 0)
    model.net_g.load_state_dict(sd, strict=False)
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    ost = OS.AdamWState(OrderedDict((k, v.data) for k, v in params.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    oema = OS.EMAState(OrderedDict((k, v.data) for k, v in params.items()), beta=0.999, power=0.75)
    fns = OrderedDict(l_g_l1=(OL.l1, 1.0), l_g_charbonnier=(OL.charbonnier, 0.5))
    lq, gt = _data(seed=5)
    _, ld, _, _ = OS.generator_step(params, lambda p, x: OW.forward(p, x, hp), fns, lq, gt, ost, oema)
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(1, 0, True)
    log = model.get_current_log()
    for k in ("l_g_l1", "l_g_charbonnier", "l_g_total"):
        assert abs(log[k] - float(ld[k])) < 1e-5 * abs(float(ld[k])), (k, log[k], float(ld[k]))
    sdm = model.net_g.state_dict()
    e_w = max(rel_l2(sdm[k].detach().cpu(), v.data) for k, v in params.items())
    print("swinir weights after one step", e_w)
    assert e_w < 5e-4       # AdamW first-step sign sensitivity, as above


def test_checkpoint_resume_is_bit_exact(cuda, tmp_path):
    """save() + save_training_state() after 2 steps, a fresh model + load_network() + resume_training(), then step 3 on both:
    generator, EMA and optimizer moments continue bit for bit (base_model.py:713-845, sr_model.py:1174-1205)."""
    from trainner_redux_amd.models import SRModel

    def make():
        opt = _opt([dict(type="l1loss", loss_weight=1.0)], blocks=1)
        opt.path.models = str(tmp_path / "models")
        opt.path.resume_models = str(tmp_path / "models" / "resume_models")
        opt.path.training_states = str(tmp_path / "training_states")
        opt.train.scheduler = dict(type="MultiStepLR", milestones=[2], gamma=0.5)
        return SRModel(opt)

    a = make()
    sd0 = {k: v.clone() for k, v in a.net_g.state_dict().items()}
    for it in (1, 2):
        lq, gt = _data(seed=it)
        a.feed_data({"lq": lq, "gt": gt})
        a.optimize_parameters(it, 0, True)
        a.update_learning_rate(it, -1)
    a.save(0, 2)
    a.save_training_state(0, 2)
    assert sorted(os.listdir(tmp_path / "models")) == ["net_g_ema_2.safetensors", "resume_models"]
    b = make()
    b.net_g.load_state_dict(sd0)
    b.load_network(b.net_g, str(tmp_path / "models" / "resume_models" / "net_g_2.safetensors"))
    b.load_network(b.net_g_ema.ema_model, str(tmp_path / "models" / "net_g_ema_2.safetensors"))
    state = torch.load(tmp_path / "training_states" / "2.state", weights_only=False)      # our own file
    assert state["iter"] == 2 and set(state["optimizers"][0]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    b.resume_training(state)
    assert b.get_current_learning_rate() == a.get_current_learning_rate()
    lq, gt = _data(seed=3)
    for m in (a, b):
        m.feed_data({"lq": lq, "gt": gt})
        m.optimize_parameters(3, 0, True)
    for (k, va), vb in zip(a.net_g.state_dict().items(), b.net_g.state_dict().values()):
        assert torch.equal(va, vb), k
    for va, vb in zip(a.net_g_ema.ema_model.state_dict().values(), b.net_g_ema.ema_model.state_dict().values()):
        assert torch.equal(va, vb)


def test_span_step_matches_oracle(cuda):
    """SPAN generator behind the trainer (the registry default `norm: false`, 2x): one L1 step in fp32 against the oracle's step."""
    from oracle import losses as OL, span as OSP, step as OS
    from trainner_redux_amd.models import SRModel

    opt = _opt([dict(type="l1loss", loss_weight=1.0)], network_g=dict(type="span", feature_channels=16))
    model = SRModel(opt)
    hp = OSP.hparams(feature_channels=16, scale=4, norm=False)
    sd = {k: v.detach().cpu().clone() for k, v in model.net_g.state_dict().items() if k != "no_norm"}
    assert list(sd) == list(OSP.param_table(hp))
    params = OrderedDict((k, v.clone().requires_grad_("eval_conv" not in k)) for k, v in sd.items())
    train = OrderedDict((k, v) for k, v in params.items() if "eval_conv" not in k)
    ost = OS.AdamWState(OrderedDict((k, v.data) for k, v in train.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    lq, gt = _data(seed=4)
    _, ld, _, _ = OS.generator_step(train, lambda p, x: OSP.forward({**params, **p}, x, hp), OrderedDict(l_g_l1=(OL.l1, 1.0)), lq, gt, ost, None)
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(1, 0, True)
    log = model.get_current_log()
    assert abs(log["l_g_l1"] - float(ld["l_g_l1"])) < 1e-5 * float(ld["l_g_l1"])
    assert abs(log["grad_norm_g"] - float(ld["grad_norm_g"])) < 1e-4 * float(ld["grad_norm_g"])
    sdm = model.net_g.state_dict()
    e_w = max(rel_l2(sdm[k].detach().cpu(), v.data) for k, v in train.items())
    print("span weights after one step", e_w)
    assert e_w < 5e-4


def test_span_three_steps_fused_sweep_grads_cleared_and_ema_tracks(cuda):
    """SPAN keeps 40 frozen `eval_conv.*` tensors: they must sit outside the trainable arena so that the fused sweep (AdamW + EMA + zero_grad)
    runs.  Three steps against the oracle: weights, EMA weights, and an all-zero gradient arena after every step (the wgrad kernels
    accumulate, so a missed zero_grad shows as a wrong step 2)."""
    from oracle import losses as OL, span as OSP, step as OS
    from trainner_redux_amd.models import SRModel

    opt = _opt([dict(type="l1loss", loss_weight=1.0)], network_g=dict(type="span", feature_channels=16))
    model = SRModel(opt)
    hp = OSP.hparams(feature_channels=16, scale=4, norm=False)
    sd = {k: v.detach().cpu().clone() for k, v in model.net_g.state_dict().items() if k != "no_norm"}
    params = OrderedDict((k, v.clone().requires_grad_("eval_conv" not in k)) for k, v in sd.items())
    train = OrderedDict((k, v) for k, v in params.items() if "eval_conv" not in k)
    parr, garr = model.net_g.flat_parameters()
    assert parr.numel() == sum((v.numel() + 3) // 4 * 4 for v in train.values())        # the arena holds the trainable tensors only
    assert model.net_g_ema.ema_arena().numel() == parr.numel()
    ost = OS.AdamWState(OrderedDict((k, v.data) for k, v in train.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    ema = OS.EMAState(OrderedDict((k, v.data) for k, v in train.items()), beta=0.999, power=0.75, update_after_step=0)
    for it in range(3):
        lq, gt = _data(seed=40 + it)
        _, ld, _, _ = OS.generator_step(train, lambda p, x: OSP.forward({**params, **p}, x, hp), OrderedDict(l_g_l1=(OL.l1, 1.0)), lq, gt, ost, ema)
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it + 1, 0, True)
        assert float(garr.abs().max()) == 0.0, f"gradient arena not cleared after step {it + 1}"
        assert not model.optimizer_g._group_arenas(model.optimizer_g.param_groups[0])[1], "SPAN parameters fell back to per-tensor launches"
        log = model.get_current_log()
        model.reset_current_log()
        assert abs(log["grad_norm_g"] - float(ld["grad_norm_g"])) < 1e-3 * float(ld["grad_norm_g"]), it
    sdm = model.net_g.state_dict()
    e_w = max(rel_l2(sdm[k].detach().cpu(), v.data) for k, v in train.items())
    sde = model.net_g_ema.ema_model.state_dict()
    e_e = max(rel_l2(sde[k].detach().cpu(), ema.shadow[k]) for k in train)
    moved = max(rel_l2(sde[k].detach().cpu(), sd[k]) for k in train)
    print("span after three steps: weights", e_w, "ema", e_e, "ema moved", moved)
    assert e_w < 1e-3 and e_e < 1e-3 and moved > 0


def test_partial_arena_fallback_still_clears_grads_and_updates_ema(cuda):
    """A user freezing part of a HIP generator hands the optimizer a partial arena: the per-tensor fallback must still leave a zero gradient
    arena and a moving EMA (the fused sweep's side effects)."""
    from trainner_redux_amd.models import SRModel

    opt = _opt([dict(type="l1loss", loss_weight=1.0)], blocks=1)
    model = SRModel(opt)
    frozen = model.net_g.conv_first.weight if hasattr(model.net_g, "conv_first") else next(model.net_g.parameters())
    frozen.requires_grad_(False)
    model.optimizers.clear(); model.optimizers_skipped.clear()
    model.setup_optimizers()
    parr, garr = model.net_g.flat_parameters()
    w0 = parr.clone()
    e0 = model.net_g_ema.ema_arena().clone()
    for it in range(3):
        lq, gt = _data(seed=50 + it)
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it + 1, 0, True)
        assert float(garr.abs().max()) == 0.0
    assert not torch.equal(parr, w0)
    ema = model.net_g_ema.ema_arena()
    assert not torch.equal(ema, e0) and torch.isfinite(ema).all()
    # decay at step 3 of power 0.75: the EMA sits strictly between the initial and the online weights
    assert float((ema - parr).abs().max()) > 0


def test_ema_switch_iter_copies_ema_back_into_the_online_model(cuda):
    """`ema_switch_iter: 3` (ema-pytorch `update_model_with_ema_every`, sr_model.py:223-234).  EMA step counter s = 0, 1: the EMA is a copy of
    the online weights anyway (update_after_step 0, then the `initted` copy); s = 2: a real lerp, the two differ; s = 3: lerp, then the EMA is
    copied back into the online model; s = 4: they differ again."""
    from trainner_redux_amd.models import SRModel
    from trainner_redux_amd.utils.options import parse_options_dict

    d = dict(name="t", scale=4, num_gpu=1, path={}, use_amp=False, network_g=dict(type="esrgan", num_blocks=1),
             train=dict(total_iter=100, ema_decay=0.999, ema_power=0.75, ema_switch_iter=3, optim_g=dict(type="AdamW", lr=2e-4, weight_decay=0, betas=[0.9, 0.99]),
                        losses=[dict(type="l1loss", loss_weight=1.0)]))
    model = SRModel(parse_options_dict(d))
    same = []
    for it in range(5):
        lq, gt = _data(seed=60 + it)
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it + 1, 0, True)
        same.append(torch.equal(model.net_g.flat_parameters()[0], model.net_g_ema.ema_arena()))
    assert same == [True, True, False, True, False], same


def test_non_finite_gradient_norm_leaves_weights_untouched_and_raises_before_saving(cuda, tmp_path):
    """A NaN batch: the device-side guard of tr_adamw_ema skips the update (weights, moments, EMA unchanged, gradients cleared) and `save`
    raises the reference's error (sr_model.py:665-666) instead of writing a poisoned checkpoint."""
    from trainner_redux_amd.models import SRModel

    opt = _opt([dict(type="l1loss", loss_weight=1.0)], blocks=1)
    opt.path.models = str(tmp_path / "models")
    opt.path.training_states = str(tmp_path / "states")
    model = SRModel(opt)
    lq, gt = _data(seed=70)
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(1, 0, True)
    model.get_current_log()
    parr, garr = model.net_g.flat_parameters()
    w1, e1 = parr.clone(), model.net_g_ema.ema_arena().clone()
    bad = gt.clone()
    bad[0, 0, 0, 0] = float("nan")
    model.feed_data({"lq": lq, "gt": bad})
    model.optimize_parameters(2, 0, True)
    assert torch.equal(parr, w1) and torch.equal(model.net_g_ema.ema_arena(), e1) and float(garr.abs().max()) == 0.0
    with pytest.raises(RuntimeError, match="NaN/Inf found in loss"):
        model.save(0, 2)
    assert not (tmp_path / "models").exists() or not any((tmp_path / "models").iterdir())


def test_mosr_step_matches_oracle(cuda):
    """MoSR generator behind the trainer (`mosr_t` shape, 4x): one L1 step in fp32 against the oracle's step."""
    from oracle import losses as OL, mosr as OM, step as OS
    from trainner_redux_amd.models import SRModel

    opt = _opt([dict(type="l1loss", loss_weight=1.0)], network_g=dict(type="mosr_t", n_block=2, dim=32))
    model = SRModel(opt)
    hp = OM.hparams(scale=4, n_block=2, dim=32)
    sd = {k: v.detach().cpu().clone() for k, v in model.net_g.state_dict().items()}
    assert list(sd) == list(OM.param_table(hp))
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    ost = OS.AdamWState(OrderedDict((k, v.data) for k, v in params.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    lq, gt = _data(seed=5)
    _, ld, _, _ = OS.generator_step(params, lambda p, x: OM.forward(p, x, hp), OrderedDict(l_g_l1=(OL.l1, 1.0)), lq, gt, ost, None)
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(1, 0, True)
    log = model.get_current_log()
    assert abs(log["l_g_l1"] - float(ld["l_g_l1"])) < 1e-5 * float(ld["l_g_l1"])
    assert abs(log["grad_norm_g"] - float(ld["grad_norm_g"])) < 1e-4 * float(ld["grad_norm_g"])
    sdm = model.net_g.state_dict()
    e_w = max(rel_l2(sdm[k].detach().cpu(), v.data) for k, v in params.items())
    print("mosr weights after one step", e_w)
    assert e_w < 5e-4


@pytest.mark.parametrize("arch", [dict(type="esrgan", num_blocks=2), dict(type="span", feature_channels=16), dict(type="mosr_t", n_block=2, dim=32)])
def test_hip_graph_replay_is_bit_identical_to_eager_steps(cuda, arch):
    """`use_compile: true` = hipGraph replay of the generator's forward / backward (archs/_arena.py): six steps on changing batches give
    bit-identical weights, EMA and logs to the eager trainer; a batch of another shape falls back to eager launches and re-captures."""
    from trainner_redux_amd.models import SRModel
    from trainner_redux_amd.utils.misc import set_random_seed

    def run(graphs):
        set_random_seed(3)
        opt = _opt([dict(type="l1loss", loss_weight=1.0)], network_g=arch)
        opt.use_compile = graphs
        m = SRModel(opt)
        logs = []
        for it in range(1, 7):
            lq, gt = _data(seed=100 + it, n=2, s=16 if it != 5 else 12)       # step 5: another tile size
            m.feed_data({"lq": lq, "gt": gt})
            m.optimize_parameters(it, 0, True)
            logs.append(m.get_current_log()["l_g_l1"])
        return m, logs

    a, la = run(False)
    b, lb = run(True)
    assert b.net_g.hip_graphs and b.net_g._gs is not None and not a.net_g.hip_graphs
    assert la == lb
    for (k, va), vb in zip(a.net_g.state_dict().items(), b.net_g.state_dict().values()):
        assert torch.equal(va, vb), k
    for va, vb in zip(a.net_g_ema.ema_model.state_dict().values(), b.net_g_ema.ema_model.state_dict().values()):
        assert torch.equal(va, vb)


def test_yuv444_pixel_format_matches_reference_colour_functions_and_trains(cuda):
    """`input_pixel_format` / `output_pixel_format: yuv444` (sr_model.py:471-481): the BT.601 kernels against the reference's OWN
    utils/color_util.py outputs (tests/golden/otf_degradations.pt), the adjoint against autograd, and one trainer step against the oracle
    step with the conversions wrapped around the generator."""
    from oracle import esrgan as OE, losses as OL, step as OS
    from trainner_redux_amd import ops
    from trainner_redux_amd.models import SRModel

    c = torch.load(os.path.join(os.path.dirname(__file__), "golden", "otf_degradations.pt"), weights_only=True)["color"]
    assert (ops.color601(c["rgb"].to(cuda), 0).cpu() - c["ycbcr"]).abs().max() < 2e-7
    assert (ops.color601(c["any_ycbcr"].to(cuda), 1).cpu() - c["any_rgb"]).abs().max() < 1e-6
    assert (ops.color601(c["ycbcr"].to(cuda), 1).cpu() - c["back"]).abs().max() < 1e-6

    def rgb2ycc(x):
        w = torch.tensor([[65.481, -37.797, 112.0], [128.553, -74.203, -93.786], [24.966, 112.0, -18.214]])
        return (torch.matmul(x.permute(0, 2, 3, 1), w).permute(0, 3, 1, 2) + torch.tensor([16.0, 128.0, 128.0]).view(1, 3, 1, 1)) / 255.0

    def ycc2rgb(x):
        y, cb, cr = (x * 255.0).split(1, 1)
        return torch.cat((1.164 * (y - 16) + 1.596 * (cr - 128), 1.164 * (y - 16) - 0.392 * (cb - 128) - 0.813 * (cr - 128),
                          1.164 * (y - 16) + 2.017 * (cb - 128)), 1) / 255.0

    v = c["any_ycbcr"].clone().requires_grad_(True)
    gr = torch.randn(v.shape, generator=torch.Generator().manual_seed(1))
    ycc2rgb(v).backward(gr)
    assert (ops.color601(gr.to(cuda), 2).cpu() - v.grad).abs().max() < 1e-6
    with pytest.raises(NotImplementedError):
        o = _opt([dict(type="l1loss", loss_weight=1.0)]); o.input_pixel_format = "y"; SRModel(o)

    opt = _opt([dict(type="l1loss", loss_weight=1.0)])
    opt.input_pixel_format = opt.output_pixel_format = "yuv444"
    model = SRModel(opt)
    hp = OE.esrgan_hparams(scale=4, num_blocks=2)
    sd = {k: v.detach().cpu().clone() for k, v in model.net_g.state_dict().items()}
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    ost = OS.AdamWState(OrderedDict((k, v.data) for k, v in params.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    lq, gt = _data(seed=9)
    _, ld, _, _ = OS.generator_step(params, lambda p, x: ycc2rgb(OE.forward(p, rgb2ycc(x), hp)), OrderedDict(l_g_l1=(OL.l1, 1.0)), lq, gt, ost, None)
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(1, 0, True)
    log = model.get_current_log()
    assert abs(log["l_g_l1"] - float(ld["l_g_l1"])) < 1e-5 * float(ld["l_g_l1"])
    e_w = max(rel_l2(model.net_g.state_dict()[k].detach().cpu(), v.data) for k, v in params.items())
    assert e_w < 5e-4, e_w


def test_configs2_composite_step_on_an_otf_pair_matches_oracle(cuda):
    """BASELINE.json configs[2] as ONE step: RealESRGANModel.feed_data degrades a GT batch on the device (lens distortion, blur, sensor noise,
    exposure, resize, sinc filter, a JPEG round trip on the host, 8-bit rounding, paired crop), then `optimize_parameters` runs the ESRGAN generator
    with L1 + perceptual (fp32 VGG19, dict-valued) + vanilla GAN through UNetDiscriminatorSN and the discriminator update -- checked against the oracle
    pieces composed in the reference's order (sr_model.py:452-828) on the very pair the feed produced: every log value, G / D weights after their AdamW
    steps, spectral-norm u / v after the three power iterations."""
    import warnings

    from oracle import discriminator as OD, esrgan as OE, losses as OL, step as OS, vgg as OV
    from trainner_redux_amd.models import RealESRGANModel, build_model
    from trainner_redux_amd.utils.options import parse_options_dict

    wg, wp = 0.1, 0.01
    opt = parse_options_dict(dict(
        name="c2", scale=4, num_gpu=1, path={}, manual_seed=11, use_amp=False, high_order_degradation=True, queue_size=2,
        lens_distort_prob=1.0, blur_prob=1.0, sensor_noise_prob=1.0, exposure_prob=1.0, resize_mode_list3=["bicubic"], resize_mode_prob3=[1.0],
        compression_formats=["jpeg"], compression_weights=[1.0],
        network_g=dict(type="esrgan", num_blocks=1), network_d=dict(type="unetdiscriminatorsn", num_in_ch=3, num_feat=16, skip_connection=True),
        datasets=dict(train=dict(name="x", type="realesrgandataset", lq_size=16, batch_size_per_gpu=2)),
        train=dict(total_iter=100, ema_decay=0.999, ema_power=0.75, optim_g=dict(type="AdamW", lr=2e-4, weight_decay=0, betas=[0.9, 0.99]),
                   optim_d=dict(type="AdamW", lr=1e-4, weight_decay=0, betas=[0.9, 0.99]),
                   losses=[dict(type="l1loss", loss_weight=1.0), dict(type="perceptualloss", loss_weight=wp, criterion="charbonnier"),
                           dict(type="ganloss", gan_type="vanilla", loss_weight=wg)])))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model = build_model(opt)
    assert isinstance(model, RealESRGANModel)
    hp = OE.esrgan_hparams(scale=4, num_blocks=1)
    sd = OE.init_state_dict(hp, 3)
    vsd = OV.synthetic_state_dict(7)
    model.net_g.load_state_dict(sd)
    model.losses["l_g_perceptual"].vgg.load_torchvision_state_dict(vsd)
    dsd = {k: v.detach().cpu().clone() for k, v in model.net_d.state_dict().items()}
    g = torch.Generator().manual_seed(21)
    gt_full = torch.rand(2, 3, 96, 96, generator=g)
    k1 = torch.rand(2, 21, 21, generator=g)
    k1 = k1 / k1.sum((1, 2), keepdim=True)
    sinc = torch.zeros(2, 21, 21)
    sinc[:, 10, 10] = 1.0
    model.feed_data({"gt": gt_full, "kernel1": k1, "kernel2": k1, "sinc_kernel": sinc})
    lq, gt = model.lq.cpu().clone(), model.gt.cpu().clone()
    assert lq.shape == (2, 3, 16, 16) and gt.shape == (2, 3, 64, 64) and torch.equal(lq, (lq * 255).round() / 255)
    assert 0.02 < (lq - torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", antialias=True)).abs().mean() < 0.5      # really degraded

    gp = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    dkeys = [k for k in dsd if not k.endswith(("weight_u", "weight_v"))]
    dp = OrderedDict((k, dsd[k].clone().requires_grad_(True)) for k in dkeys)
    state = dict(dsd)

    def dnet(x):
        cur = {**state, **dp}
        y, uv = OD.forward(cur, x, training=True)
        for name, (u, v) in uv.items():
            state[f"{name}.weight_u"], state[f"{name}.weight_v"] = u, v
        return y

    out = OE.forward(gp, lq, hp)
    l1 = OL.l1(out, gt)
    perc = OV.perceptual_loss(vsd, out, gt)
    lgan = OL.gan(dnet(out), True, is_disc=False)
    total = l1 * 1.0 + sum(v * wp for v in perc.values()) + lgan * wg          # accum_iters = 1: scalar / dict scaling coincide (sr_model.py:654-663)
    gg = torch.autograd.grad(total, list(gp.values()))
    ogs = OS.AdamWState(OrderedDict((k, v.data) for k, v in gp.items()), lr=2e-4, betas=(0.9, 0.99), weight_decay=0.0)
    ogs.step(OrderedDict((k, v.data) for k, v in gp.items()), OrderedDict(zip(gp.keys(), gg)))
    ldr = OL.gan(dnet(gt), True, is_disc=True)
    ldf = OL.gan(dnet(out.detach()), False, is_disc=True)
    dg = torch.autograd.grad((ldr + ldf) * wg, list(dp.values()))
    ods = OS.AdamWState(OrderedDict((k, v.data) for k, v in dp.items()), lr=1e-4, betas=(0.9, 0.99), weight_decay=0.0)
    ods.step(OrderedDict((k, v.data) for k, v in dp.items()), OrderedDict(zip(dp.keys(), dg)))

    model.optimize_parameters(1, 0, True)
    log = model.get_current_log()
    want = {"l_g_l1": l1, "l_g_gan": lgan * wg, "l_g_total": total, "l_d_real": ldr, "l_d_fake": ldf}
    want.update({f"l_g_perceptual_{k}": v * wp for k, v in perc.items()})
    assert set(want) | {"grad_norm_g", "grad_norm_d"} <= set(log), set(log)
    for k, ref in want.items():
        assert abs(log[k] - float(ref)) < 3e-5 * abs(float(ref)), (k, log[k], float(ref))
    gn = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(t, 2) for t in gg]))
    assert abs(log["grad_norm_g"] - float(gn)) < 1e-4 * float(gn)
    # first moments (0.1 x gradient) are the rounding-tight comparison; the weights carry AdamW's first-step sign sensitivity (see above)
    og = model.optimizers[0]
    m = {k: og.state[p]["exp_avg"].cpu() for k, p in model.net_g.named_parameters()}
    e_m = max(rel_l2(m[k], ogs.m[k]) for k in gp)
    e_g = max(rel_l2(p.detach().cpu(), gp[k].data) for k, p in model.net_g.state_dict().items())
    dnow = model.net_d.state_dict()
    e_d = max(rel_l2(dnow[k].cpu(), dp[k].data) for k in dkeys)
    e_uv = max(rel_l2(dnow[k].cpu(), state[k]) for k in dsd if k.endswith(("weight_u", "weight_v")))
    print(f"configs[2] composite step: G first moments {e_m:.2e}, G weights {e_g:.2e}, D weights {e_d:.2e}, u/v {e_uv:.2e}")
    assert e_m < 2e-4 and e_g < 5e-4 and e_d < 5e-4 and e_uv < 1e-5


def test_negative_loss_weight_selects_the_bicubic_lq_target(cuda):
    """sr_model.py:503-516: `loss_weight < 0` makes that loss compare the output with clamp(bicubic-antialias upsample of LQ) instead of GT, weighted by
    |loss_weight|; the other losses keep GT.  One fp32 step against the oracle's step with the same two targets."""
    from oracle import esrgan as OE, losses as OL, step as OS
    from trainner_redux_amd.models import SRModel

    model = SRModel(_opt([dict(type="l1loss", loss_weight=1.0), dict(type="charbonnierloss", loss_weight=-0.5)], blocks=1))
    hp = OE.esrgan_hparams(scale=4, num_blocks=1)
    sd = OE.init_state_dict(hp, 2)
    model.net_g.load_state_dict(sd)
    lq, gt = _data(seed=13, n=2, s=12)
    lq_up = torch.nn.functional.interpolate(lq, scale_factor=4, mode="bicubic", antialias=True).clamp(0, 1)
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    out = OE.forward(params, lq, hp)
    l1, lc = OL.l1(out, gt), OL.charbonnier(out, lq_up)
    total = l1 + 0.5 * lc
    grads = torch.autograd.grad(total, list(params.values()))
    model.feed_data({"lq": lq, "gt": gt})
    model.optimize_parameters(1, 0, True)
    log = model.get_current_log()
    for k, ref in (("l_g_l1", l1), ("l_g_charbonnier", 0.5 * lc), ("l_g_total", total)):
        assert abs(log[k] - float(ref)) < 2e-5 * abs(float(ref)), (k, log[k], float(ref))
    og = model.optimizers[0]
    e = max(rel_l2(og.state[p]["exp_avg"].cpu(), 0.1 * g) for p, g in zip(model.net_g.parameters(), grads))
    assert e < 1e-4, e
    assert abs(float(lc) - float(OL.charbonnier(out, gt))) > 1e-3          # the two targets really differ
