"""CPU suite (no GPU): pins the oracle to the reference's own numbers and fixtures.

  * parameter-count checksums from /root/reference/docs/source/benchmarks.md:64,221,42
  * tests/golden/pixel_gan_losses.pt -- outputs of the reference's basic_loss.py / gan_loss.py (make_golden.py)
  * behavioural invariants the reference's tests assert (tests/test_losses/test_losses.py:114-206,
    tests/test_archs/test_archs.py:182-185,246-268)
"""
import os
from collections import OrderedDict

import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def test_esrgan_param_count_checksums():
    from oracle import esrgan as OE

    assert OE.param_count(OE.esrgan_hparams(scale=4)) == 16_697_987          # benchmarks.md:64
    assert OE.param_count(OE.esrgan_hparams(scale=2)) == 16_703_171          # benchmarks.md:221 (pixel-unshuffle front)
    assert OE.param_count(OE.esrgan_hparams(scale=4, num_filters=32, num_blocks=12)) == 5_021_155   # esrgan_lite, :42
    hp = OE.esrgan_hparams(scale=4)
    keys = list(OE.init_state_dict(hp).keys())
    assert keys[0] == "model.0.weight" and "model.1.sub.0.RDB1.conv1.0.weight" in keys and "model.1.sub.23.weight" in keys
    assert keys[-1] == "model.10.bias"


@pytest.mark.parametrize("scale,kw", [(4, {}), (2, {}), (2, dict(use_pixel_unshuffle=False)), (1, {})])
def test_esrgan_oracle_shape_and_one_step_is_finite(scale, kw):
    """tests/test_archs/test_archs.py: output H, W == scale x input; one L1 + AdamW step leaves everything finite."""
    from oracle import esrgan as OE, losses as OL, step as OS

    hp = OE.esrgan_hparams(scale=scale, num_blocks=1, **kw)
    sd = OE.init_state_dict(hp, 1)
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    x = torch.rand(1, 3, 16, 16)
    gt = torch.rand(1, 3, 16 * scale, 16 * scale)
    st = OS.AdamWState(OrderedDict((k, v.data) for k, v in params.items()))
    out, ld, log, grads = OS.generator_step(params, lambda p, t: OE.forward(p, t, hp), {"l_g_l1": (OL.l1, 1.0)}, x, gt, st, None)
    assert out.shape == gt.shape
    assert all(torch.isfinite(g).all() for g in grads.values()) and all(torch.isfinite(p).all() for p in params.values())
    assert set(ld) == {"l_g_l1", "l_g_total", "grad_norm_g"}


def test_oracle_losses_match_reference_golden():
    from oracle import losses as OL

    fix = torch.load(os.path.join(GOLD, "pixel_gan_losses.pt"), weights_only=True)
    pred, target, logits = fix["pred"], fix["target"], fix["logits"]
    fns = {"l1": OL.l1, "mse": OL.mse, "charbonnier": OL.charbonnier}
    for key, ref in fix["pixel"].items():
        name, red = key.split("/")
        p = pred.clone().requires_grad_(True)
        v = fns[name](p, target, reduction=red)
        v.backward()
        assert torch.equal(v.detach(), ref["value"]), key
        assert torch.equal(p.grad, ref["grad"]), key
    for key, ref in fix["gan"].items():
        gt_, real, disc = key.split("/")
        v = OL.gan(logits, bool(int(real)), is_disc=bool(int(disc)), gan_type=gt_)
        assert torch.equal(v, ref), key


def test_loss_invariants_of_the_reference_tests():
    from oracle import losses as OL, vgg as OV

    black = torch.zeros(2, 3, 32, 32)
    for fn in (OL.l1, OL.mse, OL.charbonnier):
        assert float(fn(black, black)) <= 1e-5                                     # test_losses.py:114-128
    sd = OV.synthetic_state_dict(0)
    assert all(float(v) <= 1e-5 for v in OV.perceptual_loss(sd, black, black).values())
    a, b = torch.rand(1, 3, 32, 32), torch.rand(1, 3, 32, 32)
    one = OV.perceptual_loss(sd, a, b)
    two = OV.perceptual_loss(sd, torch.cat([a, a]), torch.cat([b, b]))             # batch-duplication invariance, :185-206
    for k in one:
        assert abs(float(one[k]) - float(two[k])) < 1e-6
    assert list(one.keys()) == ["conv1_2", "conv2_2", "conv3_4", "conv4_4", "conv5_4"]   # stage set, :130-179
    with pytest.raises(ValueError):
        OL.l1(a, b, reduction="bogus")


def test_vgg_table_matches_torchvision_indices():
    from oracle import vgg as OV

    t = OV.conv_table()
    assert len(t) == 16 and [i for _, i, _, _ in t][:5] == [0, 2, 5, 7, 10] and t[-1][1] == 34
    assert [c for _, _, _, c in t] == [64, 64, 128, 128, 256, 256, 256, 256, 512, 512, 512, 512, 512, 512, 512, 512]


def test_adamw_oracle_equals_torch_optim():
    from oracle import step as OS

    p = torch.randn(257)
    q = p.clone().requires_grad_(True)
    o = torch.optim.AdamW([q], lr=2e-4, betas=(0.9, 0.99), weight_decay=0.01)
    st = OS.AdamWState(OrderedDict(p=p), 2e-4, (0.9, 0.99), 1e-8, 0.01)
    for _ in range(7):
        g = torch.randn(257)
        q.grad = g.clone()
        o.step()
        st.step(OrderedDict(p=p), OrderedDict(p=g))
    assert torch.equal(p, q.detach())


def test_bf16_policy_sits_closer_to_fp32_than_autocast():
    """oracle/precision.py: the HIP bf16 contract (bf16 MFMA operands, fp32 everything else) vs the reference's AMP path."""
    from oracle import esrgan as OE
    from oracle.precision import BF16_OPS

    hp = OE.esrgan_hparams(scale=4, num_blocks=4)
    sd = OE.init_state_dict(hp, 0)
    x = torch.rand(1, 3, 16, 16)
    with torch.no_grad():
        y = OE.forward(sd, x, hp)
        yb = OE.forward(sd, x, hp, BF16_OPS)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ya = OE.forward(sd, x, hp).float()
    rel = lambda a: ((a - y).norm() / y.norm()).item()  # noqa: E731
    assert rel(yb) < rel(ya) < 5e-2


def test_discriminator_oracle_matches_reference_golden():
    """oracle/discriminator.py vs the reference module's own tensors (UNetDiscriminatorSN, num_feat 16)."""
    from oracle import discriminator as OD

    fix = torch.load(os.path.join(GOLD, "unet_discriminator_sn.pt"), weights_only=True)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and ("weight_orig" in k or k.startswith(("conv0", "conv9"))) else v.clone())
          for k, v in fix["state_before"].items()}
    x = fix["x"].clone().requires_grad_(True)
    y, uv = OD.forward(sd, x, training=True)
    assert torch.allclose(y, fix["y"], rtol=1e-5, atol=1e-6)
    (y * fix["gy"]).sum().backward()
    assert torch.allclose(x.grad, fix["gx"], rtol=1e-4, atol=1e-6)
    for k, g in fix["grads"].items():
        assert torch.allclose(sd[k].grad, g, rtol=1e-4, atol=1e-6), k
    for name, (u, v) in uv.items():
        assert torch.allclose(u, fix["state_after"][f"{name}.weight_u"], atol=1e-6)
        assert torch.allclose(v, fix["state_after"][f"{name}.weight_v"], atol=1e-6)
    sd_after = {k: v for k, v in fix["state_after"].items()}
    with torch.no_grad():
        ye, _ = OD.forward(sd_after, fix["x"], training=False)
    assert torch.allclose(ye, fix["y_eval"], rtol=1e-5, atol=1e-6)


def test_span_oracle_matches_reference_module_goldens():
    """oracle/span.py against the reference's OWN SPAN module (tests/golden/span.pt, made by make_golden.py from span_arch.py):
    training-mode output (Conv3XC's 1x1 -> 3x3 -> 1x1 + skip form), every parameter gradient, eval-mode output (merged 3x3), merged weights."""
    import os
    from collections import OrderedDict

    from oracle import span as OS

    fix = torch.load(os.path.join(os.path.dirname(__file__), "golden", "span.pt"), weights_only=True)
    for tag, e in fix.items():
        kw = e["kwargs"]
        hp = OS.hparams(feature_channels=kw["feature_channels"], scale=kw["upscale"], norm=kw["norm"], img_range=kw.get("img_range", 255.0))
        sd = e["state"]
        assert [k for k in OS.param_table(hp)] == [k for k in sd if k != "no_norm"], tag
        assert all(tuple(sd[k].shape) == shp for k, shp in OS.param_table(hp).items())
        params = OrderedDict((k, v.clone().requires_grad_(v.dtype.is_floating_point and "eval_conv" not in k)) for k, v in sd.items())
        y = OS.forward(params, e["x"], hp, train=True)
        assert rel_l2(y, e["y_train"]) < 2e-6, (tag, rel_l2(y, e["y_train"]))
        names = list(e["grads"].keys())
        grads = torch.autograd.grad(y, [params[k] for k in names], e["gy"])
        worst = max(rel_l2(g, e["grads"][k]) for k, g in zip(names, grads))
        assert worst < 2e-5, (tag, worst)
        with torch.no_grad():
            ye = OS.forward(sd, e["x"], hp, train=False)
            assert rel_l2(ye, e["y_eval"]) < 2e-6
            # training and eval forms are the same function (the merged kernel reproduces the bias seen at the padded border)
            assert rel_l2(ye, e["y_train"]) < 1e-5
            for name in OS.conv3xc_names(hp):
                w, b = OS.merged_kernel(sd, name)
                assert rel_l2(w, e["state_eval"][name + ".eval_conv.weight"]) < 1e-6 and rel_l2(b, e["state_eval"][name + ".eval_conv.bias"]) < 1e-6


def test_mosr_oracle_reproduces_every_published_parameter_count():
    """spandrel's MoSR is absent: the restated structure is anchored on the reference's own checksums
    (docs/source/benchmarks.md:43,132,226,315 mosr 4x / 3x / 2x / 1x; :29,117,211 mosr_t)."""
    from oracle import mosr as OM

    want = {("mosr", 4): 4_287_600, ("mosr", 3): 4_275_483, ("mosr", 2): 4_266_828, ("mosr", 1): 4_261_635,
            ("mosr_t", 4): 609_888, ("mosr_t", 3): 600_795, ("mosr_t", 2): 594_300}
    for (name, sc), cnt in want.items():
        assert OM.num_params(OM.hparams(scale=sc, **OM.PRESETS[name])) == cnt, (name, sc)
    hp = OM.hparams(scale=2, n_block=2, dim=16)
    sd = OM.init_params(hp, 0, std=0.1, bias_std=0.1)
    for v in sd.values():
        v.requires_grad_(True)
    x = torch.rand(2, 3, 12, 10)
    y = OM.forward(sd, x, hp)
    assert y.shape == (2, 3, 24, 20)                                    # tests/test_archs/test_archs.py:148-268: shape = scale x, finite
    y.square().mean().backward()
    assert all(torch.isfinite(v.grad).all() and v.grad.abs().sum() > 0 for v in sd.values())
    keep = {1: torch.tensor([0.0, 2.0])}                                # a dropped sample passes through its block unchanged
    y0 = OM.forward(sd, x, hp, keep_masks=keep)
    assert torch.isfinite(y0).all() and not torch.allclose(y0[1], y[1])
