"""HIP path against outputs of the REFERENCE'S OWN in-tree classes / functions (tests/golden/*_twin.pt, paragon_otf.pt, psnr.pt -- generated in
the build container by tests/golden/make_golden_twins.py from reference files loaded by path; tests/test_twins_cpu.py holds the oracle to the
same files).  fp32 compute (`use_amp: false`, BASELINE configs[0]): v_mfma_f32_32x32x2_f32 is an exact fp32 FMA chain, so the only difference to
the reference's CPU ATen path is the summation order -- forward <= 1e-5 rel-L2, parameter gradients <= 2e-4 (tolerances of
tests/test_esrgan_gpu.py / test_swinir_gpu.py).

  ESRGAN generator       vs  archs/scunet_aaf6aa_arch.py:310-384   ResidualDenseBlock_ / RRDB_ / RRDBUpsample in ESRGAN's layer order
  SwinIR generator       vs  archs/drct_arch.py:52-223,415-575,624-725   SwinTransformerBlock (W-MSA / SW-MSA) / PatchEmbed / Upsample in SwinIR's order
  OTF stage primitives   vs  models/paragon_otf_degradations.py:251-575   each codec-free static stage, seeded
  calculate_psnr         vs  metrics/psnr_ssim.py:13-61
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _load(name):
    return torch.load(os.path.join(GOLD, name), weights_only=True)


@pytest.mark.parametrize("tag", ["nf32_x4", "nf64_x2"])
def test_esrgan_hip_matches_reference_rrdb_classes(cuda, tag):
    from oracle import esrgan as OE
    from trainner_redux_amd.archs import build_network

    fix = _load("esrgan_twin.pt")[tag]
    hp = OE.esrgan_hparams(**fix["kwargs"])
    sd = OE.init_state_dict(hp, fix["seed"])            # the seeded weights the fixture generator loaded into the reference classes
    net = build_network({"type": "esrgan", **fix["kwargs"]})
    net.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in net.parameters()) == fix["n_params"]
    net = net.to(cuda)
    net.compute_dtype = torch.float32
    net.train()
    out = net.forward_hip(fix["x"].to(cuda), train=True)
    e = rel_l2(out.cpu(), fix["out"])
    assert out.shape == fix["out"].shape and e < 1e-5, e
    net.backward_hip(fix["cotangent"].to(cuda))
    grads = {k: p.grad.cpu() for k, p in net.state_dict(keep_vars=True).items()}
    errs = {k: rel_l2(grads[k], g) for k, g in fix["grads"].items()}
    worst = max(errs, key=errs.get)
    print(f"ESRGAN {tag} vs reference RRDB_ classes: forward {e:.2e}, worst gradient {worst} {errs[worst]:.2e} over {len(errs)} tensors")
    assert errs[worst] < 2e-4, (worst, errs[worst])


def test_swinir_hip_matches_net_of_reference_blocks(cuda):
    from oracle import swinir as OS
    from trainner_redux_amd.archs import build_network

    fix = _load("swin_twin.pt")["net"]
    hp = OS.hparams(fix["name"], **fix["over"])
    sd = OS.init_state_dict(hp, fix["seed"])
    net = build_network({"type": fix["name"], **fix["over"], "drop_path_rate": 0.0})
    res = net.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all("relative_position_index" in k or "attn_mask" in k for k in res.missing_keys), res
    net = net.to(cuda)
    net.compute_dtype = torch.float32
    net.train()
    out = net.forward_hip(fix["x"].to(cuda), train=True)
    e = rel_l2(out.cpu(), fix["out"])
    assert out.shape == fix["out"].shape and e < 1e-5, e
    net.backward_hip(fix["cotangent"].to(cuda))
    errs = {k: rel_l2(p.grad.cpu(), fix["grads"][k]) for k, p in net.named_parameters()}
    assert set(errs) == set(fix["grads"])
    worst = max(errs, key=errs.get)
    print(f"SwinIR vs reference SwinTransformerBlock classes: forward {e:.2e}, worst gradient {worst} {errs[worst]:.2e}")
    assert errs[worst] < 2e-4, (worst, errs[worst])


def test_otf_primitives_match_reference_stage_functions(cuda):
    """Every codec-free ParagonOTF stage: the device primitive the degrader calls for it, with the parameters the reference drew, against the
    reference function's output.  Elementwise / stencil stages: fp32 rounding (2e-6); grid_sample stages: bilinear sampling is continuous in the
    coordinates, fp32 coordinate noise moves values by <= 5e-5 (tests/test_otf_gpu.py)."""
    from trainner_redux_amd.data import otf

    fix = _load("paragon_otf.pt")
    img = fix["img"]
    d = img.to(cuda)
    h, w = img.shape[2:]
    for tag, e in fix["stages"].items():
        kind, p, ref = tag.split("/")[0], e["params"], e["out"]
        tol = 2e-6
        if kind == "motion_blur":
            ks = p["kernel_size"]
            got = otf.conv_shared_zeropad(d, otf.motion_blur_kernel(ks, p["angle"]), ks // 2)
        elif kind == "lens_distortion":
            got, tol = otf.warp_bilinear(d, 0, [p["strength"]]), 5e-5
        elif kind == "exposure":
            got = otf.gain_noise_clamp(d, p["factor"])
        elif kind == "color_temperature":
            s = p["shift"]
            got = otf.gain_noise_clamp(d, (1 + s * 0.3, 1 + s * 0.1, 1.0) if s > 0 else (1.0, 1 - s * 0.1, 1 - s * 0.3))
        elif kind == "sensor_noise":
            got = otf.gain_noise_clamp(d, 1.0, p["noise"].to(cuda), p["std"])
        elif kind == "rolling_shutter":
            got, tol = otf.warp_bilinear(d, 1, [p["strength"] * h / w]), 5e-5
        elif kind == "oversharpen":
            blurred = otf.conv_shared_zeropad(d, torch.full((5, 5), 1.0 / 25.0), 2)
            got = otf.axpby_clamp01(d, 1.0 + p["strength"], blurred, -p["strength"])
        elif kind == "chromatic_aberration":
            got, tol = otf.warp_bilinear(d, 2, [1.001, 1.0, 0.999]), 5e-5
        elif kind == "aliasing":
            f = p["scale"]
            got = otf.resize_pt(otf.resize_pt(d, "nearest", size=(int(h * f), int(w * f)), clamp=False), "nearest", size=(h, w), clamp=False)
            tol = 0.0
        else:
            raise AssertionError(kind)
        err = (got.cpu() - ref).abs().max().item()
        assert got.shape == ref.shape and err <= tol, (tag, err)


def test_degrader_single_stage_runs_reproduce_reference_outputs(cuda):
    """The OTFDegrader itself (gates, draw order, primitives) with one stage's gate open and the fixture's seeds: its pre-resize image must be the
    reference stage function's output.  The degrader always ends in the final resize + sinc filter, so the comparison is on that composite:
    resize_pt(reference output) filtered by a delta kernel."""
    import random

    import numpy as np

    from trainner_redux_amd.data import otf
    from trainner_redux_amd.utils.options import parse_options_dict

    fix = _load("paragon_otf.pt")
    img = fix["img"]
    delta = torch.zeros(img.shape[0], 21, 21)
    delta[:, 10, 10] = 1.0
    for tag, e in fix["stages"].items():
        kind = tag.split("/")[0]
        if kind == "sensor_noise":
            continue                                    # its noise field is drawn on the device here (torch.randn_like on a cuda tensor), on the CPU in the fixture
        base = dict(name="t", scale=2, num_gpu=1, path={}, manual_seed=11, high_order_degradation=True, queue_size=4,      # 2x: the 21-tap sinc filter needs more than 10 rows
                    network_g=dict(type="span", feature_channels=16, norm=True),
                    datasets=dict(train=dict(name="x", type="realesrgandataset", lq_size=8, batch_size_per_gpu=2)),
                    train=dict(total_iter=100, ema_decay=0.999, optim_g=dict(type="AdamW", lr=2e-4, weight_decay=0, betas=[0.9, 0.99]),
                               losses=[dict(type="l1loss", loss_weight=1.0)]),
                    resize_mode_list3=["bicubic"], resize_mode_prob3=[1.0])
        base.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in e["opt"].items()})
        opt = parse_options_dict(base)
        # the gates in front of this stage are closed (probability 0) but still consume one draw each: give the degrader a generator that has
        # been advanced so that THIS stage's gate sees the fixture's first draw
        order = ["lens_distortion", "chromatic_aberration", "motion_blur", "blur", "demosaic", "sensor_noise", "rolling_shutter", "exposure", "color_temperature",
                 "oversharpen", "aliasing"]
        n_before = order.index(kind)

        class Shifted:
            """numpy Generator whose first `n_before` uniform() draws (closed gates) come from a throw-away stream."""

            def __init__(self, seed):
                self.real, self.skip = np.random.default_rng(seed), n_before

            def uniform(self, *a):
                if self.skip > 0 and not a:
                    self.skip -= 1
                    return 0.5
                return self.real.uniform(*a)

            def choice(self, *a, **k):
                return self.real.choice(*a, **k)

        deg = otf.OTFDegrader(opt, rng=Shifted(e["seed"]), rnd=random.Random(e["seed"]))
        got = deg._before_codec_stages(img.to(cuda), delta.to(cuda), delta.to(cuda))
        ref_stage = e["out"]
        hh, ww = img.shape[2] // 2, img.shape[3] // 2
        want = otf.filter2d(otf.resize_pt(ref_stage.to(cuda), "bicubic", size=(hh, ww)), delta.to(cuda))
        err = (got - want).abs().max().item()
        assert err < 1e-4, (tag, err)


def test_psnr_matches_reference_function(cuda):
    from trainner_redux_amd.metrics import calculate_psnr

    fix = _load("psnr.pt")
    a, b = fix["a"][:1].to(cuda), fix["b"][:1].to(cuda)
    for key, ref in fix["np"].items():
        crop, y = key.split("/")
        got = calculate_psnr(a, b, crop_border=int(crop), test_y_channel=bool(int(y)))
        assert abs(got - ref) < 1e-5 * abs(ref), (key, got, ref)
    assert calculate_psnr(a, a, crop_border=0) == float("inf")
