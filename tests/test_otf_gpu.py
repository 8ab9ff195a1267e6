"""OTF degradation primitives (first slice of SURVEY.md 8f-1): `filter2d` against the REFERENCE function's own outputs
(tests/golden/filter2d.pt from utils/img_process_util.py), the 8-bit clamp/round, and the training pair pool's bookkeeping."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def test_filter2d_matches_reference_function(cuda):
    from trainner_redux_amd.data import otf

    fix = torch.load(os.path.join(os.path.dirname(__file__), "golden", "filter2d.pt"), weights_only=True)
    for tag, e in fix.items():
        out = otf.filter2d(e["img"].to(cuda), e["kernel"].to(cuda))
        err = rel_l2(out.cpu(), e["out"])
        assert out.shape == e["out"].shape and err < 2e-6, (tag, err)
    with pytest.raises(ValueError, match="Wrong kernel size"):
        otf.filter2d(torch.rand(1, 3, 8, 8, device=cuda), torch.rand(1, 4, 4, device=cuda))


def test_clamp_round_and_pair_pool(cuda):
    from trainner_redux_amd.data import otf

    x = torch.randn(2, 3, 9, 11, generator=torch.Generator().manual_seed(1)) * 0.7 + 0.5
    assert torch.equal(otf.clamp_round_u8(x.to(cuda)).cpu(), torch.clamp((x * 255.0).round(), 0, 255) / 255.0)
    pool = otf.PairPool(queue_size=6)
    seen = []
    for i in range(3):                                                  # filling: pairs pass through unchanged
        lq, gt = torch.full((2, 1, 2, 2), float(i), device=cuda), torch.full((2, 1, 4, 4), float(i), device=cuda)
        a, b = pool(lq, gt)
        assert torch.equal(a, lq) and torch.equal(b, gt)
    torch.manual_seed(0)
    lq, gt = torch.full((2, 1, 2, 2), 9.0, device=cuda), torch.full((2, 1, 4, 4), 9.0, device=cuda)
    a, b = pool(lq, gt)                                                 # full: shuffled, first b handed out, replaced by the newcomers
    assert a.shape == lq.shape and set(a.flatten().tolist()) <= {0.0, 1.0, 2.0}
    assert torch.equal(a[:, 0, 0, 0], b[:, 0, 0, 0])                    # lq / gt stay paired through the shuffle
    vals = sorted(pool.queue_lr[:, 0, 0, 0].tolist())
    assert vals.count(9.0) == 2 and len(vals) == 6


@pytest.mark.parametrize("mode", ["bilinear", "bicubic", "nearest-exact", "lanczos"])
@pytest.mark.parametrize("shape,size", [((4, 3, 256, 256), (64, 64)), ((2, 3, 37, 53), (23, 31)), ((1, 1, 20, 24), (33, 40)), ((1, 3, 40, 40), (40, 17))])
def test_resize_pt_matches_torch_interpolate(cuda, mode, shape, size):
    """`resize_pt` (data/degradations.py:996-1021) = F.interpolate (+ antialias) / Lanczos prefilter + bicubic: the HIP gathers vs the same
    calls on the CPU; 3e-6 absolute on [0, 1] images (fp32 summation order)."""
    from oracle import otf as OO
    from trainner_redux_amd.data import otf

    img = torch.rand(*shape, generator=torch.Generator().manual_seed(3))
    ref = OO.resize_pt(img, mode, size=size)
    out = otf.resize_pt(img.to(cuda), mode, size=size)
    assert out.shape == ref.shape
    err = (out.cpu() - ref).abs().max().item()
    assert err < (1e-7 if mode == "nearest-exact" else 3e-6), err
    ref2 = OO.resize_pt(img, mode, scale_factor=0.5)
    assert (otf.resize_pt(img.to(cuda), mode, scale_factor=0.5).cpu() - ref2).abs().max().item() < 3e-6


def test_elementwise_and_blur_stages_match_oracle(cuda):
    from oracle import otf as OO
    from trainner_redux_amd.data import otf

    g = torch.Generator().manual_seed(9)
    img = torch.rand(2, 3, 45, 52, generator=g)
    d = img.to(cuda)
    assert torch.allclose(otf.gain_noise_clamp(d, 1.7).cpu(), OO.exposure(img, 1.7), atol=1e-7)
    for shift in (0.15, -0.12):
        gains = (1 + shift * 0.3, 1 + shift * 0.1, 1.0) if shift > 0 else (1.0, 1 - shift * 0.1, 1 - shift * 0.3)
        assert torch.allclose(otf.gain_noise_clamp(d, gains).cpu(), OO.color_temperature(img, shift), atol=1e-7)
    noise = torch.randn(img.shape, generator=g)
    assert torch.allclose(otf.gain_noise_clamp(d, 1.0, noise.to(cuda), 0.07).cpu(), OO.sensor_noise(img, noise, 0.07), atol=1e-7)
    for ks, angle in ((7, 30.0), (6, 120.0), (15, 271.0)):            # even sizes grow the image by one, as F.conv2d(padding=k // 2) does
        k = otf.motion_blur_kernel(ks, angle)
        ref = OO.motion_blur(img, k, ks)
        out = otf.conv_shared_zeropad(d, k, ks // 2)
        assert out.shape == ref.shape and (out.cpu() - ref).abs().max().item() < 2e-6
    s = 1.4
    blurred = otf.conv_shared_zeropad(d, torch.full((5, 5), 1.0 / 25.0), 2)
    assert (otf.axpby_clamp01(d, 1.0 + s, blurred, -s).cpu() - OO.oversharpen(img, s)).abs().max().item() < 2e-6
    f = 0.73
    h, w = img.shape[2:]
    al = otf.resize_pt(otf.resize_pt(d, "nearest", size=(int(h * f), int(w * f)), clamp=False), "nearest", size=(h, w), clamp=False)
    assert torch.equal(al.cpu(), OO.aliasing(img, f))
    c = otf.crop(d, 5, 7, 20, 30)
    assert torch.equal(c.cpu(), img[:, :, 5:25, 7:37])
    # grid_sample stages: bilinear sampling is continuous in the coordinates, so fp32 coordinate noise moves values by ~1e-5 at most
    for k in (0.27, -0.3):
        assert (otf.warp_bilinear(d, 0, [k]).cpu() - OO.lens_distortion(img, k)).abs().max().item() < 5e-5
    sq = torch.rand(1, 3, 40, 40, generator=g)
    assert (otf.warp_bilinear(sq.to(cuda), 0, [0.2]).cpu() - OO.lens_distortion(sq, 0.2)).abs().max().item() < 5e-5
    for st in (0.1, -0.07):
        assert (otf.warp_bilinear(d, 1, [st * 45 / 52]).cpu() - OO.rolling_shutter(img, st)).abs().max().item() < 5e-5
    assert (otf.warp_bilinear(d, 2, [1.001, 1.0, 0.999]).cpu() - OO.chromatic_aberration(img)).abs().max().item() < 5e-5


def _otf_opt(**kw):
    from trainner_redux_amd.utils.options import parse_options_dict

    base = dict(name="t", scale=4, num_gpu=1, path={}, manual_seed=11, high_order_degradation=True, queue_size=4,
                network_g=dict(type="span", feature_channels=16, norm=True),
                datasets=dict(train=dict(name="x", type="realesrgandataset", lq_size=16, batch_size_per_gpu=2)),
                train=dict(total_iter=100, ema_decay=0.999, optim_g=dict(type="AdamW", lr=2e-4, weight_decay=0, betas=[0.9, 0.99]),
                           losses=[dict(type="l1loss", loss_weight=1.0)]))
    base.update(kw)
    return parse_options_dict(base)


def test_degrader_follows_the_reference_stage_order_and_rng(cuda):
    """Whole stage list with every built stage switched on, against the oracle stages driven by a twin RNG in the reference's draw order
    (realesrgan_model.py:456-664).  Compression format `heif`: PIL has no HEIF plugin here, so -- exactly like the reference -- samples pass
    through and the comparison stays exact; the codec round-trip is covered separately."""
    import random

    import numpy as np

    from oracle import otf as OO
    from trainner_redux_amd.data import otf

    kw = dict(lens_distort_prob=1.0, chromatic_aberration_prob=1.0, rolling_shutter_prob=1.0, motion_blur_prob=1.0, motion_blur_kernel_size=(7, 7), blur_prob=1.0, sensor_noise_prob=1.0, exposure_prob=1.0, color_temp_prob=1.0,
              oversharpen_prob=1.0, aliasing_prob=1.0, resize_mode_list3=["bicubic"], resize_mode_prob3=[1.0], compression_formats=["heif"],
              compression_weights=[1.0], editing_prob=1.0, editing_exposure_prob=1.0)
    opt = _otf_opt(**kw)
    deg = otf.OTFDegrader(opt, rng=np.random.default_rng(5), rnd=random.Random(6))
    g = torch.Generator().manual_seed(2)
    gt = torch.rand(2, 3, 96, 96, generator=g)
    k1 = torch.rand(2, 21, 21, generator=g); k1 = k1 / k1.sum((1, 2), keepdim=True)
    sinc = torch.zeros(2, 21, 21); sinc[:, 10, 10] = 1.0
    torch.manual_seed(77)
    out = deg.degrade(gt.to(cuda), k1.to(cuda), sinc.to(cuda))
    # ---- the same walk on the CPU
    rng, rnd = np.random.default_rng(5), random.Random(6)
    torch.manual_seed(77)
    x = gt
    rng.uniform(); x = OO.lens_distortion(x, rng.uniform(-0.3, 0.3))
    rng.uniform(); x = OO.chromatic_aberration(x)
    assert rng.uniform() < 1.0
    ks = rnd.randint(7, 7); angle = rng.uniform(0, 360)
    x = OO.motion_blur(x, otf.motion_blur_kernel(ks, angle), ks)
    rng.uniform(); x = OO.filter2d(x, k1)
    rng.uniform()                                                         # demosaic gate
    rng.uniform(); std = rng.uniform(0.01, 0.1)
    noise = torch.randn(x.shape, device=cuda).cpu()                       # the device generator drew the field: replay it from the same seed
    x = OO.sensor_noise(x, noise, std)
    rng.uniform(); x = OO.rolling_shutter(x, rng.uniform(-0.1, 0.1))
    rng.uniform(); x = OO.exposure(x, rng.uniform(0.5, 2.0))
    rng.uniform(); x = OO.color_temperature(x, rng.uniform(-0.2, 0.2))
    rng.uniform(); x = OO.oversharpen(x, rng.uniform(1.0, 2.0))
    rng.uniform(); x = OO.aliasing(x, rng.uniform(0.6, 0.9))
    mode = rnd.choices(["bicubic"], weights=[1.0])[0]
    x = OO.resize_pt(x, mode, size=(24, 24))
    x = OO.filter2d(x, sinc)
    rng.choice(["heif"], p=[1.0]); rng.uniform(70, 90)                    # format + quality draws; no plugin -> unchanged
    rng.uniform()                                                         # recompression gate
    rng.uniform(); rng.uniform(); x = OO.exposure(x, rng.uniform(0.9, 1.1)); rng.uniform()
    assert out.shape == x.shape
    err = (out.cpu() - x).abs().max().item()
    assert err < 2e-4, err                                                # three bilinear warps + eight more stages in fp32
    assert deg.rng.uniform() == rng.uniform()                             # both streams consumed the same number of draws


def test_jpeg_round_trip_is_the_host_codec(cuda):
    from trainner_redux_amd.data import otf

    img = torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(4)).mul(0.5).add(0.25)
    out = otf.compress_round_trip(img.to(cuda), "jpeg", 60.7)
    again = otf.compress_round_trip(img, "jpeg", 60.7)                    # the codec runs on the host either way: same bytes
    assert out.is_cuda and torch.equal(out.cpu(), again) and not torch.equal(again, img)
    assert (again - img).abs().mean() < 0.2 and torch.equal(again, (again * 255).round() / 255)      # 8-bit levels
    assert torch.equal(otf.compress_round_trip(img, "nosuchformat", 50), img)


def test_realesrgan_model_trains_on_otf_pairs(cuda):
    """build_model picks RealESRGANModel for `high_order_degradation`; feed_data turns GT + kernels into a (lq, gt) pair of the configured
    tile size on 8-bit levels, the pair pool fills then shuffles, and optimize_parameters runs on the result."""
    from trainner_redux_amd.models import RealESRGANModel, build_model

    model = build_model(_otf_opt(compression_formats=["jpeg"], compression_weights=[1.0], blur_prob=1.0))
    assert isinstance(model, RealESRGANModel)
    g = torch.Generator().manual_seed(8)
    for it in range(1, 4):
        gt = torch.rand(2, 3, 128, 128, generator=g)
        k = torch.rand(2, 21, 21, generator=g); k = k / k.sum((1, 2), keepdim=True)
        sinc = torch.zeros(2, 21, 21); sinc[:, 10, 10] = 1.0
        model.feed_data({"gt": gt, "kernel1": k, "kernel2": k, "sinc_kernel": sinc})
        assert model.lq.shape == (2, 3, 16, 16) and model.gt.shape == (2, 3, 64, 64)
        lq_host = model.lq.cpu()
        assert torch.equal(lq_host, (lq_host * 255).round() / 255)          # 8-bit levels (checked with the CPU's correctly rounded division)
        model.optimize_parameters(it, 0, True)
    log = model.get_current_log()
    assert log["l_g_l1"] > 0 and model.otf.pool.queue_ptr == 4
    # prefetch: the next batch's feed runs on a side stream; feed_data of the same batch object collects it
    nxt = {"gt": torch.rand(2, 3, 128, 128, generator=g), "kernel1": k, "kernel2": k, "sinc_kernel": sinc}
    model.prefetch(nxt)
    model.optimize_parameters(4, 0, True)
    model.feed_data(nxt)
    assert model.lq.shape == (2, 3, 16, 16) and torch.isfinite(model.lq).all() and model._otf_pending is None
    model.optimize_parameters(5, 0, True)
    model.prefetch(nxt)
    with pytest.raises(RuntimeError, match="different batch"):
        model.feed_data({"gt": nxt["gt"], "kernel1": k, "kernel2": k, "sinc_kernel": sinc})
    assert type(build_model(_otf_opt(enable_sequences=True))).__name__ == "RealESRGANModel"      # accepted; the reference never consults the controller
    model.is_train = False
    model.feed_data({"lq": torch.rand(1, 3, 8, 8), "gt": torch.rand(1, 3, 32, 32)})      # validation batches stay paired
    assert model.lq.shape == (1, 3, 8, 8)


def test_resize_pt_matches_the_reference_function(cuda):
    """HIP resize against the outputs of the reference's OWN resize_pt (tests/golden/otf_degradations.pt, made by tests/golden/make_golden_otf.py
    from data/degradations.py loaded by path): 3e-6 absolute on [0, 1] images, nearest-exact bit-exact."""
    from trainner_redux_amd.data import otf

    fix = torch.load(os.path.join(os.path.dirname(__file__), "golden", "otf_degradations.pt"), weights_only=True)["resize"]
    for tag, e in fix.items():
        for mode, ref in e["out"].items():
            out = otf.resize_pt(e["img"].to(cuda), mode, **e["kw"]).cpu()
            assert out.shape == ref.shape, (tag, mode)
            err = (out - ref).abs().max().item()
            assert err < (1e-7 if mode == "nearest-exact" else 3e-6), (tag, mode, err)


def test_otf_feed_end_to_end_from_image_files(cuda, tmp_path):
    """Files on disk -> RealESRGANDataset (GT tile + three kernels per sample) -> DataLoader -> DevicePrefetcher (pinned H2D on a side
    stream) -> RealESRGANModel.feed_data (device degradations, codec hop, crop, pool) -> optimize_parameters: the whole OTF feed path."""
    import random

    import numpy as np
    from PIL import Image

    from trainner_redux_amd.data import DevicePrefetcher, EnlargedSampler, RealESRGANDataset, build_dataloader
    from trainner_redux_amd.models import build_model

    g = np.random.default_rng(1)
    for i in range(6):
        Image.fromarray(g.integers(0, 255, (150 + 7 * i, 170, 3), dtype=np.uint8)).save(tmp_path / f"{i}.png")
    opt = _otf_opt(compression_formats=["jpeg", "webp"], compression_weights=[0.5, 0.5], blur_prob=1.0, sensor_noise_prob=0.5, exposure_prob=0.5,
                   datasets=dict(train=dict(name="x", type="realesrgandataset", dataroot_gt=[str(tmp_path)], lq_size=16, gt_size=64, batch_size_per_gpu=2,
                                            sinc_prob=0.3, final_sinc_prob=0.5)))
    random.seed(3)
    ds = RealESRGANDataset(opt.datasets["train"], rng=np.random.default_rng(3))
    loader = build_dataloader(ds, batch_size=2, sampler=EnlargedSampler(ds, 1, 0, ratio=2))
    feed = DevicePrefetcher(loader, cuda)
    model = build_model(opt)
    steps = 0
    while True:
        batch = feed.next()
        if batch is None:
            break
        assert batch["gt"].is_cuda and batch["gt"].shape == (2, 3, 96, 96) and batch["kernel1"].shape == (2, 21, 21)
        model.feed_data(batch)
        assert model.lq.shape == (2, 3, 16, 16) and model.gt.shape == (2, 3, 64, 64)
        steps += 1
        model.optimize_parameters(steps, 0, True)
    assert steps == 6 and model.get_current_log()["l_g_l1"] > 0


def test_demosaic_stage_is_bit_exact_against_the_oracle(cuda):
    """tr_bayer_round_trip against oracle.otf.demosaic_artifacts (realesrgan_model.py:312-363).  Byte work: bit-exact.  cv2 is absent, so the oracle's
    bilinear demosaicing is itself a restatement of OpenCV's published algorithm -- PARITY UNPINNED for this stage; what is pinned here is the
    reference's own part: the 8-bit truncation, which channel each site keeps, and that kept sites pass through unchanged."""
    from oracle import otf as OO
    from trainner_redux_amd.data import otf

    g = torch.Generator().manual_seed(12)
    for shape in [(2, 3, 16, 16), (1, 3, 7, 9), (3, 3, 3, 3), (1, 3, 34, 21)]:
        img = torch.rand(shape, generator=g) * 1.2 - 0.1                     # out-of-range values exercise the clamp
        out = otf.bayer_round_trip(img.to(cuda)).cpu()
        want = OO.demosaic_artifacts(img)
        assert torch.equal(out, want), shape
        q = (img.clamp(0, 1) * 255).to(torch.uint8).float() / 255.0         # img.astype("uint8") truncates
        h, w = shape[2:]
        assert torch.equal(out[:, 2, 2:h - 1:2, 2:w - 1:2], q[:, 2, 2:h - 1:2, 2:w - 1:2])       # interior sites keep their own colour
        assert torch.equal(out[:, 0, 1:h - 1:2, 1:w - 1:2], q[:, 0, 1:h - 1:2, 1:w - 1:2])
        assert torch.equal(out[:, 1, 1:h - 1:2, 2:w - 1:2], q[:, 1, 1:h - 1:2, 2:w - 1:2])
    flat = torch.full((1, 3, 8, 8), 0.5)
    assert torch.equal(otf.bayer_round_trip(flat.to(cuda)).cpu(), torch.full((1, 3, 8, 8), 127 / 255.0))      # a flat field stays flat
    with pytest.raises(ValueError):
        otf.bayer_round_trip(torch.rand(1, 1, 8, 8).to(cuda))
    with pytest.raises(RuntimeError, match="tr_bayer_round_trip"):
        otf.bayer_round_trip(torch.rand(1, 3, 2, 8).to(cuda))


def test_demosaic_gate_runs_in_the_pipeline(cuda):
    import random

    import numpy as np

    from trainner_redux_amd.data import otf

    deg = otf.OTFDegrader(_otf_opt(demosaic_prob=1.0), rng=np.random.default_rng(1), rnd=random.Random(1))
    gt = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(3)).to(cuda)
    ident = torch.zeros(1, 21, 21); ident[:, 10, 10] = 1.0
    staged = deg._before_codec_stages(gt, ident.to(cuda), ident.to(cuda))
    assert staged.shape[1] == 3 and torch.isfinite(staged).all()
