"""Host side of the OTF resize: the (index, weight) tables built by trainner_redux_amd/data/otf.py, applied with numpy, must reproduce
torch's own F.interpolate -- the function the reference's resize_pt calls (data/degradations.py:996-1021) -- and the Lanczos path of its
`_resample`.  No GPU: this pins the table construction; tests/test_otf_gpu.py then holds the kernel to the same oracle."""
import numpy as np
import pytest
import torch

from oracle import otf as OO


def _apply(img, axis, tables):
    idx, wts = tables
    a = img.numpy()
    g = np.take(a, idx, axis=2 + axis)                       # (..., out, taps, ...) inserted at the axis
    if axis == 0:
        return torch.from_numpy(np.einsum("bcotw,ot->bcow", g, wts).astype(np.float32))
    return torch.from_numpy(np.einsum("bchot,ot->bcho", g, wts).astype(np.float32))


@pytest.mark.parametrize("mode", ["bilinear", "bicubic", "nearest-exact", "nearest", "lanczos"])
@pytest.mark.parametrize("shape,size", [((2, 3, 64, 48), (16, 12)), ((1, 3, 37, 53), (23, 31)), ((1, 1, 20, 24), (33, 40)), ((1, 3, 40, 40), (40, 17))])
def test_resize_tables_reproduce_torch_interpolate(mode, shape, size):
    from trainner_redux_amd.data import otf

    img = torch.rand(*shape, generator=torch.Generator().manual_seed(7))
    h, w = shape[2:]
    dh, dw = size
    if mode == "lanczos":
        ref = OO.resize_pt(img, "lanczos", size=size)
        x = img
        if dh < h:
            x = _apply(x, 0, otf._tables_reflect_conv(h, otf._lanczos_kernel(dh / h)))
        if dw < w:
            x = _apply(x, 1, otf._tables_reflect_conv(w, otf._lanczos_kernel(dw / w)))
        x = _apply(_apply(x, 1, otf._tables_bicubic(w, dw)), 0, otf._tables_bicubic(h, dh)).clamp(0, 1)
    elif mode == "nearest":
        ref = torch.nn.functional.interpolate(img, size=size, mode="nearest")
        x = _apply(_apply(img, 1, otf._tables_nearest(w, dw, False)), 0, otf._tables_nearest(h, dh, False))
    elif mode == "nearest-exact":
        ref = OO.resize_pt(img, mode, size=size)
        x = _apply(_apply(img, 1, otf._tables_nearest(w, dw, True)), 0, otf._tables_nearest(h, dh, True))
    else:
        ref = OO.resize_pt(img, mode, size=size)
        x = _apply(_apply(img, 1, otf._tables_aa(mode, w, dw)), 0, otf._tables_aa(mode, h, dh)).clamp(0, 1)
    assert x.shape == ref.shape
    err = (x - ref).abs().max().item()
    assert err < (1e-7 if mode.startswith("nearest") else 3e-6), err


def test_paired_crop_checks_and_motion_kernel():
    from trainner_redux_amd.data import otf

    with pytest.raises(ValueError, match="Scale mismatches"):
        otf.paired_random_crop(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64), 32, 4)
    with pytest.raises(ValueError, match="smaller than patch size"):
        otf.paired_random_crop(torch.zeros(1, 3, 16, 16), torch.zeros(1, 3, 4, 4), 32, 4)
    k = otf.motion_blur_kernel(7, 0.0)
    assert k.shape == (7, 7) and abs(k.sum().item() - 1) < 1e-6 and (k[3] > 0).all() and k[0].sum() == 0     # angle 0: the centre row (x = i - c = 0)
    with pytest.raises(ValueError, match="scale_factor or size is required"):
        OO.resize_pt(torch.zeros(1, 3, 8, 8), "bicubic")


GOLD = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "otf_degradations.pt")


def test_kernel_synthesis_matches_the_reference_functions():
    """data/kernels.py under the same seeds as the fixture's generator (tests/golden/make_golden_otf.py ran the reference's own
    random_mixed_kernels / circular_lowpass_kernel): every family, three sizes, with and without kernel noise; a mixed stream of 12."""
    import math
    import random

    from trainner_redux_amd.data import kernels as K

    fix = torch.load(GOLD, weights_only=True)
    for tag, e in fix["kernels"].items():
        kind, size, noise = tag.split("/")
        rnd = random.Random(e["seed"])
        k = K.random_mixed_kernels([kind], [1.0], int(size), (0.2, 3.0), (0.2, 3.0), (-math.pi, math.pi), (0.5, 4.0), (1.0, 2.0),
                                   noise_range=(0.75, 1.25) if noise == "noise" else None, rng=np.random.default_rng(e["seed"]), rnd=rnd)
        assert np.abs(k - e["kernel"].numpy()).max() < 1e-12, tag
    rng, rnd = np.random.default_rng(7), random.Random(7)
    kinds = ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"]
    for ref in fix["mixed"]["kernels"]:
        k = K.random_mixed_kernels(kinds, [0.45, 0.25, 0.12, 0.03, 0.12, 0.03], 15, (0.2, 2.0), (0.2, 2.0), (-math.pi, math.pi), (0.5, 4.0), (1.0, 2.0),
                                   rng=rng, rnd=rnd)
        assert np.abs(k - ref.numpy()).max() < 1e-12
    for c, s, p in ((math.pi / 3, 7, 21), (2.1, 13, 0), (math.pi, 21, 21), (0.9, 5, 21)):      # the generator's list; keys carry 3 decimals
        ref = fix["sinc"][f"{c:.3f}/{s}/{p}"].numpy()
        assert np.abs(K.circular_lowpass_kernel(c, s, pad_to=p) - ref).max() < 1e-12, (c, s, p)
    with pytest.raises(ValueError, match="Unknown kernel type"):
        K.random_mixed_kernels(["box"], [1.0], 7, rng=rng)


def test_oracle_and_tables_match_the_reference_resize_pt():
    """The reference's own resize_pt outputs (fixture): the oracle restatement is identical, the table-driven form agrees to fp32 noise."""
    from trainner_redux_amd.data import otf

    fix = torch.load(GOLD, weights_only=True)["resize"]
    for tag, e in fix.items():
        for mode, ref in e["out"].items():
            assert torch.equal(OO.resize_pt(e["img"], mode, **e["kw"]), ref), (tag, mode)
    e = fix["down4"]
    h, w, (dh, dw) = 64, 48, e["kw"]["size"]
    x = _apply(_apply(e["img"], 1, otf._tables_aa("bicubic", w, dw)), 0, otf._tables_aa("bicubic", h, dh)).clamp(0, 1)
    assert (x - e["out"]["bicubic"]).abs().max() < 3e-6


def test_realesrgan_dataset_items(tmp_path):
    import random

    from PIL import Image

    from trainner_redux_amd.data import RealESRGANDataset
    from trainner_redux_amd.utils.options import DatasetOptions

    g = np.random.default_rng(0)
    for i, (h, w) in enumerate([(120, 140), (60, 50)]):           # the second one is smaller than gt_size + 32: padded
        Image.fromarray(g.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(tmp_path / f"{i}.png")
    opt = DatasetOptions(dict(name="t", type="realesrgandataset", dataroot_gt=[str(tmp_path)], gt_size=64, sinc_prob=0.5, sinc_prob2=0.5, final_sinc_prob=0.5,
                              kernel_range=(7, 21), kernel_range2=(7, 21), final_kernel_range=(7, 21)))
    ds = RealESRGANDataset(opt, rng=np.random.default_rng(3))
    assert len(ds) == 2 and ds.label == "images"
    random.seed(5)
    for i in range(2):
        for _ in range(4):
            it = ds[i]
            assert set(it) == {"gt", "kernel1", "kernel2", "sinc_kernel", "gt_path"}          # realesrgan_dataset.py:213-219
            assert it["gt"].shape == (3, 96, 96) and it["gt"].dtype == torch.float32 and 0 <= it["gt"].min() and it["gt"].max() <= 1
            for k in ("kernel1", "kernel2", "sinc_kernel"):
                assert it[k].shape == (21, 21) and abs(it[k].sum().item() - 1) < 1e-4
    assert (ds[1]["gt"] == 0).float().mean() > 0.3                 # zero padding of the small image survives flips / rotations


def test_demosaic_oracle_hand_worked_case():
    """oracle.otf.demosaic_artifacts on a 4x4 image worked by hand (realesrgan_model.py:327-361; cv2's bilinear rule restated, unpinned)."""
    from oracle import otf as OO

    lv = torch.arange(48, dtype=torch.float32).reshape(1, 3, 4, 4) * 5            # 8-bit levels 0..235, all distinct
    out = (OO.demosaic_artifacts((lv + 0.4) / 255.0) * 255).round().int()         # +0.4: the uint8 cast truncates
    ch0, ch1, ch2 = lv[0, 0].int(), lv[0, 1].int(), lv[0, 2].int()
    m = torch.zeros(4, 4, dtype=torch.int32)
    m[0::2, 0::2], m[0::2, 1::2], m[1::2, 0::2], m[1::2, 1::2] = ch2[0::2, 0::2], ch1[0::2, 1::2], ch1[1::2, 0::2], ch0[1::2, 1::2]
    # (1, 1) keeps channel 0; green from the four edge neighbours, channel 2 from the four diagonal ones
    assert out[0, 0, 1, 1] == m[1, 1]
    assert out[0, 1, 1, 1] == (m[0, 1] + m[1, 0] + m[1, 2] + m[2, 1] + 2) // 4
    assert out[0, 2, 1, 1] == (m[0, 0] + m[0, 2] + m[2, 0] + m[2, 2] + 2) // 4
    # (1, 2) is a green site on an odd row: channel 0 beside it, channel 2 above / below
    assert out[0, 1, 1, 2] == m[1, 2]
    assert out[0, 0, 1, 2] == (m[1, 1] + m[1, 3] + 1) // 2
    assert out[0, 2, 1, 2] == (m[0, 2] + m[2, 2] + 1) // 2
    # borders repeat the nearest interior pixel
    assert torch.equal(out[0, :, 0, 0], out[0, :, 1, 1]) and torch.equal(out[0, :, 3, 3], out[0, :, 2, 2])
    assert torch.equal(out[0, :, 0, 2], out[0, :, 1, 2]) and torch.equal(out[0, :, 2, 0], out[0, :, 2, 1])
