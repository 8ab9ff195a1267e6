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
