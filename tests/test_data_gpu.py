"""Feed side on the GPU: uint8 -> fp32 NCHW kernel, the pinned double-buffered prefetcher, and `SRModel.feed_data` on uint8 patches."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_u8hwc_to_nchw_kernel(cuda):
    from trainner_redux_amd import ops

    g = torch.Generator().manual_seed(0)
    src = torch.randint(0, 256, (3, 20, 28, 3), dtype=torch.uint8, generator=g)
    out = torch.empty(3, 3, 20, 28, device=cuda)
    sd = src.to(cuda)
    ops.u8hwc_to_nchw(sd, out)
    ref = src.permute(0, 3, 1, 2).float() / 255.0
    assert torch.equal(out.cpu(), ref)                          # bit-identical to the reference's fp32 `/ 255.` (img_util.py img2tensor)
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    ops.u8hwc_to_nchw(sd, out, mean, std)
    ref2 = (ref - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    assert torch.allclose(out.cpu(), ref2, atol=2e-6)


def test_device_prefetcher_double_buffer(cuda, tmp_path):
    from tests.test_data_cpu import _make_pairs, _opt
    from trainner_redux_amd.data import DevicePrefetcher, PairedImageDataset, build_dataloader

    _make_pairs(tmp_path, n=8)
    ds = PairedImageDataset(_opt(tmp_path), as_uint8=True)
    random.seed(11)
    expect = [ds[i] for i in range(8)]
    random.seed(11)
    loader = build_dataloader(ds, batch_size=2, num_workers=0, sampler=torch.utils.data.SequentialSampler(ds))
    pf = DevicePrefetcher(loader, cuda)
    got = []
    while True:
        b = pf.next()
        if b is None:
            break
        assert b["lq"].is_cuda and b["lq"].dtype == torch.float32 and b["lq"].shape == (2, 3, 8, 8) and b["gt"].shape == (2, 3, 32, 32)
        got.append(b)
    assert len(got) == 4
    for k, b in enumerate(got):
        for j in range(2):
            ref = expect[2 * k + j]
            assert torch.equal(b["gt"][j].cpu(), ref["gt"].permute(2, 0, 1).float() / 255.0)
            assert b["gt_path"][j] == ref["gt_path"]
    pf.reset()                                                  # a second epoch starts cleanly
    assert pf.next() is not None


def test_feed_data_accepts_uint8_patches(cuda):
    from tests.test_sr_model_gpu import _opt
    from trainner_redux_amd.models import SRModel

    model = SRModel(_opt([dict(type="l1loss", loss_weight=1.0)], blocks=1))
    g = torch.Generator().manual_seed(2)
    gt = torch.randint(0, 256, (2, 64, 64, 3), dtype=torch.uint8, generator=g)
    lq = gt[:, ::4, ::4].contiguous()
    model.feed_data({"lq": lq, "gt": gt})
    assert model.lq.shape == (2, 3, 16, 16) and model.gt.dtype == torch.float32
    assert torch.equal(model.gt.cpu(), gt.permute(0, 3, 1, 2).float() / 255.0)
    model.optimize_parameters(1, 0, True)
    assert np.isfinite(model.get_current_log()["l_g_l1"])
