"""Feed side: PairedImageDataset against a plain numpy restatement of the reference's call order
(paired_image_dataset.py:86-152, transforms.py:214-273,331-407).  Parity unpinned: the reference's transforms need cv2 / pyvips."""
import os
import random

import numpy as np
import pytest
import torch


def _make_pairs(root, n=4, scale=4, lq_hw=(24, 40)):
    from PIL import Image

    rng = np.random.default_rng(0)
    os.makedirs(root / "lq"), os.makedirs(root / "gt")
    imgs = []
    for i in range(n):
        gt = rng.integers(0, 256, size=(lq_hw[0] * scale, lq_hw[1] * scale, 3), dtype=np.uint8)
        lq = gt[::scale, ::scale].copy()
        Image.fromarray(gt).save(root / "gt" / f"img{i:02d}.png")
        Image.fromarray(lq).save(root / "lq" / f"img{i:02d}.png")
        imgs.append((gt, lq))
    return imgs


def _opt(root, phase="train", **kw):
    d = dict(name="t", type="pairedimagedataset", dataroot_lq=[str(root / "lq")], dataroot_gt=[str(root / "gt")], io_backend={"type": "disk"},
             scale=4, phase=phase, gt_size=32, use_hflip=True, use_rot=True, filename_tmpl="{}", mean=None, std=None, color=None, meta_info=None)
    d.update(kw)
    return d


def test_paired_dataset_matches_reference_call_order(tmp_path):
    from trainner_redux_amd.data import PairedImageDataset

    imgs = _make_pairs(tmp_path)
    ds = PairedImageDataset(_opt(tmp_path), as_uint8=True)
    assert len(ds) == 4 and ds.label == "image pairs"
    for seed in range(6):
        idx = seed % 4
        random.seed(seed)
        item = ds[idx]
        # restatement: three flag draws (hflip, vflip, rot90 clockwise), THEN top, left of the LQ patch on the augmented image
        random.seed(seed)
        hf, vf, r9 = random.random() < 0.5, random.random() < 0.5, random.random() < 0.5
        gt, lq = imgs[idx]

        def aug(a):
            if hf:
                a = a[:, ::-1]
            if vf:
                a = a[::-1]
            if r9:
                a = np.transpose(a[::-1], (1, 0, 2))      # 90 degrees clockwise
            return a

        gt, lq = aug(gt), aug(lq)
        y = random.randint(0, lq.shape[0] - 8)
        x = random.randint(0, lq.shape[1] - 8)
        assert np.array_equal(item["lq"].numpy(), lq[y:y + 8, x:x + 8])
        assert np.array_equal(item["gt"].numpy(), gt[4 * y:4 * y + 32, 4 * x:4 * x + 32])
        assert item["lq"].dtype == torch.uint8 and item["gt_path"].endswith(f"img{idx:02d}.png")
        if not (hf or vf or r9):      # (decimation picks each cell's top-left pixel, so this identity only holds un-flipped)
            assert np.array_equal(item["gt"].numpy()[::4, ::4], item["lq"].numpy())


def test_paired_dataset_datafeed_contract_and_val_phase(tmp_path):
    from trainner_redux_amd.data import PairedImageDataset

    imgs = _make_pairs(tmp_path)
    random.seed(3)
    a = PairedImageDataset(_opt(tmp_path), as_uint8=True)[1]
    random.seed(3)
    b = PairedImageDataset(_opt(tmp_path))[1]                           # the reference's DataFeed: fp32 CHW in [0, 1]
    assert b["lq"].dtype == torch.float32 and b["lq"].shape == (3, 8, 8) and b["gt"].shape == (3, 32, 32)
    assert torch.equal(b["gt"], a["gt"].permute(2, 0, 1).float() / 255.0)
    random.seed(3)
    c = PairedImageDataset(_opt(tmp_path, mean=[0.5, 0.5, 0.5], std=[0.25, 0.25, 0.25]))[1]
    assert torch.allclose(c["lq"], (b["lq"] - 0.5) / 0.25, atol=1e-6)
    v = PairedImageDataset(_opt(tmp_path, phase="val"), as_uint8=True)[2]   # validation: whole images, GT cropped to scale x LQ
    assert np.array_equal(v["gt"].numpy(), imgs[2][0]) and np.array_equal(v["lq"].numpy(), imgs[2][1])


def test_paired_dataset_errors(tmp_path):
    from PIL import Image

    from trainner_redux_amd.data import PairedImageDataset
    from trainner_redux_amd.data.paired_image_dataset import paired_random_crop

    _make_pairs(tmp_path, n=2)
    with pytest.raises(ValueError, match="Scale mismatches"):
        paired_random_crop(np.zeros((33, 32, 3), np.uint8), np.zeros((8, 8, 3), np.uint8), 32, 4)
    with pytest.raises(ValueError, match="smaller than patch size"):
        paired_random_crop(np.zeros((16, 16, 3), np.uint8), np.zeros((4, 4, 3), np.uint8), 32, 4)
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(tmp_path / "gt" / "extra.png")
    with pytest.raises(ValueError, match="different numbers"):
        PairedImageDataset(_opt(tmp_path))
    with pytest.raises(NotImplementedError):
        PairedImageDataset(_opt(tmp_path, io_backend={"type": "lmdb"}))


def test_loader_seeding_follows_the_reference_formula():
    """data/__init__.py:168-178: worker seed = num_workers * rank + worker_id + seed; with no workers the loading process itself is seeded
    as worker 0 of its rank (= train.py:283's `manual_seed + rank`), so two ranks never draw the same crop / flip stream."""
    import random

    import numpy as np
    import torch

    from trainner_redux_amd.data.prefetch import build_dataloader, worker_seed

    assert worker_seed(1024, 8, 3, 5) == 8 * 3 + 5 + 1024
    ds = torch.utils.data.TensorDataset(torch.zeros(4, 1))
    draws = []
    for rank in (0, 1):
        build_dataloader(ds, 2, 0, seed=77, rank=rank)
        draws.append((random.random(), float(np.random.rand())))
    random.seed(77); np.random.seed(77)
    assert draws[0] == (random.random(), float(np.random.rand()))
    random.seed(78); np.random.seed(78)
    assert draws[1] == (random.random(), float(np.random.rand())) and draws[0] != draws[1]
