"""PairedImageDataset: folder-scanned LQ / GT pairs with paired flip / rotate / crop (train) -- the step in front of `feed_data`.

Mirrors /root/reference/traiNNer/data/paired_image_dataset.py:18-152 and transforms.py:214-273,331-407 (same option keys, same
registry name, same RANDOM-NUMBER CALL ORDER: hflip, vflip, rot90 draws of `random.random() < 0.5`, then `random.randint` for the
crop's top and left ON THE AUGMENTED image), so a seeded run picks the same patches as the reference.  Differences, by design:
  * decoding uses PIL (pyvips / OpenCV are not in this image); 8-bit RGB only (`color: y`, lmdb and meta-info files raise);
  * `as_uint8=True` (what `DevicePrefetcher` asks for) returns uint8 HWC patches: the host never builds fp32 CHW tensors, the
    H2D copy moves 4x fewer bytes and `tr_u8hwc_to_nchw` produces the trainer's fp32 NCHW [0,1] batch on the GPU.  With
    `as_uint8=False` the item is the reference's DataFeed (fp32 CHW in [0,1], utils/types.py:10-17).
Parity unpinned: the reference's transforms import cv2 / pyvips and cannot run here; tests check the call order and geometry against a
plain numpy restatement on synthetic PNGs.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch
from torch.utils import data

from ..utils.registry import DATASET_REGISTRY

IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp", ".tif", ".tiff")


def scan_folder(root: str) -> list[str]:
    out = []
    for dp, _, fns in sorted(os.walk(root)):
        out += [os.path.join(dp, f) for f in sorted(fns) if f.lower().endswith(IMG_EXT)]
    return out


def paired_paths_from_folder(folders, keys, filename_tmpl="{}"):
    """data_util.py `paired_paths_from_folder`: GT files drive the pairing; the LQ name is filename_tmpl.format(basename) + same ext."""
    (lq_roots, gt_roots), (lq_key, gt_key) = folders, keys
    if len(lq_roots) != len(gt_roots):
        raise ValueError("dataroot_lq and dataroot_gt must list the same number of folders")
    paths = []
    for lq_root, gt_root in zip(lq_roots, gt_roots):
        lq_all = {os.path.relpath(p, lq_root) for p in scan_folder(lq_root)}
        gts = scan_folder(gt_root)
        if len(lq_all) != len(gts):
            raise ValueError(f"{lq_key} and {gt_key} folders hold different numbers of images: {len(lq_all)}, {len(gts)}")
        for g in gts:
            rel = os.path.relpath(g, gt_root)
            base, ext = os.path.splitext(rel)
            lq_rel = filename_tmpl.format(base) + ext
            if lq_rel not in lq_all:
                raise ValueError(f"{lq_rel} is not in {lq_key} folder {lq_root}")
            paths.append({f"{lq_key}_path": os.path.join(lq_root, lq_rel), f"{gt_key}_path": g})
    return paths


def read_rgb_u8(path: str) -> np.ndarray:
    from PIL import Image

    with Image.open(path) as im:
        return np.array(im.convert("RGB"))       # a writable copy (np.asarray of a PIL image is read-only)


def augment_pair(gt: np.ndarray, lq: np.ndarray, hflip: bool, vflip: bool, rot90: bool):
    """transforms.py:331-407: three draws in this order, the same flags for both images; rot90 is 90 degrees CLOCKWISE (vips rot90)."""
    hflip = hflip and random.random() < 0.5
    vflip = vflip and random.random() < 0.5
    rot90 = rot90 and random.random() < 0.5

    def f(img):
        if hflip:
            img = img[:, ::-1]
        if vflip:
            img = img[::-1]
        if rot90:
            img = np.rot90(img, k=-1)
        return img

    return f(gt), f(lq), (hflip, vflip, rot90)


def paired_random_crop(gt: np.ndarray, lq: np.ndarray, gt_patch_size: int, scale: int, gt_path: str = ""):
    """transforms.py:214-273: randint for top, then left, on the LQ grid; GT patch at scale x the coordinates."""
    h_lq, w_lq = lq.shape[:2]
    h_gt, w_gt = gt.shape[:2]
    ps = gt_patch_size // scale
    if h_gt != h_lq * scale or w_gt != w_lq * scale:
        raise ValueError(f"Scale mismatches. GT ({h_gt}, {w_gt}) is not {scale}x multiplication of LQ ({h_lq}, {w_lq}). {gt_path}")
    if h_lq < ps or w_lq < ps:
        raise ValueError(f"LQ ({h_lq}, {w_lq}) is smaller than patch size ({ps}, {ps}). Please remove {gt_path}.")
    y = random.randint(0, h_lq - ps)
    x = random.randint(0, w_lq - ps)
    return gt[y * scale:y * scale + gt_patch_size, x * scale:x * scale + gt_patch_size], lq[y:y + ps, x:x + ps], (y, x)


@DATASET_REGISTRY.register()
class PairedImageDataset(data.Dataset):
    def __init__(self, opt, as_uint8: bool = False) -> None:
        super().__init__()
        self.opt, self.as_uint8 = opt, as_uint8
        get = (lambda k, d=None: opt.get(k, d)) if isinstance(opt, dict) else (lambda k, d=None: getattr(opt, k, d))
        self._get = get
        io = get("io_backend") or {"type": "disk"}
        if io.get("type", "disk") != "disk" or get("meta_info") is not None:
            raise NotImplementedError("PairedImageDataset (HIP build): only the folder-scanning disk backend is built")
        if get("color") == "y":
            raise NotImplementedError("PairedImageDataset (HIP build): color: y is not built")
        lq, gt = get("dataroot_lq"), get("dataroot_gt")
        assert isinstance(lq, list), f"dataroot_lq must be defined for dataset {get('name')}"
        assert isinstance(gt, list), f"dataroot_gt must be defined for dataset {get('name')}"
        self.mean, self.std = get("mean"), get("std")
        self.paths = paired_paths_from_folder((lq, gt), ("lq", "gt"), get("filename_tmpl") or "{}")

    def __len__(self) -> int:
        return len(self.paths)

    @property
    def label(self) -> str:
        return "image pairs"

    def __getitem__(self, index: int):
        get = self._get
        scale = get("scale")
        assert scale is not None
        p = self.paths[index]
        gt, lq = read_rgb_u8(p["gt_path"]), read_rgb_u8(p["lq_path"])
        if get("phase") == "train":
            assert get("gt_size") is not None and get("use_hflip") is not None and get("use_rot") is not None
            gt, lq, _ = augment_pair(gt, lq, get("use_hflip"), get("use_rot"), get("use_rot"))
            gt, lq, _ = paired_random_crop(gt, lq, get("gt_size"), scale, p["gt_path"])
        else:
            gt = gt[0:lq.shape[0] * scale, 0:lq.shape[1] * scale]
        gt_t, lq_t = torch.from_numpy(np.ascontiguousarray(gt).copy()), torch.from_numpy(np.ascontiguousarray(lq).copy())
        if not self.as_uint8:
            gt_t = gt_t.permute(2, 0, 1).float().div_(255.0)
            lq_t = lq_t.permute(2, 0, 1).float().div_(255.0)
            if self.mean is not None and self.std is not None:
                m, s = torch.tensor(self.mean).view(-1, 1, 1), torch.tensor(self.std).view(-1, 1, 1)
                lq_t.sub_(m).div_(s)
                gt_t.sub_(m).div_(s)
        return {"lq": lq_t, "gt": gt_t, "lq_path": p["lq_path"], "gt_path": p["gt_path"]}
