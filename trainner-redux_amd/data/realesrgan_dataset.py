"""RealESRGANDataset: GT images + the per-sample blur / sinc kernels the OTF feed consumes (`kernel1`, `kernel2`, `sinc_kernel`, 21 x 21).

Mirrors /root/reference/traiNNer/data/realesrgan_dataset.py:24-226 -- same registry name, option keys, item keys and the same order of
random draws (augment flags, crop position, then per kernel: `random.choice` of the size, the sinc coin, the kernel's own draws).  Decoding
uses PIL (pyvips is not in this image); folder scanning only (lmdb / meta-info raise).  Reference quirks kept: `augment_vips(img, use_hflip,
use_rot)` passes use_rot as the VERTICAL-flip switch and leaves rot90 on its default True (:131); images smaller than gt_size + 32 are
zero-padded at the bottom / right (:139-143).
"""
from __future__ import annotations

import math
import random

import numpy as np
import torch
from torch.utils import data

from ..utils.registry import DATASET_REGISTRY
from .kernels import circular_lowpass_kernel, random_mixed_kernels
from .paired_image_dataset import read_rgb_u8, scan_folder


@DATASET_REGISTRY.register()
class RealESRGANDataset(data.Dataset):
    def __init__(self, opt, rng: np.random.Generator | None = None) -> None:
        super().__init__()
        self.opt = opt
        get = (lambda k, d=None: opt.get(k, d)) if isinstance(opt, dict) else (lambda k, d=None: getattr(opt, k, d))
        self._get = get
        io = get("io_backend") or {"type": "disk"}
        if io.get("type", "disk") != "disk" or get("meta_info") is not None:
            raise NotImplementedError("RealESRGANDataset (HIP build): only the folder-scanning disk backend is built")
        gt = get("dataroot_gt")
        assert gt is not None
        self.gt_folders = gt if isinstance(gt, list) else [gt]
        self.paths = []
        for folder in self.gt_folders:
            self.paths.extend(scan_folder(folder))
        self.rng = rng if rng is not None else np.random.default_rng(get("manual_seed", 0) or 0)
        self.kernel_range = list(range(get("kernel_range")[0], get("kernel_range")[1] + 1, 2))
        self.kernel_range2 = list(range(get("kernel_range2")[0], get("kernel_range2")[1] + 1, 2))
        self.final_kernel_range = list(range(get("final_kernel_range")[0], get("final_kernel_range")[1] + 1, 2))
        self.pulse_tensor = torch.zeros(21, 21).float()       # convolving with the pulse changes nothing
        self.pulse_tensor[10, 10] = 1

    def __len__(self) -> int:
        return len(self.paths)

    @property
    def label(self) -> str:
        return "images"

    def _kernel(self, sizes, sinc_prob, kernel_list, kernel_prob, blur_sigma, betag, betap) -> np.ndarray:
        rng = self.rng
        ks = random.choice(sizes)
        if rng.uniform() < sinc_prob:
            omega_c = rng.uniform(np.pi / 3, np.pi) if ks < 13 else rng.uniform(np.pi / 5, np.pi)     # tuned for sizes in [7, 21]
            k = circular_lowpass_kernel(omega_c, ks, pad_to=False)
        else:
            k = random_mixed_kernels(kernel_list, kernel_prob, ks, blur_sigma, blur_sigma, (-math.pi, math.pi), betag, betap, noise_range=None, rng=rng)
        p = (21 - ks) // 2
        return np.pad(k, ((p, p), (p, p)))

    def __getitem__(self, index: int):
        get = self._get
        gt_path = self.paths[index]
        assert get("use_hflip") is not None and get("use_rot") is not None
        img = read_rgb_u8(gt_path)
        hflip = get("use_hflip") and random.random() < 0.5
        vflip = get("use_rot") and random.random() < 0.5
        rot90 = random.random() < 0.5                          # augment_vips' default rot90=True (see the module docstring)
        if hflip:
            img = img[:, ::-1]
        if vflip:
            img = img[::-1]
        if rot90:
            img = np.rot90(img, k=-1)
        h, w = img.shape[:2]
        assert get("gt_size") is not None
        cps = get("gt_size") + 32
        if h < cps or w < cps:
            img = np.pad(img, ((0, max(0, cps - h)), (0, max(0, cps - w)), (0, 0)))
            h, w = img.shape[:2]
        if w > cps or h > cps:
            y = random.randint(0, h - cps)
            x = random.randint(0, w - cps)
            img = img[y:y + cps, x:x + cps]
        kernel = self._kernel(self.kernel_range, get("sinc_prob"), get("kernel_list"), get("kernel_prob"), get("blur_sigma"), get("betag_range"),
                              get("betap_range"))
        kernel2 = self._kernel(self.kernel_range2, get("sinc_prob2"), get("kernel_list2"), get("kernel_prob2"), get("blur_sigma2"), get("betag_range2"),
                               get("betap_range2"))
        if self.rng.uniform() < get("final_sinc_prob"):
            ks = random.choice(self.final_kernel_range)
            sinc_kernel = torch.FloatTensor(circular_lowpass_kernel(self.rng.uniform(np.pi / 3, np.pi), ks, pad_to=21))
        else:
            sinc_kernel = self.pulse_tensor
        gt_t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div_(255.0)
        return {"gt": gt_t, "kernel1": torch.FloatTensor(kernel), "kernel2": torch.FloatTensor(kernel2), "sinc_kernel": sinc_kernel, "gt_path": gt_path}
