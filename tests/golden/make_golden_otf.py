#!/usr/bin/env python3
"""Generates tests/golden/otf_degradations.pt from /root/reference/traiNNer/data/degradations.py loaded by path (run HERE only).

Nothing from the reference is copied: the module is imported under stub parent packages -- `cv2` (absent; only its JPEG helpers use it) is an
empty stub, `torchvision.transforms.functional.rgb_to_grayscale` (absent) a placeholder, `traiNNer.utils.RNG` a two-line stand-in with the
same numpy Generator semantics (utils/rng.py:7-25) -- and only input / output arrays are written:
  kernels      random_mixed_kernels for every kernel type x 3 sizes under seeded numpy / python RNGs (with and without kernel noise),
               circular_lowpass_kernel (padded and not)
  resize       resize_pt for bilinear / bicubic / nearest-exact / lanczos on seeded images (size= and scale_factor= forms)
  color        utils/color_util.py rgb2pixelformat_pt / pixelformat2rgb_pt / ycbcr2rgb_pt for "yuv444" (BT.601)
"""
import importlib.util
import math
import os
import random
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


class RNG:
    _rng = None

    @classmethod
    def get_rng(cls):
        return cls._rng


def main():
    for n in ("traiNNer", "traiNNer.utils", "cv2", "torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        _stub(n)
    sys.modules["traiNNer.utils"].RNG = RNG
    sys.modules["torchvision.transforms.functional"].rgb_to_grayscale = lambda *a, **k: None
    spec = importlib.util.spec_from_file_location("traiNNer.data.degradations", os.path.join(REF, "traiNNer/data/degradations.py"))
    D = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(D)

    kinds = ["iso", "aniso", "generalized_iso", "generalized_aniso", "plateau_iso", "plateau_aniso"]
    kernels = {}
    for ki, kind in enumerate(kinds):
        for size in (7, 13, 21):
            for noise in (None, (0.75, 1.25)):
                seed = 100 * ki + size + (1 if noise else 0)
                RNG._rng = np.random.default_rng(seed)
                random.seed(seed)
                k = D.random_mixed_kernels([kind], [1.0], size, (0.2, 3.0), (0.2, 3.0), (-math.pi, math.pi), (0.5, 4.0), (1.0, 2.0), noise_range=noise)
                kernels[f"{kind}/{size}/{'noise' if noise else 'plain'}"] = dict(seed=seed, kernel=torch.from_numpy(np.asarray(k, np.float64)))
    RNG._rng = np.random.default_rng(7)
    random.seed(7)
    mixed = [torch.from_numpy(D.random_mixed_kernels(kinds, [0.45, 0.25, 0.12, 0.03, 0.12, 0.03], 15, (0.2, 2.0), (0.2, 2.0), (-math.pi, math.pi),
                                                    (0.5, 4.0), (1.0, 2.0), noise_range=None)) for _ in range(12)]
    sinc = {f"{c:.3f}/{s}/{p}": torch.from_numpy(D.circular_lowpass_kernel(c, s, pad_to=p)) for c, s, p in
            ((math.pi / 3, 7, 21), (2.1, 13, 0), (math.pi, 21, 21), (0.9, 5, 21))}

    resize = {}
    g = torch.Generator().manual_seed(11)
    for tag, shape, kw in (("down4", (2, 3, 64, 48), dict(size=(16, 12))), ("odd", (1, 3, 37, 53), dict(size=(23, 31))), ("up", (1, 1, 20, 24), dict(size=(33, 40))),
                           ("mixed", (1, 3, 40, 40), dict(size=(40, 17))), ("factor", (1, 3, 30, 42), dict(scale_factor=0.4))):
        img = torch.rand(*shape, generator=g)
        resize[tag] = dict(img=img, kw=kw, out={m: D.resize_pt(img, mode=m, **kw) for m in ("bilinear", "bicubic", "nearest-exact", "lanczos")})
    # utils/color_util.py (needs only the PixelFormat name from utils/types.py, which wants Python 3.11: a stub carries it)
    _stub("traiNNer.utils.types").PixelFormat = str
    spec = importlib.util.spec_from_file_location("traiNNer.utils.color_util", os.path.join(REF, "traiNNer/utils/color_util.py"))
    CU = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(CU)
    rgb = torch.rand(2, 3, 13, 17, generator=g)
    ycc = CU.rgb2pixelformat_pt(rgb, "yuv444")
    color = dict(rgb=rgb, ycbcr=ycc, back=CU.pixelformat2rgb_pt(ycc, None, "yuv444"), any_ycbcr=torch.rand(2, 3, 13, 17, generator=g))
    color["any_rgb"] = CU.ycbcr2rgb_pt(color["any_ycbcr"])
    torch.save(dict(kernels=kernels, mixed=dict(seed=7, kernels=mixed), sinc=sinc, resize=resize, color=color), os.path.join(OUT, "otf_degradations.pt"))
    print("wrote otf_degradations.pt", len(kernels), "kernels")


if __name__ == "__main__":
    main()
