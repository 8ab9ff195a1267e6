"""Oracle restatement of the VGG19 perceptual head (test infrastructure, see oracle/__init__.py).

Follows /root/reference/traiNNer/losses/perceptual_fp16_loss.py:
  * layer order VGG19_LAYERS :12-50, default tapped conv layers + weights :52-58, channels :71
  * stage split, replicate padding on the very first conv only :262-274
  * (x - mean) / std with ImageNet statistics :281-286, :321-329
  * forward(): per tapped layer criterion1(x_feat, gt_feat.detach) * (1-alpha) * layer_weight :216-243,
    criterion "charbonnier" (default) = basic_loss.charbonnier_loss (eps 1e-12), "l1" = nn.L1Loss
  * PerceptualLoss forces fp32 (perceptual_loss.py:8-12); PerceptualFP16Loss follows autocast.
The topology (`torchvision.models.vgg19().features`) and the pretrained weights come from
torchvision, which is absent here and whose download is impossible offline: values are
*parity unpinned*; weights are synthetic (seeded He-normal) unless a torchvision-format
state dict (`features.N.weight`) is supplied.  Only the conv-layer criteria (l1 / charbonnier)
named by BASELINE.json configs are restated; pd/fd are out of scope.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
from torch.nn import functional as F

from .precision import FP32, Policy

VGG19_LAYERS = []
for _b, _n in enumerate((2, 2, 4, 4, 4), start=1):
    for _i in range(1, _n + 1):
        VGG19_LAYERS += [f"conv{_b}_{_i}", f"relu{_b}_{_i}"]
    VGG19_LAYERS.append(f"pool{_b}")
VGG19_CHANNELS = (64, 128, 256, 512, 512)
CONV_LAYER_WEIGHTS = OrderedDict(conv1_2=0.1, conv2_2=0.1, conv3_4=1.0, conv4_4=1.0, conv5_4=1.0)
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def conv_table():
    """[(layer name, torchvision features index, cin, cout)] in order."""
    t, cin = [], 3
    for idx, name in enumerate(VGG19_LAYERS):
        if name.startswith("conv"):
            cout = VGG19_CHANNELS[int(name[4]) - 1]
            t.append((name, idx, cin, cout))
            cin = cout
    return t


def synthetic_state_dict(seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for _, idx, cin, cout in conv_table():
        sd[f"features.{idx}.weight"] = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9))
        sd[f"features.{idx}.bias"] = torch.randn(cout, generator=g) * 0.05
    return sd


def features(sd, x, layer_names, policy: Policy = FP32):
    """dict layer -> activation, taken right after the named layer (pre-ReLU for conv names)."""
    want = set(layer_names)
    last = max(VGG19_LAYERS.index(n) for n in want)
    mean = x.new_tensor(MEAN).view(1, 3, 1, 1)
    std = x.new_tensor(STD).view(1, 3, 1, 1)
    h = (x - mean) / std
    out = OrderedDict()
    for idx, name in enumerate(VGG19_LAYERS[: last + 1]):
        if name.startswith("conv"):
            w, b = sd[f"features.{idx}.weight"], sd[f"features.{idx}.bias"]
            if idx == 0:  # first conv of the first stage is replicate-padded (:272-274)
                h = policy.conv2d(F.pad(h, (1, 1, 1, 1), mode="replicate"), w, b, 1, 0)
            else:
                h = policy.conv2d(h, w, b, 1, 1)
        elif name.startswith("relu"):
            h = F.relu(h)
        else:
            h = F.max_pool2d(h, 2, 2)
        if name in want:
            out[name] = h
    return out


def perceptual_loss(sd, x, gt, layer_weights=None, criterion="charbonnier", policy: Policy = FP32):
    """-> OrderedDict layer -> weighted scalar (perceptual_fp16_loss.py:216-243, alpha = 0 for conv layers)."""
    lw = CONV_LAYER_WEIGHTS if layer_weights is None else layer_weights
    fx = features(sd, x, lw.keys(), policy)
    with torch.no_grad():
        fg = features(sd, gt, lw.keys(), policy)
    losses = OrderedDict()
    for k in fx:
        d = fx[k] - fg[k]
        if criterion == "charbonnier":
            v = torch.sqrt(d * d + 1e-12).mean()
        elif criterion == "l1":
            v = d.abs().mean()
        else:
            raise NotImplementedError(criterion)
        losses[k] = v * lw[k]
    return losses
