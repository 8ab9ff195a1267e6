"""SPAN (Swift Parameter-free Attention Network) -- CPU restatement of /root/reference/traiNNer/archs/span_arch.py.

Test infrastructure only (see oracle/__init__.py).  PINNED: the reference's own module, loaded by path in this container, produced
tests/golden/span.pt (state dict, training-mode output, parameter gradients, eval-mode output, merged eval weights);
tests/test_oracle_cpu.py holds this restatement to those tensors.

  Conv3XC.forward     span_arch.py:197-210   training: pad 1 -> 1x1 (C -> gC) -> 3x3 valid (gC -> gC') -> 1x1 (gC' -> C') + 1x1 skip;
                                             eval: ONE 3x3 with the merged kernel of update_params() :160-195
  SPAB.forward        span_arch.py:235-248   three Conv3XC, SiLU between, out = (out3 + x) * (sigmoid(out3) - 0.5)
  SPAN.forward        span_arch.py:304-323   optional (x - mean) * img_range, conv_1, 6 SPAB, conv_2, 1x1 over cat[feat, conv_2(b6), b1, silu(out1) of block 6],
                                             3x3 + PixelShuffle
"""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch.nn import functional as F

from .precision import FP32, Policy

RGB_MEAN = (0.4488, 0.4371, 0.4040)


def hparams(feature_channels=48, scale=4, norm=False, img_range=255.0, num_in_ch=3, num_out_ch=3):
    return dict(feature_channels=feature_channels, scale=scale, norm=norm, img_range=img_range, num_in_ch=num_in_ch, num_out_ch=num_out_ch)


def conv3xc_names(hp):
    names = ["conv_1"]
    for b in range(1, 7):
        names += [f"block_{b}.c{k}_r" for k in (1, 2, 3)]
    return names + ["conv_2"]


def param_table(hp) -> "OrderedDict[str, tuple]":
    C, cin = hp["feature_channels"], hp["num_in_ch"]
    t = OrderedDict()

    def c3x(p, ci, co, gain=2):
        t[p + ".sk.weight"] = (co, ci, 1, 1); t[p + ".sk.bias"] = (co,)
        t[p + ".conv.0.weight"] = (ci * gain, ci, 1, 1); t[p + ".conv.0.bias"] = (ci * gain,)
        t[p + ".conv.1.weight"] = (co * gain, ci * gain, 3, 3); t[p + ".conv.1.bias"] = (co * gain,)
        t[p + ".conv.2.weight"] = (co, co * gain, 1, 1); t[p + ".conv.2.bias"] = (co,)
        t[p + ".eval_conv.weight"] = (co, ci, 3, 3); t[p + ".eval_conv.bias"] = (co,)

    c3x("conv_1", cin, C)
    for b in range(1, 7):
        for k in (1, 2, 3):
            c3x(f"block_{b}.c{k}_r", C, C)
    t["conv_cat.weight"] = (C, 4 * C, 1, 1); t["conv_cat.bias"] = (C,)
    c3x("conv_2", C, C)
    r = hp["scale"]
    t["upsampler.0.weight"] = (hp["num_out_ch"] * r * r, C, 3, 3); t["upsampler.0.bias"] = (hp["num_out_ch"] * r * r,)
    return t


def merged_kernel(sd, p):
    """Conv3XC.update_params (span_arch.py:160-195): the single 3x3 equivalent of the training-time branch pair."""
    w1, b1 = sd[p + ".conv.0.weight"], sd[p + ".conv.0.bias"]
    w2, b2 = sd[p + ".conv.1.weight"], sd[p + ".conv.1.bias"]
    w3, b3 = sd[p + ".conv.2.weight"], sd[p + ".conv.2.bias"]
    w = F.conv2d(w1.flip(2, 3).permute(1, 0, 2, 3), w2, padding=2, stride=1).flip(2, 3).permute(1, 0, 2, 3)
    b = (w2 * b1.reshape(1, -1, 1, 1)).sum((1, 2, 3)) + b2
    wc = F.conv2d(w.flip(2, 3).permute(1, 0, 2, 3), w3, padding=0, stride=1).flip(2, 3).permute(1, 0, 2, 3)
    bc = (w3 * b.reshape(1, -1, 1, 1)).sum((1, 2, 3)) + b3
    wc = wc + F.pad(sd[p + ".sk.weight"], [1, 1, 1, 1])
    return wc.contiguous(), (bc + sd[p + ".sk.bias"]).contiguous()


def conv3xc(sd, p, x, policy: Policy, train: bool):
    if train:
        t = policy.conv2d(F.pad(x, (1, 1, 1, 1)), sd[p + ".conv.0.weight"], sd[p + ".conv.0.bias"], 1, 0)
        t = policy.conv2d(t, sd[p + ".conv.1.weight"], sd[p + ".conv.1.bias"], 1, 0)
        t = policy.conv2d(t, sd[p + ".conv.2.weight"], sd[p + ".conv.2.bias"], 1, 0)
        return t + policy.conv2d(x, sd[p + ".sk.weight"], sd[p + ".sk.bias"], 1, 0)
    w, b = merged_kernel(sd, p)
    return policy.conv2d(x, w, b, 1, 1)


def spab(sd, p, x, policy, train):
    # act1 is nn.SiLU(inplace=True) (span_arch.py:232,237): `out1` IS the activated tensor by the time the block returns it, so the
    # feature the network concatenates as `out_b5_2` is silu(c1_r(x)) -- pinned by tests/golden/span.pt
    out1 = F.silu(conv3xc(sd, p + ".c1_r", x, policy, train))
    out2 = conv3xc(sd, p + ".c2_r", out1, policy, train)
    out3 = conv3xc(sd, p + ".c3_r", F.silu(out2), policy, train)
    att = torch.sigmoid(out3) - 0.5
    return (out3 + x) * att, out1


def forward(sd, x, hp, policy: Policy = FP32, train: bool = True):
    if hp["norm"]:
        x = (x - x.new_tensor(RGB_MEAN).view(1, 3, 1, 1)) * hp["img_range"]
    feat = conv3xc(sd, "conv_1", x, policy, train)
    b1, _ = spab(sd, "block_1", feat, policy, train)
    b2, _ = spab(sd, "block_2", b1, policy, train)
    b3, _ = spab(sd, "block_3", b2, policy, train)
    b4, _ = spab(sd, "block_4", b3, policy, train)
    b5, _ = spab(sd, "block_5", b4, policy, train)
    b6, b5_2 = spab(sd, "block_6", b5, policy, train)
    b6 = conv3xc(sd, "conv_2", b6, policy, train)
    out = policy.conv2d(torch.cat([feat, b6, b1, b5_2], 1), sd["conv_cat.weight"], sd["conv_cat.bias"], 1, 0)
    return F.pixel_shuffle(policy.conv2d(out, sd["upsampler.0.weight"], sd["upsampler.0.bias"], 1, 1), hp["scale"])
