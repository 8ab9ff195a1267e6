"""MoSR (Mamba-Out Super-Resolution, v1) -- CPU restatement.  Test infrastructure only (see oracle/__init__.py).

The reference registers `mosr` / `mosr_t` as thin factories over `spandrel.architectures.MoSR.MoSR`
(/root/reference/traiNNer/archs/mosr_arch.py:4,23-74); the spandrel wheel is absent from /root/reference and not installable offline, so the
module body is restated from
  * the factory arguments                                      archs/mosr_arch.py:23-74 (n_block 24 / 5, dim 64 / 48, kernel_size 7,
                                                               expansion_ratio 1.5, conv_ratio 1.0, upsampler "pixelshuffle" -> "ps")
  * the in-tree twins of its GatedCNNBlock (same author)       archs/mosrv2_arch.py:88-126, archs/moesr_arch.py:66-106
                                                               (norm -> fc1 3x3 -> split [hidden, hidden - cc, cc] -> depthwise conv on the
                                                               last part -> mish(g) * cat(i, c) -> fc2 3x3 -> mish -> + shortcut; v1 has a plain
                                                               k x k depthwise conv instead of the Inception one and no `gamma`)
  * the channels-first LayerNorm of those files                archs/moesr_arch.py:16-27 (eps 1e-6, biased variance)
  * the published MoSR source as recalled (umzi2/MoSR neosr/archs/mosr_arch.py, cited at archs/mosr_arch.py:1): body = conv3x3(in, dim),
    n_block blocks, conv3x3(dim, 2 dim), Mish, conv3x3(2 dim, dim), Mish, conv1x1(dim, dim); shortcut = AST-style ConvBlock
    (conv3x3 - Mish - conv3x3 - Mish, plus a 1x1 conv of the input); upsampler "ps" = conv3x3(dim, out_ch * scale^2) + PixelShuffle.

PINNED BY CHECKSUM ONLY: this structure reproduces all six parameter counts the reference publishes
(docs/source/benchmarks.md:43,132 and its 2x table: mosr 4x 4,287,600 / 3x 4,275,483 / 2x 4,266,828; mosr_t 4x 609,888 / 3x 600,795 /
2x 594,300) -- tests/test_oracle_cpu.py.  Forward VALUES are "parity unpinned" (no fixture of the real module exists in the reference);
the state-dict key names (`gblocks.N.*`, `shortcut.block.N.*`, `shortcut.conv11.*`, `upsampler.0.*`) follow the recalled source and are
unverified against a real checkpoint.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
from torch.nn import functional as F

from .precision import FP32, Policy


def hparams(scale=4, n_block=24, dim=64, kernel_size=7, expansion_ratio=1.5, conv_ratio=1.0, in_ch=3, out_ch=3):
    return dict(scale=scale, n_block=n_block, dim=dim, kernel_size=kernel_size, expansion_ratio=expansion_ratio, conv_ratio=conv_ratio, in_ch=in_ch,
                out_ch=out_ch)


PRESETS = {"mosr": dict(n_block=24, dim=64), "mosr_t": dict(n_block=5, dim=48)}      # archs/mosr_arch.py:27-28,54-55


def split_sizes(hp):
    hidden = int(hp["expansion_ratio"] * hp["dim"])
    cc = int(hp["conv_ratio"] * hp["dim"])
    return hidden, cc


def param_table(hp) -> "OrderedDict[str, tuple]":
    D, nb, k = hp["dim"], hp["n_block"], hp["kernel_size"]
    hidden, cc = split_sizes(hp)
    t = OrderedDict()

    def conv(p, ci, co, ks):
        t[p + ".weight"] = (co, ci, ks, ks); t[p + ".bias"] = (co,)

    conv("gblocks.0", hp["in_ch"], D, 3)
    for b in range(1, nb + 1):
        p = f"gblocks.{b}"
        t[p + ".norm.weight"] = (D,); t[p + ".norm.bias"] = (D,)
        conv(p + ".fc1", D, 2 * hidden, 3)
        t[p + ".conv.weight"] = (cc, 1, k, k); t[p + ".conv.bias"] = (cc,)
        conv(p + ".fc2", hidden, D, 3)
    conv(f"gblocks.{nb + 1}", D, 2 * D, 3)
    conv(f"gblocks.{nb + 3}", 2 * D, D, 3)
    conv(f"gblocks.{nb + 5}", D, D, 1)
    conv("shortcut.block.0", hp["in_ch"], D, 3)
    conv("shortcut.block.2", D, D, 3)
    conv("shortcut.conv11", hp["in_ch"], D, 1)
    conv("upsampler.0", D, hp["out_ch"] * hp["scale"] ** 2, 3)
    return t


def num_params(hp) -> int:
    n = 0
    for shp in param_table(hp).values():
        k = 1
        for s in shp:
            k *= s
        n += k
    return n


def init_params(hp, seed=0, std=0.02, bias_std=0.0) -> "OrderedDict[str, torch.Tensor]":
    """trunc_normal_(std 0.02) conv weights, zero biases, unit LayerNorm (the `_init_weights` of the twins, mosrv2_arch.py:110-115).
    Tests pass larger std / non-zero biases so that every term of the arithmetic is exercised."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in param_table(hp).items():
        if k.endswith("norm.weight"):
            sd[k] = torch.ones(shp) + (torch.randn(shp, generator=g) * bias_std if bias_std else 0)
        elif k.endswith("bias"):
            sd[k] = torch.randn(shp, generator=g) * bias_std if bias_std else torch.zeros(shp)
        else:
            sd[k] = torch.nn.init.trunc_normal_(torch.empty(shp), std=std, a=-2 * std, b=2 * std, generator=g)
    return sd


def layernorm_cf(x, w, b, eps=1e-6):
    """Channels-first LayerNorm, moesr_arch.py:23-27."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None] * x + b[:, None, None]


def gated_block(sd, p, x, hp, policy: Policy, keep=None):
    """GatedCNNBlock.forward (mosrv2_arch.py:117-126 without gamma; v1 applies DropPath to the branch before the residual add)."""
    hidden, cc = split_sizes(hp)
    k = hp["kernel_size"]
    t = layernorm_cf(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"])
    g, i, c = torch.split(policy.conv2d(t, sd[p + ".fc1.weight"], sd[p + ".fc1.bias"], 1, 1), [hidden, hidden - cc, cc], dim=1)
    c = F.conv2d(c, sd[p + ".conv.weight"], sd[p + ".conv.bias"], stride=1, padding=k // 2, groups=cc)      # depthwise: VALU fp32 in every policy
    t = F.mish(policy.conv2d(F.mish(g) * torch.cat((i, c), dim=1), sd[p + ".fc2.weight"], sd[p + ".fc2.bias"], 1, 1))
    if keep is not None:
        t = t * keep.view(-1, 1, 1, 1)
    return t + x


def forward(sd, x, hp, policy: Policy = FP32, keep_masks=None):
    """x: B x 3 x h x w in [0, 1] -> B x 3 x (h s) x (w s).  keep_masks: optional {block index (1-based): tensor(B) of 0 or 1/keep_prob}."""
    nb = hp["n_block"]
    t = policy.conv2d(x, sd["gblocks.0.weight"], sd["gblocks.0.bias"], 1, 1)
    for b in range(1, nb + 1):
        t = gated_block(sd, f"gblocks.{b}", t, hp, policy, None if keep_masks is None else keep_masks.get(b))
    t = F.mish(policy.conv2d(t, sd[f"gblocks.{nb + 1}.weight"], sd[f"gblocks.{nb + 1}.bias"], 1, 1))
    t = F.mish(policy.conv2d(t, sd[f"gblocks.{nb + 3}.weight"], sd[f"gblocks.{nb + 3}.bias"], 1, 1))
    t = policy.conv2d(t, sd[f"gblocks.{nb + 5}.weight"], sd[f"gblocks.{nb + 5}.bias"], 1, 0)
    s = F.mish(policy.conv2d(x, sd["shortcut.block.0.weight"], sd["shortcut.block.0.bias"], 1, 1))
    s = F.mish(policy.conv2d(s, sd["shortcut.block.2.weight"], sd["shortcut.block.2.bias"], 1, 1))
    s = s + policy.conv2d(x, sd["shortcut.conv11.weight"], sd["shortcut.conv11.bias"], 1, 0)
    u = policy.conv2d(t + s, sd["upsampler.0.weight"], sd["upsampler.0.bias"], 1, 1)
    return F.pixel_shuffle(u, hp["scale"])
