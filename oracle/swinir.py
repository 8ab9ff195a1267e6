"""Oracle restatement of SwinIR (test infrastructure, see oracle/__init__.py).

Follows:
  * presets / hyper-parameters      /root/reference/traiNNer/archs/swinir_arch.py:13-163 (swinir_s / swinir_m / swinir_l)
  * block arithmetic (in-tree twin) /root/reference/traiNNer/archs/drct_arch.py:52-75 (Mlp), :78-109 (window partition/reverse),
                                    :112-223 (WindowAttention), :415-575 (SwinTransformerBlock, shift mask :492-521),
                                    :624-662 (PatchEmbed), :704-725 (Upsample = [conv, PixelShuffle(2)] x log2(scale))
  * parameter checksums             /root/reference/docs/source/benchmarks.md:71 (swinir_s 4x 929,628), :75 (swinir_m 4x 11,900,199)
The network body lives in spandrel (`spandrel.architectures.SwinIR`, pip `spandrel>=0.4.0`), absent here: the published SwinIR
topology is restated (mean/img_range handling, RSTB = blocks -> conv -> + residual, conv_before_upsample + LeakyReLU(0.01),
UpsampleOneStep for `pixelshuffledirect`); numeric values are *parity unpinned*, structure is pinned by the checksums.
Scope: upsampler `pixelshuffle` / `pixelshuffledirect`, resi_connection `1conv`, drop_path evaluated with an explicit per-sample
keep mask (None = no stochastic depth), H and W multiples of the window size.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
from torch.nn import functional as F

from .precision import FP32, Policy

RGB_MEAN = (0.4488, 0.4371, 0.4040)

PRESETS = {
    "swinir_s": dict(embed_dim=60, depths=[6] * 4, num_heads=[6] * 4, window_size=8, mlp_ratio=2.0, upsampler="pixelshuffledirect",
                     resi_connection="1conv", img_range=1.0, drop_path_rate=0.1),
    "swinir_m": dict(embed_dim=180, depths=[6] * 6, num_heads=[6] * 6, window_size=8, mlp_ratio=2.0, upsampler="pixelshuffle",
                     resi_connection="1conv", img_range=1.0, drop_path_rate=0.1),
}


def hparams(name="swinir_m", scale=4, in_chans=3, **over):
    hp = dict(PRESETS[name], scale=scale, in_chans=in_chans, num_feat=64)
    hp.update(over)
    return hp


def rel_index(ws: int) -> torch.Tensor:
    """drct_arch.py:143-160"""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij")).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shift_mask(h: int, w: int, ws: int, shift: int) -> torch.Tensor:
    """drct_arch.py:492-521 -> (nW, ws*ws, ws*ws) of 0 / -100"""
    img = torch.zeros(1, h, w, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = img.view(1, h // ws, ws, w // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def param_table(hp) -> "OrderedDict[str, tuple]":
    C, nf = hp["embed_dim"], hp["num_feat"]
    hid = int(C * hp["mlp_ratio"])
    ws = hp["window_size"]
    t = OrderedDict()
    t["conv_first.weight"] = (C, hp["in_chans"], 3, 3); t["conv_first.bias"] = (C,)
    t["patch_embed.norm.weight"] = (C,); t["patch_embed.norm.bias"] = (C,)
    for i, (d, nh) in enumerate(zip(hp["depths"], hp["num_heads"])):
        for j in range(d):
            p = f"layers.{i}.residual_group.blocks.{j}."
            t[p + "norm1.weight"] = (C,); t[p + "norm1.bias"] = (C,)
            t[p + "attn.relative_position_bias_table"] = ((2 * ws - 1) ** 2, nh)
            t[p + "attn.qkv.weight"] = (3 * C, C); t[p + "attn.qkv.bias"] = (3 * C,)
            t[p + "attn.proj.weight"] = (C, C); t[p + "attn.proj.bias"] = (C,)
            t[p + "norm2.weight"] = (C,); t[p + "norm2.bias"] = (C,)
            t[p + "mlp.fc1.weight"] = (hid, C); t[p + "mlp.fc1.bias"] = (hid,)
            t[p + "mlp.fc2.weight"] = (C, hid); t[p + "mlp.fc2.bias"] = (C,)
        t[f"layers.{i}.conv.weight"] = (C, C, 3, 3); t[f"layers.{i}.conv.bias"] = (C,)
    t["norm.weight"] = (C,); t["norm.bias"] = (C,)
    t["conv_after_body.weight"] = (C, C, 3, 3); t["conv_after_body.bias"] = (C,)
    s = hp["scale"]
    if hp["upsampler"] == "pixelshuffle":
        t["conv_before_upsample.0.weight"] = (nf, C, 3, 3); t["conv_before_upsample.0.bias"] = (nf,)
        for k in range(int(math.log2(s))):
            t[f"upsample.{2 * k}.weight"] = (4 * nf, nf, 3, 3); t[f"upsample.{2 * k}.bias"] = (4 * nf,)
        t["conv_last.weight"] = (hp["in_chans"], nf, 3, 3); t["conv_last.bias"] = (hp["in_chans"],)
    elif hp["upsampler"] == "pixelshuffledirect":
        t["upsample.0.weight"] = (s * s * hp["in_chans"], C, 3, 3); t["upsample.0.bias"] = (s * s * hp["in_chans"],)
    else:
        raise NotImplementedError(hp["upsampler"])
    return t


def param_count(hp) -> int:
    return sum(math.prod(s) for s in param_table(hp).values())


def init_state_dict(hp, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in param_table(hp).items():
        if k.endswith("norm.weight") or k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif "relative_position_bias_table" in k:
            sd[k] = 0.2 * torch.randn(shp, generator=g)          # wider than the trunc-normal(0.02) init so the bias path is exercised
        elif k.endswith("bias"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = math.prod(shp[1:])
            sd[k] = torch.randn(shp, generator=g) / math.sqrt(fan_in)
    return sd


def _linear(policy, x, w, b):
    # a Linear is a 1x1 conv for the precision policy (bf16 operands, fp32 accumulate)
    y = policy.conv2d(x.reshape(-1, x.shape[-1], 1, 1), w[:, :, None, None], b, 1, 0)
    return y.reshape(*x.shape[:-1], w.shape[0])


def block(sd, p, x, h, w, C, nh, ws, shift, policy, keep=None):
    """One SwinTransformerBlock on tokens x (B, H*W, C) (drct_arch.py:532-575)."""
    B = x.shape[0]
    short = x
    t = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5).view(B, h, w, C)
    if shift:
        t = torch.roll(t, (-shift, -shift), (1, 2))
    win = t.view(B, h // ws, ws, w // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    qkv = _linear(policy, win, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]).reshape(-1, ws * ws, 3, nh, C // nh).permute(2, 0, 3, 1, 4)
    scale = (C // nh) ** -0.5
    if policy.name == "fp32":
        q, k, v = qkv[0] * scale, qkv[1], qkv[2]          # the reference order: q * scale, then q @ k^T (drct_arch.py:188-190)
        attn = q @ k.transpose(-2, -1)
    else:
        q, k, v = qkv[0], qkv[1], qkv[2]                  # bf16 operands as stored; the (non power-of-two) scale is applied in fp32
        attn = policy.matmul(q, k.transpose(-2, -1)) * scale
    bias = sd[p + "attn.relative_position_bias_table"][rel_index(ws).view(-1)].view(ws * ws, ws * ws, nh).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if shift:
        m = shift_mask(h, w, ws, shift)
        nW = m.shape[0]
        attn = (attn.view(B, nW, nh, ws * ws, ws * ws) + m.unsqueeze(1).unsqueeze(0)).view(-1, nh, ws * ws, ws * ws)
    attn = attn.softmax(-1)
    o = policy.matmul(attn, v).transpose(1, 2).reshape(-1, ws * ws, C)
    o = _linear(policy, o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    o = o.view(B, h // ws, w // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, h, w, C)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    o = o.reshape(B, h * w, C)
    if keep is not None:
        o = o * keep.view(B, 1, 1)
    x = short + o
    t = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    t = F.gelu(_linear(policy, t, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    t = _linear(policy, t, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    if keep is not None:
        t = t * keep.view(B, 1, 1)
    return x + t


def forward(sd, x, hp, policy: Policy = FP32, keep_masks=None):
    """x: B x 3 x h x w in [0,1].  keep_masks: optional {(layer, block): tensor(B) of 0 or 1/keep_prob} (DropPath)."""
    C, ws = hp["embed_dim"], hp["window_size"]
    B, _, h, w = x.shape
    assert h % ws == 0 and w % ws == 0
    mean = x.new_tensor(RGB_MEAN).view(1, 3, 1, 1) if hp["in_chans"] == 3 else x.new_zeros(1, 1, 1, 1)
    rng = hp["img_range"]
    x = (x - mean) * rng
    conv = lambda name, t: policy.conv2d(t, sd[name + ".weight"], sd[name + ".bias"], 1, 1)  # noqa: E731
    fea = conv("conv_first", x)
    t = fea.flatten(2).transpose(1, 2)
    t = F.layer_norm(t, (C,), sd["patch_embed.norm.weight"], sd["patch_embed.norm.bias"], 1e-5)
    for i, (d, nh) in enumerate(zip(hp["depths"], hp["num_heads"])):
        res = t
        for j in range(d):
            keep = None if keep_masks is None else keep_masks.get((i, j))
            t = block(sd, f"layers.{i}.residual_group.blocks.{j}.", t, h, w, C, nh, ws, 0 if j % 2 == 0 else ws // 2, policy, keep)
        img = t.transpose(1, 2).reshape(B, C, h, w)
        t = conv(f"layers.{i}.conv", img).flatten(2).transpose(1, 2) + res
    t = F.layer_norm(t, (C,), sd["norm.weight"], sd["norm.bias"], 1e-5)
    body = conv("conv_after_body", t.transpose(1, 2).reshape(B, C, h, w)) + fea
    if hp["upsampler"] == "pixelshuffle":
        u = F.leaky_relu(conv("conv_before_upsample.0", body), 0.01)
        for k in range(int(math.log2(hp["scale"]))):
            u = F.pixel_shuffle(conv(f"upsample.{2 * k}", u), 2)
        out = conv("conv_last", u)
    else:
        out = F.pixel_shuffle(conv("upsample.0", body), hp["scale"])
    return out / rng + mean
