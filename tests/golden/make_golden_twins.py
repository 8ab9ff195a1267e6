#!/usr/bin/env python3
"""Generates tests/golden/{esrgan_twin,swin_twin,paragon_otf,psnr}.pt from reference files that load by path in this container
(run HERE only: `python tests/golden/make_golden_twins.py`; the GPU box never reads /root/reference).

Nothing from the reference is copied: each file is imported by path under stub parent packages (same technique as make_golden.py),
executed on seeded inputs, and only input / output TENSORS plus a manifest of seeds and hyper-parameters are written.  Weights are NOT
stored: they come from the oracle's seeded initialisers (oracle/esrgan.py::init_state_dict, oracle/swinir.py::init_state_dict), which the
tests call again with the same seed.

Stubs and why they are arithmetic-neutral:
  timm.layers / spandrel.util.timm   `DropPath` = identity (only ever constructed with rate 0 here), `to_2tuple`, `trunc_normal_` = torch's
                                     (initial values are overwritten by load_state_dict anyway)
  spandrel.util.store_hyperparameters  a no-op decorator (it only records constructor arguments)
  spandrel...padding.pad_to_multiple   a placeholder that raises (never called by the classes used here)
  traiNNer.utils.RNG                 the two-line stand-in of make_golden_otf.py (numpy Generator semantics of utils/rng.py:7-25)
  cv2                                an empty module: psnr_ssim.py imports it at the top, calculate_psnr / calculate_psnr_pt never touch it

Fixtures
  esrgan_twin.pt   the reference's ResidualDenseBlock_ / RRDB_ / RRDBUpsample (archs/scunet_aaf6aa_arch.py:310-384) composed in ESRGAN's order
                   (conv_first -> RRDB_ x nb -> trunk conv + skip -> RRDBUpsample(nb=0) = [nearest x2, conv, LeakyReLU] x log2(s) + conv +
                   LeakyReLU -> conv_last; the three plain convs and the skip add are nn.Conv2d / `+` glue in this script): output and every
                   parameter gradient for a fixed cotangent (nf 32, nb 1, 4x), output + selected gradients (nf 64, nb 1, 2x), and the
                   block-level tensors (one ResidualDenseBlock_, one RRDB_) the oracle's taps are compared with.
  swin_twin.pt     archs/drct_arch.py: Mlp, window_partition / window_reverse, WindowAttention (with and without shift mask),
                   SwinTransformerBlock (shift 0 and 4), and a small SwinIR-shaped net composed of the reference's SwinTransformerBlock /
                   PatchEmbed / PatchUnEmbed / Upsample classes (conv / LayerNorm / skip glue in this script): outputs + parameter gradients.
  paragon_otf.pt   every static, codec-free stage of models/paragon_otf_degradations.py:251-575 (motion blur, lens distortion, exposure,
                   colour temperature, sensor noise, rolling shutter, oversharpening, chromatic aberration, aliasing) under seeded RNGs with
                   its gate forced open: input, drawn parameters (replayed from the same seed in the reference's draw order), output; and one
                   run with the gate closed (pass-through, one draw consumed).
  psnr.pt          metrics/psnr_ssim.py: calculate_psnr (numpy, 8-bit images, RGB and Y) and calculate_psnr_pt (RGB and Y, crop 0 / 4).
  iterative_schedule.pt   losses/iterative_loss_wrapper.py:87-246: weight / is_active of eight schedules on iterations 0..44, the wrapper's forward on a scalar
                   and a dict-valued toy module at four iterations, the two constructor rejections.
  vgg_style_discriminator.pt   archs/discriminator_arch.py:9-95 VGGStyleDiscriminator (num_feat 8, input 128, batch 2): state dict, training-mode logits, input and
                   parameter gradients, BatchNorm running statistics after the forward, eval-mode logits.
"""
import importlib.util
import math
import os
import random
import sys
import types

import numpy as np
import torch
from torch import nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT))
sys.path.insert(0, ROOT)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, *a, **k):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob and self.training:
            raise RuntimeError("stub DropPath: only rate 0 is arithmetic-neutral")
        return x


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class RNG:
    _rng = None

    @classmethod
    def get_rng(cls):
        return cls._rng


def _no_op_decorator(*a, **k):
    return lambda cls: cls


def _raise(*a, **k):
    raise RuntimeError("stub: not available offline")


def setup():
    for n in ("traiNNer", "traiNNer.utils", "traiNNer.archs", "traiNNer.models", "traiNNer.metrics", "cv2", "timm", "spandrel", "spandrel.architectures",
              "spandrel.architectures.__arch_helpers"):
        _stub(n)
    _stub("timm.layers", DropPath=DropPath, trunc_normal_=nn.init.trunc_normal_)
    _stub("spandrel.util", store_hyperparameters=_no_op_decorator)
    _stub("spandrel.util.timm", DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=nn.init.trunc_normal_)
    _stub("spandrel.architectures.__arch_helpers.padding", pad_to_multiple=_raise)
    _load("traiNNer.utils.registry", "traiNNer/utils/registry.py")
    _stub("traiNNer.utils.types", PixelFormat=str)
    cu = _load("traiNNer.utils.color_util", "traiNNer/utils/color_util.py")
    sys.modules["traiNNer.utils"].RNG = RNG
    sys.modules["traiNNer.utils"].rgb2ycbcr = cu.rgb2ycbcr


# ---------------------------------------------------------------------------------------------------------------------------------
def esrgan_twin():
    from oracle import esrgan as OE

    S = _load("traiNNer.archs.scunet_aaf6aa_arch", "traiNNer/archs/scunet_aaf6aa_arch.py")

    class Composed(nn.Module):
        """ESRGAN's layer order around the reference's own RRDB_ / RRDBUpsample classes."""

        def __init__(self, nf, nb, scale):
            super().__init__()
            self.conv_first = nn.Conv2d(3, nf, 3, 1, 1)
            self.blocks = nn.ModuleList([S.RRDB_(nf, 32) for _ in range(nb)])
            self.trunk = nn.Conv2d(nf, nf, 3, 1, 1)
            self.tail = S.RRDBUpsample(nf, nb=0, scale=scale)
            self.conv_last = nn.Conv2d(nf, 3, 3, 1, 1)

        def forward(self, x, taps=None):
            fea = self.conv_first(x)
            t = fea
            for i, b in enumerate(self.blocks):
                if taps is not None and i == 0:
                    taps["rdb0"] = b.rdb1(t)
                t = b(t)
                if taps is not None and i == 0:
                    taps["rrdb0"] = t
            t = fea + self.trunk(t)
            if taps is not None:
                taps["trunk"] = t
            return self.conv_last(self.tail(t))

    def load(net, sd, hp):
        """spandrel old-arch keys (oracle/esrgan.py::layer_table) -> the composed module."""
        nb = hp["nb"]

        def put(conv, key):
            conv.weight.data.copy_(sd[key + ".weight"])
            conv.bias.data.copy_(sd[key + ".bias"])

        put(net.conv_first, "model.0")
        for i, b in enumerate(net.blocks):
            for r in (1, 2, 3):
                for c in range(1, 6):
                    put(getattr(getattr(b, f"rdb{r}"), f"conv{c}"), f"model.1.sub.{i}.RDB{r}.conv{c}.0")
        put(net.trunk, f"model.1.sub.{nb}")
        convs = [m for m in net.tail.up if isinstance(m, nn.Conv2d)]
        idx = 2
        for k in range(int(math.log2(hp["net_scale"]))):
            put(convs[k], f"model.{idx + 1}")
            idx += 3
        put(convs[-1], f"model.{idx}")
        put(net.conv_last, f"model.{idx + 2}")

    def grads_by_key(net, hp):
        nb = hp["nb"]
        out = {}

        def get(conv, key):
            out[key + ".weight"] = conv.weight.grad.clone()
            out[key + ".bias"] = conv.bias.grad.clone()

        get(net.conv_first, "model.0")
        for i, b in enumerate(net.blocks):
            for r in (1, 2, 3):
                for c in range(1, 6):
                    get(getattr(getattr(b, f"rdb{r}"), f"conv{c}"), f"model.1.sub.{i}.RDB{r}.conv{c}.0")
        get(net.trunk, f"model.1.sub.{nb}")
        convs = [m for m in net.tail.up if isinstance(m, nn.Conv2d)]
        idx = 2
        for k in range(int(math.log2(hp["net_scale"]))):
            get(convs[k], f"model.{idx + 1}")
            idx += 3
        get(convs[-1], f"model.{idx}")
        get(net.conv_last, f"model.{idx + 2}")
        return out

    fix = {}
    for tag, kw, seed, shape, keep in (("nf32_x4", dict(scale=4, num_filters=32, num_blocks=1), 11, (2, 3, 16, 24), None),
                                       ("nf64_x2", dict(scale=2, use_pixel_unshuffle=False, num_filters=64, num_blocks=1), 12, (1, 3, 12, 20),
                                        ["model.0", "model.1.sub.0.RDB1.conv1.0", "model.1.sub.0.RDB2.conv3.0", "model.1.sub.0.RDB3.conv5.0", "model.1.sub.1", "model.3",
                                         "model.5", "model.7"])):
        hp = OE.esrgan_hparams(**kw)
        sd = OE.init_state_dict(hp, seed)
        net = Composed(hp["nf"], hp["nb"], hp["net_scale"])
        load(net, sd, hp)
        g = torch.Generator().manual_seed(seed + 100)
        x = torch.rand(*shape, generator=g)
        taps = {}
        out = net(x, taps)
        cot = torch.randn(out.shape, generator=g)
        out.backward(cot)
        gr = grads_by_key(net, hp)
        if keep is not None:
            gr = {k: v for k, v in gr.items() if any(k.startswith(p + ".") for p in keep)}
        fix[tag] = dict(kwargs=kw, seed=seed, x=x, out=out.detach(), cotangent=cot, grads=gr, taps={k: v.detach() for k, v in taps.items()},
                        n_params=sum(p.numel() for p in net.parameters()))
    torch.save(fix, os.path.join(OUT, "esrgan_twin.pt"))
    print("esrgan_twin.pt:", {k: (tuple(v["out"].shape), v["n_params"], len(v["grads"])) for k, v in fix.items()})


# ---------------------------------------------------------------------------------------------------------------------------------
def swin_twin():
    from oracle import swinir as OS

    D = _load("traiNNer.archs.drct_arch", "traiNNer/archs/drct_arch.py")
    fix = {}
    g = torch.Generator().manual_seed(31)
    # window_partition / window_reverse
    t = torch.randn(2, 16, 24, 5, generator=g)
    win = D.window_partition(t, 8)
    fix["window"] = dict(x=t, windows=win, back=D.window_reverse(win, 8, 16, 24))
    # Mlp
    mlp = D.Mlp(in_features=60, hidden_features=120)
    xm = torch.randn(3, 10, 60, generator=g)
    fix["mlp"] = dict(sd={k: v.clone() for k, v in mlp.state_dict().items()}, x=xm, out=mlp(xm).detach())
    # WindowAttention with / without mask; SwinTransformerBlock shift 0 / 4
    C, nh, ws, h, w = 60, 6, 8, 16, 24
    B = 1
    for shift in (0, 4):
        torch.manual_seed(40 + shift)
        blk = D.SwinTransformerBlock(dim=C, input_resolution=(h, w), num_heads=nh, window_size=ws, shift_size=shift, mlp_ratio=2.0)
        with torch.no_grad():
            blk.attn.relative_position_bias_table.mul_(10.0)      # trunc-normal(0.02) is nearly invisible: widen so the bias path is exercised
            blk.norm1.weight.add_(0.1 * torch.randn(C, generator=g))
            blk.norm1.bias.add_(0.05 * torch.randn(C, generator=g))
        sd = {k: v.clone() for k, v in blk.state_dict().items() if "relative_position_index" not in k and "attn_mask" not in k}
        xt = torch.randn(B, h * w, C, generator=g).requires_grad_(True)
        out = blk(xt, (h, w))
        cot = torch.randn(out.shape, generator=g)
        grads = torch.autograd.grad(out, [xt, *blk.parameters()], cot)
        names = [k for k, _ in blk.named_parameters()]
        xw = torch.randn(B * (h // ws) * (w // ws), ws * ws, C, generator=g)
        attn_out = blk.attn(xw, mask=blk.attn_mask)
        fix[f"block_shift{shift}"] = dict(sd=sd, x=xt.detach(), out=out.detach(), cotangent=cot, gx=grads[0], grads=dict(zip(names, grads[1:])),
                                          attn_x=xw, attn_out=attn_out.detach(), attn_mask=None if blk.attn_mask is None else blk.attn_mask.clone(),
                                          dims=dict(C=C, nh=nh, ws=ws, h=h, w=w, shift=shift))

    # a small SwinIR-shaped net out of the reference's classes
    class Group(nn.Module):
        def __init__(self, blocks):
            super().__init__()
            self.blocks = nn.ModuleList(blocks)

    class RSTB(nn.Module):
        def __init__(self, C, nh, ws, depth, mlp_ratio, res):
            super().__init__()
            self.residual_group = Group([D.SwinTransformerBlock(dim=C, input_resolution=res, num_heads=nh, window_size=ws, shift_size=0 if j % 2 == 0 else ws // 2,
                                                                mlp_ratio=mlp_ratio) for j in range(depth)])
            self.conv = nn.Conv2d(C, C, 3, 1, 1)

    class Composed(nn.Module):
        def __init__(self, hp, res):
            super().__init__()
            C, nf = hp["embed_dim"], hp["num_feat"]
            self.hp, self.res = hp, res
            self.conv_first = nn.Conv2d(3, C, 3, 1, 1)
            self.patch_embed = D.PatchEmbed(img_size=res, patch_size=1, in_chans=C, embed_dim=C, norm_layer=nn.LayerNorm)
            self.patch_unembed = D.PatchUnEmbed(img_size=res, patch_size=1, in_chans=C, embed_dim=C)
            self.layers = nn.ModuleList([RSTB(C, nh, hp["window_size"], d, hp["mlp_ratio"], res) for d, nh in zip(hp["depths"], hp["num_heads"])])
            self.norm = nn.LayerNorm(C)
            self.conv_after_body = nn.Conv2d(C, C, 3, 1, 1)
            self.conv_before_upsample = nn.Sequential(nn.Conv2d(C, nf, 3, 1, 1), nn.LeakyReLU(inplace=True))
            self.upsample = D.Upsample(hp["scale"], nf)
            self.conv_last = nn.Conv2d(nf, 3, 3, 1, 1)

        def forward(self, x):
            hp, res = self.hp, self.res
            mean = x.new_tensor(OS.RGB_MEAN).view(1, 3, 1, 1)
            x = (x - mean) * hp["img_range"]
            fea = self.conv_first(x)
            t = self.patch_embed(fea)
            for layer in self.layers:
                r = t
                for b in layer.residual_group.blocks:
                    t = b(t, res)
                t = self.patch_embed.__class__.forward(_NoNorm, layer.conv(self.patch_unembed(t, res))) + r
            t = self.norm(t)
            body = self.conv_after_body(self.patch_unembed(t, res)) + fea
            out = self.conv_last(self.upsample(self.conv_before_upsample(body)))
            return out / hp["img_range"] + mean

    class _NoNormT:
        norm = None
    _NoNorm = _NoNormT()

    hp = OS.hparams("swinir_s", scale=2, depths=[2, 2], num_heads=[6, 6], upsampler="pixelshuffle")
    sd = OS.init_state_dict(hp, seed=21)
    res = (16, 24)
    net = Composed(hp, res)
    r = net.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys and all("relative_position_index" in k or "attn_mask" in k for k in r.missing_keys), r
    x = torch.rand(2, 3, *res, generator=g)
    out = net(x)
    cot = torch.randn(out.shape, generator=g)
    out.backward(cot)
    grads = {k: p.grad.clone() for k, p in net.named_parameters()}
    assert set(grads) == set(sd), (set(grads) ^ set(sd))
    fix["net"] = dict(name="swinir_s", over=dict(scale=2, depths=[2, 2], num_heads=[6, 6], upsampler="pixelshuffle"), seed=21, x=x, out=out.detach(),
                      cotangent=cot, grads=grads)
    torch.save(fix, os.path.join(OUT, "swin_twin.pt"))
    print("swin_twin.pt:", list(fix), tuple(out.shape), sum(v.numel() for v in grads.values()), "params")


# ---------------------------------------------------------------------------------------------------------------------------------
def paragon_otf():
    P = _load("traiNNer.models.paragon_otf_degradations", "traiNNer/models/paragon_otf_degradations.py").ParagonOTF
    g = torch.Generator().manual_seed(77)
    img = torch.rand(2, 3, 40, 56, generator=g)
    NS = types.SimpleNamespace
    fix = {"img": img, "stages": {}}

    def run(tag, fn, opt, seed, replay):
        RNG._rng = np.random.default_rng(seed)
        random.seed(seed)
        torch.manual_seed(seed)
        out = fn(img.clone(), opt)
        # the drawn parameters: the same seeds walked again in the reference's draw order (the gate draw first)
        rng = np.random.default_rng(seed)
        rnd = random.Random(seed)
        torch.manual_seed(seed)
        gate = rng.uniform()
        params = replay(rng, rnd)
        fix["stages"][tag] = dict(seed=seed, opt=dict(vars(opt)), gate_draw=gate, params=params, out=out)

    for i, (ks, ang) in enumerate((((5, 15), (0, 360)), ((7, 7), (30, 60)))):
        run(f"motion_blur/{i}", P.apply_motion_blur, NS(motion_blur_prob=1.0, motion_blur_kernel_size=ks, motion_blur_angle_range=ang), 200 + i,
            lambda rng, rnd, ks=ks, ang=ang: dict(kernel_size=rnd.randint(ks[0], ks[1]), angle=rng.uniform(ang[0], ang[1])))
    for i, rg in enumerate(((-0.3, 0.3), (0.1, 0.3))):
        run(f"lens_distortion/{i}", P.apply_lens_distortion, NS(lens_distort_prob=1.0, lens_distort_strength_range=rg), 210 + i,
            lambda rng, rnd, rg=rg: dict(strength=rng.uniform(rg[0], rg[1])))
    for i, rg in enumerate(((0.5, 2.0), (1.2, 1.6))):
        run(f"exposure/{i}", P.apply_exposure_errors, NS(exposure_prob=1.0, exposure_factor_range=rg), 220 + i, lambda rng, rnd, rg=rg: dict(factor=rng.uniform(rg[0], rg[1])))
    for i, rg in enumerate(((0.05, 0.2), (-0.2, -0.05))):
        run(f"color_temperature/{i}", P.apply_color_temperature_shift, NS(color_temp_prob=1.0, color_temp_shift_range=rg), 230 + i,
            lambda rng, rnd, rg=rg: dict(shift=rng.uniform(rg[0], rg[1])))
    run("sensor_noise/0", P.apply_sensor_noise, NS(sensor_noise_prob=1.0, sensor_noise_std_range=(0.01, 0.1)), 240,
        lambda rng, rnd: dict(std=rng.uniform(0.01, 0.1), noise=torch.randn_like(img)))
    for i, rg in enumerate(((-0.1, 0.1), (0.05, 0.1))):
        run(f"rolling_shutter/{i}", P.apply_rolling_shutter, NS(rolling_shutter_prob=1.0, rolling_shutter_strength_range=rg), 250 + i,
            lambda rng, rnd, rg=rg: dict(strength=rng.uniform(rg[0], rg[1])))
    run("oversharpen/0", P.apply_oversharpening, NS(oversharpen_prob=1.0, oversharpen_strength=(1.0, 2.0)), 260, lambda rng, rnd: dict(strength=rng.uniform(1.0, 2.0)))
    run("chromatic_aberration/0", P.apply_chromatic_aberration, NS(chromatic_aberration_prob=1.0), 270, lambda rng, rnd: dict())
    for i, rg in enumerate(((0.6, 0.9), (0.3, 0.5))):
        run(f"aliasing/{i}", P.apply_aliasing_artifacts, NS(aliasing_prob=1.0, aliasing_scale_range=rg), 280 + i, lambda rng, rnd, rg=rg: dict(scale=rng.uniform(rg[0], rg[1])))
    # closed gate: pass-through, exactly one draw consumed
    RNG._rng = np.random.default_rng(5)
    out = P.apply_exposure_errors(img.clone(), NS(exposure_prob=0.0, exposure_factor_range=(0.5, 2.0)))
    fix["closed_gate"] = dict(seed=5, equal=bool(torch.equal(out, img)), next_draw=float(RNG._rng.uniform()))
    k = P._create_motion_blur_kernel(9, 37.0)
    fix["motion_kernel_9_37"] = k
    torch.save(fix, os.path.join(OUT, "paragon_otf.pt"))
    print("paragon_otf.pt:", len(fix["stages"]), "stage runs")


# ---------------------------------------------------------------------------------------------------------------------------------
def psnr():
    _load("traiNNer.metrics.metric_util", "traiNNer/metrics/metric_util.py")
    M = _load("traiNNer.metrics.psnr_ssim", "traiNNer/metrics/psnr_ssim.py")
    g = torch.Generator().manual_seed(91)
    a = torch.rand(2, 3, 40, 52, generator=g)
    b = (a + 0.05 * torch.randn(a.shape, generator=g)).clamp(0, 1)
    fix = dict(a=a, b=b, pt={}, np={})
    for crop in (0, 4):
        for y in (False, True):
            fix["pt"][f"{crop}/{int(y)}"] = M.calculate_psnr_pt(a, b, crop_border=crop, test_y_channel=y)
            a8 = (a[0].clamp(0, 1) * 255.0).round().permute(1, 2, 0).numpy().astype(np.uint8)
            b8 = (b[0].clamp(0, 1) * 255.0).round().permute(1, 2, 0).numpy().astype(np.uint8)
            fix["np"][f"{crop}/{int(y)}"] = float(M.calculate_psnr(a8, b8, crop_border=crop, input_order="HWC", test_y_channel=y))
    fix["identical_np"] = float(M.calculate_psnr(a8, a8, crop_border=0))
    torch.save(fix, os.path.join(OUT, "psnr.pt"))
    print("psnr.pt:", fix["pt"], fix["np"])


def vgg_style_discriminator():
    """archs/discriminator_arch.py:9-95 (torch only): training-mode forward of a seeded batch (batch statistics, running statistics updated once),
    input and parameter gradients for a fixed cotangent, then the eval-mode output on the updated running statistics."""
    D = _load("traiNNer.archs.discriminator_arch", "traiNNer/archs/discriminator_arch.py")
    fix = {}
    for tag, kw, nb in (("nf8_128", dict(num_in_ch=3, num_feat=8, input_size=128), 2),):
        torch.manual_seed(77)
        net = D.VGGStyleDiscriminator(**kw).train()
        g = torch.Generator().manual_seed(5)
        with torch.no_grad():            # non-trivial affine parameters and running statistics
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.add_(0.2 * torch.randn(m.weight.shape, generator=g))
                    m.bias.add_(0.1 * torch.randn(m.bias.shape, generator=g))
                    m.running_mean.add_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                    m.running_var.mul_(1.0 + 0.2 * torch.rand(m.running_var.shape, generator=g))
        state = {k: v.clone() for k, v in net.state_dict().items()}
        x = torch.rand(nb, 3, kw["input_size"], kw["input_size"], generator=g).requires_grad_(True)
        out = net(x)
        cot = torch.randn(out.shape, generator=g)
        grads = torch.autograd.grad(out, [x, *net.parameters()], cot)
        names = [k for k, _ in net.named_parameters()]
        after = {k: v.clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k}
        net.eval()
        with torch.no_grad():
            out_eval = net(x.detach())
        fix[tag] = dict(kwargs=kw, state=state, x=x.detach(), out=out.detach(), cotangent=cot, gx=grads[0], grads=dict(zip(names, grads[1:])), after=after,
                        out_eval=out_eval, n_params=sum(p.numel() for p in net.parameters()))
    torch.save(fix, os.path.join(OUT, "vgg_style_discriminator.pt"))
    print("vgg_style_discriminator.pt:", {k: (tuple(v["out"].shape), v["n_params"]) for k, v in fix.items()})


def iterative_schedule():
    """losses/iterative_loss_wrapper.py (torch only): the weight of every schedule shape on a grid of iterations, and the wrapper's forward on
    a scalar and on a dict-valued toy module (values only -- the toy modules are this script's)."""
    import warnings

    W = _load("traiNNer.losses.iterative_loss_wrapper", "traiNNer/losses/iterative_loss_wrapper.py")

    class Scalar(nn.Module):
        def forward(self, a, b):
            return (a - b).abs().mean()

    class Pair(nn.Module):
        def forward(self, a, b):
            return {"p": (a - b).abs().mean(), "q": ((a - b) ** 2).mean()}

    cfgs = [dict(loss_weight=1.0), dict(loss_weight=0.5, start_iter=10), dict(loss_weight=0.0, start_iter=5, target_iter=25, target_weight=2.0),
            dict(loss_weight=1.0, start_iter=0, target_iter=40, target_weight=0.1, schedule_type="cosine"),
            dict(loss_weight=0.2, start_iter=8, target_iter=16, target_weight=0.9, schedule_type="step", disable_after=30),
            dict(loss_weight=0.3, start_iter=0, target_iter=10, target_weight=0.6, disable_after=12, schedule_type="LINEAR"),
            dict(loss_weight=0.3, start_iter=2, target_iter=10, target_weight=0.6, schedule_type="nonsense"),
            dict(loss_weight=0.7, start_iter=3, target_iter=3, target_weight=1.5)]
    iters = list(range(0, 45))
    g = torch.Generator().manual_seed(3)
    a, b = torch.rand(2, 3, 8, 8, generator=g), torch.rand(2, 3, 8, 8, generator=g)
    rows = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for c in cfgs:
            w = W.create_iterative_loss(Scalar(), c)
            wp = W.create_iterative_loss(Pair(), c)
            rows.append(dict(cfg=c, weights=[float(w.get_current_weight(i)) for i in iters], active=[bool(w.is_active(i)) for i in iters],
                             scalar=[float(w(a, b, current_iter=i)) for i in (0, 9, 20, 44)],
                             pair=[{k: float(v) for k, v in wp(a, b, current_iter=i).items()} for i in (0, 9, 20, 44)], info=w.get_schedule_info()))
    bad = []
    for kw in (dict(loss_weight=1.0, start_iter=10, target_iter=5), dict(loss_weight=1.0, target_weight=-0.5)):
        try:
            W.create_iterative_loss(Scalar(), kw)
            bad.append(None)
        except ValueError as e:
            bad.append(str(e))
    torch.save(dict(iters=iters, fwd_iters=[0, 9, 20, 44], a=a, b=b, rows=rows, rejected=bad), os.path.join(OUT, "iterative_schedule.pt"))
    print("iterative_schedule.pt:", len(rows), "schedules;", bad)


if __name__ == "__main__":
    setup()
    which = sys.argv[1:] or ["esrgan", "swin", "otf", "psnr", "vggd", "sched"]
    for w in which:
        {"esrgan": esrgan_twin, "swin": swin_twin, "otf": paragon_otf, "psnr": psnr, "vggd": vgg_style_discriminator, "sched": iterative_schedule}[w]()
