#!/usr/bin/env python3
"""Generates tests/golden/*.pt from the reference files that load by path in this container.

Run HERE only (needs /root/reference): `python tests/golden/make_golden.py`.  Nothing from the reference is
copied: each torch-only file is imported by path under stub parent packages, executed on
seeded inputs, and only input / output TENSORS plus a manifest are written.  The GPU box never reads /root/reference.

Fixtures
  pixel_gan_losses.pt   basic_loss.{L1Loss,MSELoss,CharbonnierLoss} (mean / sum) and gan_loss.GANLoss
                        (vanilla, lsgan, hinge, wgan, wgan_softplus x real/fake x is_disc)
  registry.pt           registry key behaviour (lower-casing, suffix fallback, duplicate assertion)
  sampler.pt            data_sampler.EnlargedSampler index streams (epoch-seeded permutation, rank stride)
  unet_discriminator_sn.pt  discriminator_arch.UNetDiscriminatorSN(num_feat=16): weights/u/v before and after one training-mode
                        forward, output, input- and parameter-gradients, eval-mode output
  filter2d.pt           img_process_util.filter2d (cv2 is only imported by that module, not used by the function): a shared 5x5 kernel and
                        per-sample 21x21 kernels on small images
  span.pt               span_arch.SPAN (feature_channels 8 / x2 / no norm, and 12 / x4 / norm): state dict, training-mode output (the
                        1x1 -> 3x3 -> 1x1 + skip form of Conv3XC), parameter gradients for a fixed cotangent, eval-mode output (merged
                        3x3) and the merged eval_conv weights.  `spandrel.util.store_hyperparameters` (absent wheel) is a no-op decorator here.
"""
import importlib.util
import os
import sys
import types

import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def main():
    for n in ("traiNNer", "traiNNer.utils", "traiNNer.losses", "traiNNer.data"):
        _stub(n)
    # torchvision is absent: basic_loss only needs the names at import time
    tv = _stub("torchvision")
    tvt = _stub("torchvision.transforms")
    tvt.GaussianBlur = object
    tvt.InterpolationMode = types.SimpleNamespace(BICUBIC="bicubic")
    tvt.v2 = types.SimpleNamespace()
    tv.transforms = tvt
    reg = _load("traiNNer.utils.registry", "traiNNer/utils/registry.py")
    cu = _stub("traiNNer.utils.color_util")
    cu.rgb2ycbcr_pt = cu.rgb_to_luma = lambda *a, **k: None
    hs = _stub("traiNNer.utils.hsluv")
    hs.rgb_to_hsluv = lambda *a, **k: None
    _load("traiNNer.losses.loss_util", "traiNNer/losses/loss_util.py")
    bl = _load("traiNNer.losses.basic_loss", "traiNNer/losses/basic_loss.py")
    gl = _load("traiNNer.losses.gan_loss", "traiNNer/losses/gan_loss.py")

    g = torch.Generator().manual_seed(20261004)
    pred = torch.rand(2, 3, 24, 20, generator=g)
    target = torch.rand(2, 3, 24, 20, generator=g)
    fix = {"pred": pred, "target": target, "pixel": {}, "gan": {}}
    for name, cls in (("l1", bl.L1Loss), ("mse", bl.MSELoss), ("charbonnier", bl.CharbonnierLoss)):
        for red in ("mean", "sum"):
            p = pred.clone().requires_grad_(True)
            v = cls(loss_weight=1.0, reduction=red)(p, target)
            v.backward()
            fix["pixel"][f"{name}/{red}"] = {"value": v.detach(), "grad": p.grad.clone()}
    logits = torch.randn(2, 1, 8, 8, generator=g)
    fix["logits"] = logits
    for gt_ in ("vanilla", "lsgan", "hinge", "wgan", "wgan_softplus"):
        m = gl.GANLoss(loss_weight=0.1, gan_type=gt_)
        for real in (True, False):
            for disc in (True, False):
                fix["gan"][f"{gt_}/{int(real)}/{int(disc)}"] = m(logits, real, is_disc=disc).detach()
    torch.save(fix, os.path.join(OUT, "pixel_gan_losses.pt"))

    r = reg.Registry("t")

    @r.register()
    class FooBar:  # noqa: N801
        pass

    @r.register(suffix="traiNNer")
    def baz():
        pass

    dup = False
    try:
        r.register(FooBar)
    except AssertionError:
        dup = True
    missing = False
    try:
        r.get("nope")
    except KeyError:
        missing = True
    torch.save({"keys": list(r.keys()), "get_FOOBAR": r.get("FOOBAR").__name__, "get_baz": r.get("baz").__name__, "contains_FooBar": "FooBar" in r,
                "duplicate_asserts": dup, "missing_keyerror": missing,
                "singletons": sorted(k for k in vars(reg) if k.endswith("_REGISTRY"))}, os.path.join(OUT, "registry.pt"))

    ds = _load("traiNNer.data.data_sampler", "traiNNer/data/data_sampler.py")
    out = {}
    data = list(range(10))
    for world in (1, 2, 4):
        for rank in range(world):
            s = ds.EnlargedSampler(data, world, rank, ratio=3)
            for epoch in (0, 1):
                s.set_epoch(epoch)
                out[f"{world}/{rank}/{epoch}"] = torch.tensor(list(iter(s)))
            out[f"{world}/{rank}/len"] = len(s)
    torch.save(out, os.path.join(OUT, "sampler.pt"))
    # ---- UNetDiscriminatorSN (discriminator_arch.py:98-160): training-mode forward (one power iteration) + backward -------
    da = _load("traiNNer.archs.discriminator_arch", "traiNNer/archs/discriminator_arch.py")
    _stub("traiNNer.archs")
    torch.manual_seed(77)
    net = da.UNetDiscriminatorSN(num_in_ch=3, num_feat=16, skip_connection=True)
    net.train()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.rand(2, 3, 32, 48).requires_grad_(True)
    y = net(x)
    wsum = torch.randn(y.shape)
    (y * wsum).sum().backward()
    fix = {"state_before": sd0, "x": x.detach(), "y": y.detach(), "gy": wsum, "gx": x.grad.clone(),
           "state_after": {k: v.clone() for k, v in net.state_dict().items()},
           "grads": {k: p.grad.clone() for k, p in net.named_parameters()}}
    net.eval()
    with torch.no_grad():
        fix["y_eval"] = net(x.detach())
    torch.save(fix, os.path.join(OUT, "unet_discriminator_sn.pt"))

    # ---- SPAN (span_arch.py is torch-only apart from spandrel's hyper-parameter recorder)
    _stub("spandrel")
    su = _stub("spandrel.util")
    su.store_hyperparameters = lambda *a, **k: (lambda cls: cls)
    sp = _load("traiNNer.archs.span_arch", "traiNNer/archs/span_arch.py") if "traiNNer.archs" in sys.modules else None
    if sp is None:
        _stub("traiNNer.archs")
        sp = _load("traiNNer.archs.span_arch", "traiNNer/archs/span_arch.py")
    fix = {}
    for tag, kw, shape in (("fc8_x2", dict(feature_channels=8, upscale=2, norm=False), (2, 3, 12, 10)),
                           ("fc12_x4_norm", dict(feature_channels=12, upscale=4, norm=True, img_range=255.0), (1, 3, 9, 16))):
        torch.manual_seed(7)
        net = sp.SPAN(num_in_ch=3, num_out_ch=3, **kw).train()
        # default init leaves every bias at nn.Conv2d's small uniform value; widen them so the bias-at-the-padded-border effect shows
        with torch.no_grad():
            for k, p_ in net.named_parameters():
                if k.endswith("bias"):
                    p_.mul_(4.0)
            for m in net.modules():
                if isinstance(m, sp.Conv3XC):
                    m.update_params()
        g = torch.Generator().manual_seed(11)
        x = torch.rand(*shape, generator=g)
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        y = net(x)
        gy = torch.randn(y.shape, generator=g) / y.numel()
        y.backward(gy)
        e = {"kwargs": kw, "x": x, "gy": gy, "state": sd0, "y_train": y.detach().clone(),
             "grads": {k: p_.grad.clone() for k, p_ in net.named_parameters() if p_.grad is not None}}
        net.eval()
        with torch.no_grad():
            e["y_eval"] = net(x)
        e["state_eval"] = {k: v.clone() for k, v in net.state_dict().items() if "eval_conv" in k}
        fix[tag] = e
    torch.save(fix, os.path.join(OUT, "span.pt"))

    # ---- filter2d (the module imports cv2 / numpy at the top; the function itself is pure torch)
    _stub("cv2")
    ipu = _load("traiNNer.utils.img_process_util", "traiNNer/utils/img_process_util.py")
    g = torch.Generator().manual_seed(5)
    fix = {}
    for tag, (b, c, h, w, k, shared) in {"shared5": (2, 3, 17, 23, 5, True), "per_sample21": (3, 3, 40, 36, 21, False), "single7": (1, 1, 9, 12, 7, True)}.items():
        img = torch.rand(b, c, h, w, generator=g)
        ker = torch.rand(1 if shared else b, k, k, generator=g)
        ker = ker / ker.sum(dim=(1, 2), keepdim=True)
        fix[tag] = {"img": img, "kernel": ker, "out": ipu.filter2d(img, ker)}
    torch.save(fix, os.path.join(OUT, "filter2d.pt"))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.endswith(".pt")))


if __name__ == "__main__":
    main()
