"""Oracle restatement of the ESRGAN / RRDBNet generator (test infrastructure, see oracle/__init__.py).

Follows:
  * factory semantics     /root/reference/traiNNer/archs/rrdbnet_arch.py:10-56
                          (pixel-unshuffle front for scale 1/2, esrgan_lite = 32 filters x 12 blocks)
  * RDB / RRDB arithmetic /root/reference/traiNNer/archs/scunet_aaf6aa_arch.py:310-360
                          (5 dense convs, LeakyReLU(0.2) on the first four, x5*0.2+x, out*0.2+x)
  * nearest-upconv tail   /root/reference/traiNNer/archs/scunet_aaf6aa_arch.py:363-384
  * parameter checksums   /root/reference/docs/source/benchmarks.md:64 (16,697,987),
                          :221 (16,703,171), :42 (5,021,155)

The network body itself lives in spandrel (`spandrel.architectures.ESRGAN`, pip `spandrel>=0.4.0`,
/root/reference/pyproject.toml:21), which is NOT in /root/reference and not installable here, so its
published "old-arch" topology is restated: numeric values are *parity unpinned*; structure is pinned
by the checksums above.  State-dict keys use spandrel's old-arch layout
(`model.0`, `model.1.sub.{i}.RDB{1..3}.conv{1..5}.0`, `model.1.sub.{nb}`, `model.3/6/8/10`) so
checkpoints interchange with the reference (base_model.py:590-605 loads ESRGAN through spandrel).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch
from torch.nn import functional as F

from .precision import FP32, Policy

SLOPE = 0.2
RES_SCALE = 0.2


def esrgan_hparams(scale=4, use_pixel_unshuffle=True, in_nc=3, out_nc=3, num_filters=64, num_blocks=23):
    """rrdbnet_arch.py:10-37 -> dict(in_nc, out_nc, nf, nb, gc, net_scale, shuffle_factor)."""
    shuffle = None
    net_scale = scale
    if use_pixel_unshuffle and scale in (1, 2):
        in_nc = in_nc * 4 ** (3 - scale)
        shuffle = int(math.sqrt(in_nc / out_nc))
        net_scale = 4
    return dict(in_nc=in_nc, out_nc=out_nc, nf=num_filters, nb=num_blocks, gc=32,
                net_scale=net_scale, shuffle_factor=shuffle, scale=scale)


def layer_table(hp) -> list[tuple[str, int, int]]:
    """(state-dict prefix, c_in, c_out) of every 3x3 conv in forward order."""
    nf, gc, nb = hp["nf"], hp["gc"], hp["nb"]
    t = [("model.0", hp["in_nc"], nf)]
    for i in range(nb):
        for r in (1, 2, 3):
            for c in range(1, 6):
                cin = nf + (c - 1) * gc
                cout = gc if c < 5 else nf
                t.append((f"model.1.sub.{i}.RDB{r}.conv{c}.0", cin, cout))
    t.append((f"model.1.sub.{nb}", nf, nf))
    idx = 2
    for _ in range(int(math.log2(hp["net_scale"]))):  # [Upsample, conv, lrelu] flattened
        t.append((f"model.{idx + 1}", nf, nf))
        idx += 3
    t.append((f"model.{idx}", nf, nf))          # HR conv (+ lrelu at idx+1)
    t.append((f"model.{idx + 2}", nf, hp["out_nc"]))  # conv_last
    return t


def init_state_dict(hp, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """torch.nn.Conv2d default init (kaiming_uniform(a=sqrt(5)) + uniform bias), seeded."""
    g = torch.Generator().manual_seed(seed)
    sd: OrderedDict[str, torch.Tensor] = OrderedDict()
    for name, cin, cout in layer_table(hp):
        bound = 1.0 / math.sqrt(cin * 9)
        sd[name + ".weight"] = (torch.rand(cout, cin, 3, 3, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound
    return sd


def param_count(hp) -> int:
    return sum(cout * cin * 9 + cout for _, cin, cout in layer_table(hp))


def forward(sd, x: torch.Tensor, hp, policy: Policy = FP32, taps: dict | None = None) -> torch.Tensor:
    """x: N x in_nc0 x h x w in [0,1] -> N x out_nc x (h*scale) x (w*scale).

    `taps` (optional dict) receives a few intermediate activations for layer-wise parity tests.
    """
    conv = lambda name, t: policy.conv2d(t, sd[name + ".weight"], sd[name + ".bias"], 1, 1)  # noqa: E731
    lrelu = lambda t: F.leaky_relu(t, SLOPE)  # noqa: E731
    nb = hp["nb"]
    if hp["shuffle_factor"]:
        x = F.pixel_unshuffle(x, hp["shuffle_factor"])
    fea = conv("model.0", x)
    if taps is not None:
        taps["conv_first"] = fea
    t = fea
    for i in range(nb):
        rrdb_in = t
        for r in (1, 2, 3):
            p = f"model.1.sub.{i}.RDB{r}."
            x0 = t
            x1 = lrelu(conv(p + "conv1.0", x0))
            x2 = lrelu(conv(p + "conv2.0", torch.cat((x0, x1), 1)))
            x3 = lrelu(conv(p + "conv3.0", torch.cat((x0, x1, x2), 1)))
            x4 = lrelu(conv(p + "conv4.0", torch.cat((x0, x1, x2, x3), 1)))
            x5 = conv(p + "conv5.0", torch.cat((x0, x1, x2, x3, x4), 1))
            t = x5 * RES_SCALE + x0
        t = t * RES_SCALE + rrdb_in
        if taps is not None and i in (0, nb - 1):
            taps[f"rrdb{i}"] = t
    t = fea + conv(f"model.1.sub.{nb}", t)
    if taps is not None:
        taps["trunk"] = t
    idx = 2
    for _ in range(int(math.log2(hp["net_scale"]))):
        t = lrelu(conv(f"model.{idx + 1}", F.interpolate(t, scale_factor=2, mode="nearest")))
        idx += 3
    t = lrelu(conv(f"model.{idx}", t))
    return conv(f"model.{idx + 2}", t)
