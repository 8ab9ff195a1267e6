"""SPAN on the HIP kernels against the REFERENCE MODULE's own tensors (tests/golden/span.pt, produced by tests/golden/make_golden.py from
/root/reference/traiNNer/archs/span_arch.py) and, for bf16, against the same-policy oracle (oracle/span.py, itself pinned to that fixture).

Tolerances: fp32 compute vs the reference: output rel-L2 <= 1e-5, every parameter gradient <= 1e-4; bf16: output <= 1e-2 vs the
bf16-policy oracle, gradient cosine >= 0.99 (end-to-end rounding-flip regime, see tests/test_esrgan_gpu.py)."""
import os
from collections import OrderedDict

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "span.pt")


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


def _net(cuda, e, dtype):
    from trainner_redux_amd.archs import build_network

    kw = e["kwargs"]
    net = build_network({"type": "span", "feature_channels": kw["feature_channels"], "scale": kw["upscale"], "norm": kw["norm"],
                         "img_range": kw.get("img_range", 255.0)})
    net.load_state_dict(e["state"], strict=True)
    net = net.to(cuda).train()
    net.compute_dtype = dtype
    return net


@pytest.mark.parametrize("tag", ["fc8_x2", "fc12_x4_norm"])
def test_span_fp32_matches_reference_module(cuda, tag):
    e = torch.load(GOLD, weights_only=True)[tag]
    net = _net(cuda, e, torch.float32)
    y = net.forward_hip(e["x"].to(cuda), train=True)
    e_y = rel_l2(y.cpu(), e["y_train"])
    assert y.shape == e["y_train"].shape and e_y < 1e-5, e_y
    net.backward_hip(e["gy"].to(cuda))
    errs = {k: rel_l2(p.grad.cpu(), e["grads"][k]) for k, p in net.named_parameters() if k in e["grads"]}
    assert set(errs) == set(e["grads"])
    worst = max(errs, key=errs.get)
    print(f"span fp32 {tag}: fwd {e_y:.2e}, worst grad {worst} {errs[worst]:.2e}")
    assert errs[worst] < 1e-4, (worst, errs[worst])
    net.eval()
    with torch.no_grad():
        ye = net(e["x"].to(cuda))
    assert rel_l2(ye.cpu(), e["y_eval"]) < 1e-5          # inference runs the same three-conv form; the reference ran the merged 3x3
    sd = net.state_dict()                                # exports the merged kernels like the reference's eval forward does
    for k, v in e["state_eval"].items():
        assert rel_l2(sd[k].cpu(), v) < 1e-5, k


def test_span_bf16_vs_policy_oracle(cuda):
    from oracle import span as OS
    from oracle.precision import BF16_OPS

    e = torch.load(GOLD, weights_only=True)["fc12_x4_norm"]
    kw = e["kwargs"]
    hp = OS.hparams(feature_channels=kw["feature_channels"], scale=kw["upscale"], norm=kw["norm"], img_range=kw["img_range"])
    params = OrderedDict((k, v.clone().requires_grad_("eval_conv" not in k)) for k, v in e["state"].items() if k != "no_norm")
    x = torch.rand(2, 3, 16, 24, generator=torch.Generator().manual_seed(5))
    y_ref = OS.forward(params, x, hp, BF16_OPS, train=True)
    gy = torch.randn(y_ref.shape, generator=torch.Generator().manual_seed(6)) / y_ref.numel()
    names = [k for k in params if "eval_conv" not in k]
    g_ref = dict(zip(names, torch.autograd.grad(y_ref, [params[k] for k in names], gy)))
    net = _net(cuda, e, torch.bfloat16)
    y = net.forward_hip(x.to(cuda), train=True)
    e_y = rel_l2(y.cpu(), y_ref.detach())
    net.backward_hip(gy.to(cuda))
    cs = {k: cosine(p.grad.cpu(), g_ref[k]) for k, p in net.named_parameters() if k in g_ref}
    worst = min(cs, key=cs.get)
    print(f"span bf16: fwd {e_y:.2e}, worst grad cosine {worst} {cs[worst]:.4f}")
    assert e_y < 1e-2 and cs[worst] > 0.99


def test_span_module_surface_and_full_size(cuda):
    """autograd bridge; registry names / checksums; span_s 4x at 64 -> 256, batch 4: one training step's worth of kernels, finite."""
    from trainner_redux_amd.archs import build_network

    net = build_network({"type": "span_s", "scale": 4}).to(cuda).train()
    assert sum(p.numel() for p in net.parameters()) == 2_236_728          # docs/source/benchmarks.md:25
    x = torch.rand(4, 3, 64, 64, device=cuda)
    y = net(x)
    assert y.shape == (4, 3, 256, 256) and y.requires_grad
    (y * y).mean().backward()
    for k, p in net.named_parameters():
        if "eval_conv" in k:
            assert not p.requires_grad
        else:
            assert torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0, k
    net2 = build_network({"type": "span", "scale": 2, "norm": True}).to(cuda).eval()
    assert sum(p.numel() for p in net2.parameters()) > 0 and net2(torch.rand(1, 3, 20, 28, device=cuda)).shape == (1, 3, 40, 56)
