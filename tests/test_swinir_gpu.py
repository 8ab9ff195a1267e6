"""SwinIR: HIP forward + hand-scheduled backward against the oracle (oracle/swinir.py, pinned to the reference's parameter
counts docs/source/benchmarks.md:71,75; the reference's own SwinIR lives in the absent `spandrel` wheel, so arithmetic parity is
anchored on the in-tree twin drct_arch.py the oracle restates -- "parity unpinned" beyond the checksums, see oracle/__init__.py).

Tolerances: fp32 compute vs the fp32 oracle: rel-L2 <= 1e-5 forward, <= 2e-4 every gradient tensor.  bf16 compute vs the
same-policy oracle (bf16 MFMA operands, fp32 accumulate / LayerNorm / softmax / residual stream): <= 1e-2 forward, gradients
cosine >= 0.99 (the end-to-end rounding-flip regime of tests/test_esrgan_gpu.py's docstring applies here as well).
"""
from collections import OrderedDict

import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


CFGS = {
    "s_like": ("swinir_s", dict(scale=4, depths=[2, 2], num_heads=[6, 6])),
    "m_like": ("swinir_m", dict(scale=4, depths=[2, 1], num_heads=[6, 6])),
    "m_x2": ("swinir_m", dict(scale=2, depths=[2], num_heads=[6])),
    "s_x2_heads3": ("swinir_s", dict(scale=2, depths=[3], num_heads=[3])),
}


def _build(cuda, name, over, compute_dtype, seed=0):
    from oracle import swinir as OS
    from trainner_redux_amd.archs import build_network

    hp = OS.hparams(name, **over)
    sd = OS.init_state_dict(hp, seed)
    net = build_network({"type": name, **over})
    res = net.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all("relative_position_index" in k or "attn_mask" in k for k in res.missing_keys), res
    net = net.to(cuda)
    net.compute_dtype = compute_dtype
    net.train()
    return hp, sd, net


def _oracle_grads(hp, sd, x, gt, policy, keep):
    from oracle import losses as OL
    from oracle import swinir as OS

    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    out = OS.forward(params, x, hp, policy, keep_masks=keep)
    loss = OL.l1(out, gt)
    grads = torch.autograd.grad(loss, [out, *params.values()])
    return out.detach(), grads[0], OrderedDict(zip(params.keys(), grads[1:]))


def _keep(hp, b):
    # DropPath keep masks as the trainer draws them: 0 or 1/keep_prob per (block, sample); block (0, 1) drops sample 0
    k = {(0, 1): torch.tensor([0.0] + [1.0 / 0.9] * (b - 1)), (0, 0): torch.tensor([1.0 / 0.95] * b)}
    if len(hp["depths"]) > 1:
        k[(1, 0)] = torch.tensor([1.0 / 0.8] * (b - 1) + [0.0])
    return k


@pytest.mark.parametrize("cfg", list(CFGS))
def test_swinir_fp32_forward_backward(cuda, cfg):
    from oracle.precision import FP32

    name, over = CFGS[cfg]
    hp, sd, net = _build(cuda, name, over, torch.float32)
    g = torch.Generator().manual_seed(1)
    s = hp["scale"]
    x = torch.rand(2, 3, 16, 24, generator=g)
    gt = torch.rand(2, 3, 16 * s, 24 * s, generator=g)
    keep = _keep(hp, 2)
    ref_out, ref_gout, ref_g = _oracle_grads(hp, sd, x, gt, FP32, keep)
    out = net.forward_hip(x.to(cuda), train=True, keep_masks={k: v.to(cuda) for k, v in keep.items()})
    assert out.shape == ref_out.shape
    e = rel_l2(out.cpu(), ref_out)
    assert e < 1e-5, e
    net.backward_hip(ref_gout.to(cuda))
    errs = {k: rel_l2(p.grad.cpu(), ref_g[k]) for k, p in net.named_parameters()}
    worst = max(errs, key=errs.get)
    print(f"fp32 {cfg}: fwd rel-L2 {e:.2e}; worst grad {worst} {errs[worst]:.2e}")
    assert errs[worst] < 2e-4, (worst, errs[worst])


@pytest.mark.parametrize("cfg", ["s_like", "m_like"])
def test_swinir_bf16_forward_backward(cuda, cfg):
    from oracle.precision import BF16_OPS

    name, over = CFGS[cfg]
    hp, sd, net = _build(cuda, name, over, torch.bfloat16)
    g = torch.Generator().manual_seed(2)
    s = hp["scale"]
    x = torch.rand(2, 3, 24, 16, generator=g)
    gt = torch.rand(2, 3, 24 * s, 16 * s, generator=g)
    ref_out, ref_gout, ref_g = _oracle_grads(hp, sd, x, gt, BF16_OPS, None)
    out = net.forward_hip(x.to(cuda), train=True, keep_masks={})
    e = rel_l2(out.cpu(), ref_out)
    assert e < 1e-2, e
    net.backward_hip(ref_gout.to(cuda))
    cs = {k: cosine(p.grad.cpu(), ref_g[k]) for k, p in net.named_parameters()}
    worst = min(cs, key=cs.get)
    print(f"bf16 {cfg}: fwd rel-L2 vs bf16-policy oracle {e:.2e}; worst grad cosine {worst} {cs[worst]:.4f}")
    assert cs[worst] > 0.99, (worst, cs[worst])


def test_swinir_module_surface(cuda):
    """`net(x)` + `loss.backward()` (autograd bridge), eval path == train path without DropPath, deepcopy (EMA twin), state-dict keys."""
    import copy

    name, over = CFGS["s_like"]
    hp, sd, net = _build(cuda, name, dict(over, drop_path_rate=0.0), torch.float32)
    keys = set(net.state_dict().keys())
    assert set(sd.keys()) <= keys
    assert "layers.0.residual_group.blocks.1.attn_mask" in keys and "layers.0.residual_group.blocks.0.attn.relative_position_index" in keys
    x = torch.rand(1, 3, 16, 16, generator=torch.Generator().manual_seed(3)).to(cuda)
    y = net(x)
    assert y.requires_grad and y.shape == (1, 3, 64, 64)
    (y * y).mean().backward()
    assert all(torch.isfinite(p.grad).all() and p.grad.abs().sum() > 0 for p in net.parameters())
    twin = copy.deepcopy(net).eval()
    with torch.no_grad():
        assert torch.equal(twin(x), y.detach())
    net.eval()
    assert torch.equal(net(x), y.detach())
    with pytest.raises(NotImplementedError):
        net(torch.rand(1, 3, 20, 16, device=cuda))


def test_swinir_drop_path_draws(cuda):
    """Training mode draws per-sample keep masks with the linearly rising rate (drct_arch.py:572-573): outputs differ run to run and
    from eval mode, and every drawn factor is 0 or 1/(1-p)."""
    name, over = CFGS["s_like"]
    hp, sd, net = _build(cuda, name, dict(over, drop_path_rate=0.5), torch.float32)
    torch.manual_seed(0)
    masks = net._keep_masks(64, cuda)
    assert (0, 0) not in masks                      # first block: rate 0
    last = masks[(1, 1)]
    assert set(last.unique().tolist()) <= {0.0, 2.0} and 0.2 < (last == 0).float().mean() < 0.8
    x = torch.rand(4, 3, 16, 16, device=cuda)
    a = net.forward_hip(x, train=True)
    b = net.forward_hip(x, train=True)
    assert not torch.equal(a, b)


def test_swinir_full_size_step_runs(cuda):
    """BASELINE.json configs[3] shape (swinir_m x4, 64x64 tiles, bf16) at batch 2: one forward/backward, finite gradients."""
    from trainner_redux_amd.archs import build_network

    net = build_network({"type": "swinir_m", "scale": 4}).to(cuda).train()
    assert sum(p.numel() for p in net.parameters()) == 11_900_199     # docs/source/benchmarks.md:71
    x = torch.rand(2, 3, 64, 64, device=cuda)
    out = net.forward_hip(x, train=True)
    assert out.shape == (2, 3, 256, 256) and torch.isfinite(out).all()
    net.backward_hip(torch.randn_like(out) / out.numel())
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())


def test_swinir_trainer_replays_the_step_as_hip_graphs(cuda, monkeypatch):
    """SwinIR asks the trainer for hipGraph replay of its step (archs/swinir_arch.py `prefer_hip_graphs`; `use_compile` for the other generators).
    (a) without DropPath: six steps replayed == six steps launched eagerly, bit for bit; (b) with DropPath (the presets' drop_path_rate 0.1, raised
    here): the keep masks are drawn on the device INSIDE the captured forward, every replay draws new ones, the backward graph reads the same buffer."""
    from tests.test_sr_model_gpu import _data, _opt
    from trainner_redux_amd.models import SRModel

    def run(graphs: bool, dpr: float, steps: int):
        monkeypatch.setenv("TRAINNER_HIP_GRAPHS", "1" if graphs else "0")
        torch.manual_seed(5)
        model = SRModel(_opt([dict(type="l1loss", loss_weight=1.0)], use_amp=True,
                             network_g=dict(type="swinir_s", depths=[2, 2], num_heads=[6, 6], drop_path_rate=dpr)))
        assert bool(model.net_g.hip_graphs) == graphs
        masks, modes = [], []
        for it in range(1, steps + 1):
            lq, gt = _data(seed=it, n=2, s=16)
            model.feed_data({"lq": lq, "gt": gt})
            model.optimize_parameters(it, 0, True)
            modes.append((model.net_g._gs or {}).get("mode"))
            if dpr > 0:
                masks.append(model.net_g._keep_state["keep"].clone())
        log = model.get_current_log()
        return model.net_g.flat_parameters()[0].detach().cpu().clone(), log, masks, modes

    w_g, log_g, _, modes = run(True, 0.0, 6)
    assert modes[:2] == ["eager", "eager"] and modes[2:] == ["graph"] * 4, modes
    w_e, log_e, _, modes_e = run(False, 0.0, 6)
    assert all(m is None for m in modes_e)
    assert torch.equal(w_g, w_e) and log_g == log_e
    w_d, log_d, masks, modes = run(True, 0.6, 6)
    assert modes[2:] == ["graph"] * 4 and torch.isfinite(w_d).all() and all(v == v for v in log_d.values())
    distinct = {tuple(m.flatten().tolist()) for m in masks[2:]}
    assert len(distinct) == 4, "every replay must draw fresh DropPath masks"
    k = masks[-1]
    assert ((k == 0) | (k >= 1.0)).all() and (k == 0).any() and (k > 0).any()
