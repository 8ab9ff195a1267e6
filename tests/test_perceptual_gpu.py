"""VGG19 perceptual head on HIP vs oracle/vgg.py (same synthetic weights): per-layer values and d(loss)/d(x).

`perceptualloss` (fp32 forced like the reference) is held to 2e-5 / 2e-4; `perceptualfp16loss` under bf16 AMP is compared
with the same-policy oracle and lands in the end-to-end bf16 regime explained in tests/test_esrgan_gpu.py (16 convs deep:
a few 1e-3 on the deepest layer's value), so its bounds are 5e-3 / 5e-2."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("criterion", ["charbonnier", "l1"])
@pytest.mark.parametrize("name,dtype,tol_v,tol_g,wino", [("perceptualloss", torch.bfloat16, 2e-5, 2e-4, "1"), ("perceptualloss", torch.bfloat16, 2e-5, 2e-4, "2"),
                                                       ("perceptualloss", torch.bfloat16, 2e-5, 2e-4, "0"), ("perceptualfp16loss", torch.bfloat16, 5e-3, 5e-2, "1")])
def test_perceptual_value_and_grad(cuda, criterion, name, dtype, tol_v, tol_g, wino, monkeypatch):
    monkeypatch.setenv("TRAINNER_VGG_WINOGRAD", wino)       # "1": F(4x4,3x3) + fused pool epilogues (default); "2": F(2x2,3x3); "0": direct fp32 MFMA convs
    from oracle import vgg as OV
    from oracle.precision import BF16_OPS, FP32
    from trainner_redux_amd.losses import build_loss

    sd = OV.synthetic_state_dict(3)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 32, 48, generator=g).requires_grad_(True)
    gt = torch.rand(2, 3, 32, 48, generator=g)
    policy = FP32 if name == "perceptualloss" else BF16_OPS
    ref = OV.perceptual_loss(sd, x, gt, criterion=criterion, policy=policy)
    w, acc = 0.01, 1
    tot_ref = sum(v * w * acc for v in ref.values())
    (gref,) = torch.autograd.grad(tot_ref, x)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loss = build_loss(dict(type=name, loss_weight=w, criterion=criterion)).to(cuda)
    loss.vgg.load_torchvision_state_dict(sd)
    loss.set_compute_dtype(dtype)   # the AMP dtype: perceptualloss must ignore it (fp32 forced)
    assert loss.compute_dtype == (torch.float32 if name == "perceptualloss" else torch.bfloat16)
    vals = torch.zeros(8, device=cuda)
    total = torch.zeros(1, device=cuda)
    xd, gtd = x.detach().to(cuda), gt.to(cuda)
    grad = torch.full_like(xd, 3.0)
    names = loss.hip_value_and_grad(xd, gtd, value_weight=w, value_out=vals, total_weight=w * acc, total_out=total, grad_weight=w * acc,
                                    grad=grad, first=True)
    assert names == list(ref.keys())
    for j, k in enumerate(names):
        r = float(ref[k]) * w
        assert abs(vals[j].item() - r) <= tol_v * abs(r), (k, vals[j].item(), r)
    assert abs(total.item() - float(tot_ref)) <= tol_v * float(tot_ref)
    e = rel_l2(grad.cpu(), gref)
    cos = torch.nn.functional.cosine_similarity(grad.cpu().flatten().double(), gref.flatten().double(), dim=0).item()
    print(f"{name}/{criterion}: grad rel-L2 {e:.2e} cos {cos:.5f}")
    # max-pool routing is discontinuous: one 2x2 window whose two largest entries differ by less than the arithmetic noise
    # re-routes a whole gradient path (such a window was found on this draw: 0.34001 vs 0.34000).  fp32 noise (1e-6) makes
    # that rare; bf16 operands (4e-3) make it routine, so the bf16 head is held to direction (cosine), not to rounding.
    if dtype == torch.bfloat16 and name != "perceptualloss":
        assert cos > 0.97 and e < 0.3, (cos, e)
    elif wino != "1":
        # fp32 head on the F(2x2, 3x3) / direct kernels (4e-7 of rounding noise per layer): no window flips on this draw, the gradient is held to the
        # rounding-level bound (measured 5e-5).  Same program, same pool / loss kernels as the default path apart from the fused-pool epilogues.
        assert e < tol_g and cos > 0.999999, (cos, e)
    else:
        # fp32 head, default path: the F(4x4, 3x3) Winograd convs carry ~2e-6 of rounding noise per layer, enough to flip the odd near-tied 2x2
        # window on this draw, which re-routes one gradient path (measured rel-L2 1.2e-3, cosine 1.00000).  A flip is LOCAL (one receptive field);
        # a routing / mask bug in the pool-fused epilogue or in tr_pixel_loss_pool_bwd would be everywhere.  So: direction to 1e-5, size to 5e-3, and
        # all but a small fraction of the elements at rounding level (measured: 3.3 % of the elements sit in the flipped window's receptive field).
        off = ((grad.cpu() - gref).abs() > 2e-4 * gref.abs().max()).float().mean().item()
        print(f"  elements off by more than 2e-4 of the largest gradient: {off:.4f}")
        assert cos > 0.99999 and e < 5e-3 and off < 0.08, (cos, e, off)
    g1 = grad.clone()
    # accumulate form: a second call adds onto the existing gradient
    loss.hip_value_and_grad(xd, gtd, value_weight=w, value_out=vals, total_weight=0.0, total_out=None, grad_weight=w * acc, grad=grad,
                            first=False)
    assert rel_l2(grad, 2 * g1) < 1e-6
    # reference invariant (tests/test_losses/test_losses.py:114-128): identical inputs give ~0
    z = torch.zeros(1, 3, 32, 32, device=cuda)
    loss.hip_value_and_grad(z, z, value_weight=1.0, value_out=vals, total_weight=0.0, total_out=None, grad_weight=0.0, grad=None, first=True)
    assert vals[:5].abs().max().item() <= 1e-5


def test_perceptual_autograd_bridge(cuda):
    import warnings

    from oracle import vgg as OV
    from trainner_redux_amd.losses import build_loss

    sd = OV.synthetic_state_dict(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loss = build_loss(dict(type="perceptualloss", loss_weight=1.0)).to(cuda)
    loss.vgg.load_torchvision_state_dict(sd)
    g = torch.Generator().manual_seed(5)   # a draw without a near-tied max-pool window (see the comment above)
    x = torch.rand(2, 3, 32, 48, generator=g)
    gt = torch.rand(2, 3, 32, 48, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = OV.perceptual_loss(sd, xr, gt)
    sum(ref.values()).backward()
    xd = x.to(cuda).requires_grad_(True)
    out = loss(xd, gt.to(cuda))
    assert list(out.keys()) == list(ref.keys())
    for k in ref:
        assert abs(float(out[k]) - float(ref[k])) < 2e-5 * float(ref[k])
    (sum(out.values()) * 2.0).backward()   # upstream scale goes through the per-layer scales of the bridge
    assert rel_l2(xd.grad.cpu(), 2 * xr.grad) < 5e-3      # max-pool re-routing under the F(4x4, 3x3) rounding noise, see test_perceptual_value_and_grad


def test_perceptual_fast_matmul_split_bf16(cuda):
    """`fast_matmul: true` (the reference's "trade precision for performance" switch; TF32 on its CUDA targets) = split-bf16 MFMAs in the
    fp32-forced VGG head: per-layer values within 5e-5 of the fp32 oracle (measured 5e-6; total 5e-7), d(loss)/dx cosine >= 0.9999 / rel-L2 <= 2e-2 (a near-tied max-pool
    window re-routes one gradient path under 3e-6 of noise, see above: which window flips depends on the last bits of conv1_1, measured 0.9e-2 with the
    MFMA stem and 1.3e-2 with the exact VALU stem kernel)
    -- two orders of magnitude tighter than the bf16 head, far inside north_star's 1e-3 on the loss."""
    import warnings

    from oracle import vgg as OV
    from oracle.precision import FP32
    from trainner_redux_amd.losses import build_loss

    sd = OV.synthetic_state_dict(3)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 32, 48, generator=g).requires_grad_(True)
    gt = torch.rand(2, 3, 32, 48, generator=g)
    ref = OV.perceptual_loss(sd, x, gt, criterion="charbonnier", policy=FP32)
    w = 0.01
    tot_ref = sum(v * w for v in ref.values())
    (gref,) = torch.autograd.grad(tot_ref, x)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        loss = build_loss(dict(type="perceptualloss", loss_weight=w, criterion="charbonnier")).to(cuda)
    loss.vgg.load_torchvision_state_dict(sd)
    loss.set_compute_dtype(torch.bfloat16)
    loss.set_fast_matmul(True)
    vals, total = torch.zeros(8, device=cuda), torch.zeros(1, device=cuda)
    xd, gtd = x.detach().to(cuda), gt.to(cuda)
    grad = torch.zeros_like(xd)
    names = loss.hip_value_and_grad(xd, gtd, value_weight=w, value_out=vals, total_weight=w, total_out=total, grad_weight=w, grad=grad, first=True)
    worst = max(abs(vals[j].item() - float(ref[k]) * w) / (float(ref[k]) * w) for j, k in enumerate(names))
    e = rel_l2(grad.cpu(), gref)
    print(f"fast_matmul perceptual: worst layer value rel {worst:.2e}, total rel {abs(total.item() - float(tot_ref)) / float(tot_ref):.2e}, grad rel-L2 {e:.2e}")
    cos = torch.nn.functional.cosine_similarity(grad.cpu().flatten().double(), gref.flatten().double(), dim=0).item()
    assert worst < 5e-5 and e < 2e-2 and cos > 0.9999, (worst, e, cos)
