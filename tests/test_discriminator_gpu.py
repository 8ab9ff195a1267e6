"""UNetDiscriminatorSN + GANLoss on HIP against the REFERENCE's own tensors (tests/golden/*.pt, produced by running
/root/reference/traiNNer/archs/discriminator_arch.py and losses/gan_loss.py in the build container)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel_l2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def test_gan_loss_matches_reference_golden(cuda):
    from trainner_redux_amd.losses import build_loss

    fix = torch.load(os.path.join(GOLD, "pixel_gan_losses.pt"), weights_only=True)
    logits = fix["logits"].to(cuda)
    for key, ref in fix["gan"].items():
        gt_, real, disc = key.split("/")
        loss = build_loss(dict(type="ganloss", loss_weight=0.1, gan_type=gt_))
        x = logits.clone().requires_grad_(True)
        v = loss(x, bool(int(real)), is_disc=bool(int(disc)))
        assert abs(v.item() - ref.item()) <= 2e-6 * max(abs(ref.item()), 1e-3), (key, v.item(), ref.item())
        v.backward()
        xr = fix["logits"].clone().requires_grad_(True)
        from oracle import losses as OL
        OL.gan(xr, bool(int(real)), is_disc=bool(int(disc)), gan_type=gt_).backward()
        assert rel_l2(x.grad.cpu(), xr.grad) < 2e-6, key


def test_small_kernels_of_the_discriminator(cuda):
    from torch.nn import functional as F

    from trainner_redux_amd import ops

    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 8, 12, 20, generator=g)
    xd = x.permute(0, 2, 3, 1).contiguous().to(cuda)
    # bilinear x2 and its adjoint
    up = torch.empty(2, 24, 40, 8, device=cuda)
    ops.bilinear_up2x(ops.view(xd), ops.view(up), 2, 12, 20)
    xr = x.clone().requires_grad_(True)
    ref = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=False)
    assert rel_l2(up.permute(0, 3, 1, 2).cpu(), ref.detach()) < 1e-6
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    gyd = gy.permute(0, 2, 3, 1).contiguous().to(cuda)
    gx = torch.empty_like(xd)
    ops.bilinear_up2x_bwd(ops.view(gyd), ops.view(gx), 2, 12, 20)
    assert rel_l2(gx.permute(0, 3, 1, 2).cpu(), xr.grad) < 1e-6
    # space-to-depth round trip and the 4x4-stride-2 == 3x3-over-s2d identity
    s = torch.empty(2, 6, 10, 32, device=cuda)
    ops.space_to_depth2(ops.view(xd), ops.view(s), 2, 12, 20)
    back = torch.empty_like(xd)
    ops.depth_to_space2(ops.view(s), ops.view(back), 2, 12, 20)
    assert torch.equal(back, xd)
    w4 = torch.randn(16, 8, 4, 4, generator=g)
    w3 = torch.empty(16, 32, 3, 3, device=cuda)
    w4d = w4.to(cuda)
    ops.expand_w4x4s2(w4d, w3)
    ref4 = F.conv2d(x, w4, None, stride=2, padding=1)
    got = F.conv2d(s.permute(0, 3, 1, 2).cpu(), w3.cpu(), None, padding=1)
    assert rel_l2(got, ref4) < 1e-6
    g3 = torch.randn(16, 32, 3, 3, generator=g).to(cuda)
    g4 = torch.empty(16, 8, 4, 4, device=cuda)
    ops.collapse_g3x3s2d(g3, g4)
    # adjoint identity: <expand(w4), g3> == <w4, collapse(g3)>
    assert abs((w3 * g3).sum().item() - (w4d * g4).sum().item()) < 1e-3
    # spectral norm forward / backward against autograd
    from oracle import discriminator as OD

    w = torch.randn(24, 10, 3, 3, generator=g)
    u = F.normalize(torch.randn(24, generator=g), dim=0)
    v = F.normalize(torch.randn(90, generator=g), dim=0)
    wr = w.clone().requires_grad_(True)
    wsn, u2, v2 = OD.spectral_weight(wr, u, v, training=True)
    G = torch.randn(wsn.shape, generator=g)
    (wsn * G).sum().backward()
    wd, ud, vd = w.to(cuda), u.to(cuda), v.to(cuda)
    out, sig = torch.empty_like(wd), torch.zeros(1, device=cuda)
    ops.spectral_norm(wd, ud, vd, out, sig, power_iteration=True)
    assert rel_l2(out.cpu(), wsn.detach()) < 1e-6 and rel_l2(ud.cpu(), u2) < 1e-6 and rel_l2(vd.cpu(), v2) < 1e-6
    gw = torch.zeros_like(wd)
    Gd = G.to(cuda)
    ops.spectral_norm_bwd(Gd, out, ud, vd, sig, gw)
    assert rel_l2(gw.cpu(), wr.grad) < 1e-5


@pytest.mark.parametrize("compute", [torch.float32, torch.bfloat16])
def test_unet_discriminator_matches_reference_golden(cuda, compute):
    from trainner_redux_amd.archs import build_network

    fix = torch.load(os.path.join(GOLD, "unet_discriminator_sn.pt"), weights_only=True)
    net = build_network(dict(type="unetdiscriminatorsn", num_in_ch=3, num_feat=16, skip_connection=True))
    assert list(net.state_dict().keys()) == list(fix["state_before"].keys())
    net.load_state_dict(fix["state_before"])
    net = net.to(cuda)
    net.compute_dtype = compute
    net.train()
    x = fix["x"].to(cuda)
    y = net.forward_hip(x, train=True)
    tol = 2e-5 if compute == torch.float32 else 2e-2
    assert rel_l2(y.cpu(), fix["y"]) < tol, rel_l2(y.cpu(), fix["y"])
    sd = net.state_dict()
    for k in fix["state_after"]:
        if k.endswith(("weight_u", "weight_v")):   # the power iteration ran in fp32 in both
            assert rel_l2(sd[k].cpu(), fix["state_after"][k]) < 1e-5, k
    gx = torch.zeros(2, 3, 32, 48, device=cuda)
    net.backward_hip(fix["gy"].to(cuda), param_grads=True, input_grad=gx, input_grad_accumulate=False)
    # bf16 here is compared with the fp32 reference itself (not a same-policy oracle): activations, skip sums and gradient
    # operands are all bf16 in this net, and the weight gradients of the deepest layers collect that noise (measured 0.06-0.09)
    gtol = 2e-4 if compute == torch.float32 else 0.15
    assert rel_l2(gx.cpu(), fix["gx"]) < gtol, rel_l2(gx.cpu(), fix["gx"])
    worst = max((rel_l2(p.grad.cpu(), fix["grads"][k]), k) for k, p in net.named_parameters())
    print(compute, "fwd", rel_l2(y.cpu(), fix["y"]), "gx", rel_l2(gx.cpu(), fix["gx"]), "worst param grad", worst)
    assert worst[0] < gtol, worst
    if compute == torch.float32:
        net.eval()
        ye = net.forward_hip(x, train=False)
        assert rel_l2(ye.cpu(), fix["y_eval"]) < 2e-5
        # autograd bridge: stock-torch loss on top, gradient to the image
        net.train()
        xi = x.clone().requires_grad_(True)
        net(xi).mean().backward()
        assert xi.grad is not None and torch.isfinite(xi.grad).all() and xi.grad.abs().sum() > 0


def test_vgg_style_discriminator_matches_reference_module(cuda):
    """`vggstylediscriminator` on HIP against the REFERENCE MODULE's own tensors (tests/golden/vgg_style_discriminator.pt, made by
    tests/golden/make_golden_twins.py from archs/discriminator_arch.py:9-95 loaded by path): training-mode logits (BatchNorm on batch statistics),
    input gradient and every parameter gradient, running statistics after the forward, eval-mode logits on the updated statistics.  fp32 compute;
    ten conv + BatchNorm stages deep, so the bound is the fp32-network bound of the other archs (1e-5 forward, 2e-4 gradients)."""
    import os

    from trainner_redux_amd.archs import build_network

    e = torch.load(os.path.join(os.path.dirname(__file__), "golden", "vgg_style_discriminator.pt"), weights_only=True)["nf8_128"]
    net = build_network({"type": "vggstylediscriminator", **e["kwargs"]})
    assert sum(p.numel() for p in net.parameters()) == e["n_params"]
    assert list(net.state_dict().keys()) == list(e["state"].keys())
    net.load_state_dict(e["state"], strict=True)
    net = net.to(cuda).train()
    net.compute_dtype = torch.float32
    out = net.forward_hip(e["x"].to(cuda), train=True)
    e_out = rel_l2(out.cpu(), e["out"])
    assert out.shape == e["out"].shape and e_out < 1e-5, e_out
    gx = torch.zeros_like(e["x"]).to(cuda)
    net.backward_hip(e["cotangent"].to(cuda), param_grads=True, input_grad=gx, input_grad_accumulate=False)
    e_gx = rel_l2(gx.cpu(), e["gx"])
    errs = {k: rel_l2(p.grad.cpu(), e["grads"][k]) for k, p in net.named_parameters()}
    worst = max(errs, key=errs.get)
    print(f"VGGStyleDiscriminator vs reference module: logits {e_out:.2e}, input gradient {e_gx:.2e}, worst parameter gradient {worst} {errs[worst]:.2e}")
    assert e_gx < 2e-4 and errs[worst] < 2e-4, (e_gx, worst, errs[worst])
    sd = net.state_dict()
    for k, v in e["after"].items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v), k
        else:
            assert rel_l2(sd[k].cpu(), v) < 1e-5, k
    net.eval()
    with torch.no_grad():
        oe = net(e["x"].to(cuda))
    assert rel_l2(oe.cpu(), e["out_eval"]) < 1e-5
    # the autograd bridge (what a stock GAN loss on top of it uses)
    net.train()
    xr = e["x"].to(cuda).requires_grad_(True)
    net(xr).backward(e["cotangent"].to(cuda))
    assert rel_l2(xr.grad.cpu(), e["gx"]) < 2e-4


def test_vgg_style_discriminator_trains_in_the_gan_step(cuda):
    """network_d: vggstylediscriminator through SRModel.optimize_parameters (bf16 AMP): G-side GAN term, D update, finite logs, both nets move."""
    from tests.test_sr_model_gpu import _opt
    from trainner_redux_amd.models import SRModel

    model = SRModel(_opt([dict(type="l1loss", loss_weight=1.0), dict(type="ganloss", gan_type="vanilla", loss_weight=0.1)], use_amp=True, blocks=1,
                         network_d=dict(type="vggstylediscriminator", num_feat=16, input_size=128), optim_d=dict(type="AdamW", lr=1e-4, weight_decay=0, betas=[0.9, 0.99])))
    g = torch.Generator().manual_seed(3)
    gt = torch.rand(4, 3, 128, 128, generator=g)
    lq = torch.nn.functional.interpolate(gt, scale_factor=0.25, mode="bicubic", antialias=True).clamp(0, 1)
    wg0, wd0 = model.net_g.flat_parameters()[0].clone(), model.net_d.flat_parameters()[0].clone()
    for it in (1, 2):
        model.feed_data({"lq": lq, "gt": gt})
        model.optimize_parameters(it, 0, True)
    log = model.get_current_log()
    assert {"l_g_l1", "l_g_gan", "l_d_real", "l_d_fake", "grad_norm_g", "grad_norm_d"} <= set(log) and all(v == v and abs(v) < 1e6 for v in log.values()), log
    assert (model.net_g.flat_parameters()[0] - wg0).abs().max() > 0 and (model.net_d.flat_parameters()[0] - wd0).abs().max() > 0
    assert int(model.net_d.state_dict()["bn0_1.num_batches_tracked"]) == 6          # three training-mode D forwards per step (sr_model.py:547,757-758)


@pytest.mark.parametrize("compute", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("kind", ["unetdiscriminatorsn"])
def test_discriminator_batch_is_the_union_of_its_halves(cuda, compute, kind):
    """configs[2]'s discriminator at the bench size (batch 8, 256 x 256) through the sample-independence property: logits and the input gradient of a batch
    equal its halves' bit for bit, parameter gradients are their sum.  Eval-mode spectral norm (stored u / v: training mode advances the power iteration
    per call).  (The VGG-style discriminator couples the samples through BatchNorm's batch statistics in training mode and has no eval-mode backward.)"""
    from trainner_redux_amd.archs import build_network

    torch.manual_seed(3)
    if kind == "unetdiscriminatorsn":
        net, size = build_network(dict(type=kind, num_in_ch=3, num_feat=64, skip_connection=True)), 256
    else:
        net, size = build_network(dict(type=kind, num_in_ch=3, num_feat=32, input_size=128)), 128
    net = net.to(cuda).eval()
    net.compute_dtype = compute
    g = torch.Generator().manual_seed(4)
    x = torch.rand(8, 3, size, size, generator=g).to(cuda)
    for _ in range(5):                       # a few power iterations first: the random initial u / v under-estimate sigma by orders of magnitude (fp16 overflows)
        net.forward_hip(x[:1], train=True)
    net._saved = None
    _, garr = net.flat_parameters()

    def run(parts):
        garr.zero_()
        outs, gins = [], []
        for sl in parts:
            y = net.forward_hip(x[sl], train=False)
            gy = torch.ones_like(y) / y[0].numel()
            gin = torch.zeros_like(x[sl])
            net.backward_hip(gy, param_grads=True, input_grad=gin, input_grad_accumulate=False)
            outs.append(y.clone()); gins.append(gin)
        torch.cuda.synchronize()
        return torch.cat(outs), torch.cat(gins), garr.double().clone()

    y_f, gi_f, g_f = run([slice(0, 8)])
    y_h, gi_h, g_h = run([slice(0, 4), slice(4, 8)])
    assert torch.equal(y_f, y_h) and torch.equal(gi_f, gi_h), ((y_f - y_h).abs().max().item(), (gi_f - gi_h).abs().max().item())
    rel = ((g_f - g_h).norm() / g_h.norm()).item()
    print(f"{kind} {compute}: logits and input gradient identical, parameter gradient rel-L2 {rel:.2e}")
    assert rel < 1e-5, rel
