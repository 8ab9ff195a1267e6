"""Round-3 diagnostics (one GPU call): (1) ring D=1 pipelined instance on forward shapes, (2) ring vs register kernel mismatch statistics,
(3) RRDB gradient distances ring / reg / oracle."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.nn import functional as F
import trainner_redux_amd  # noqa
from trainner_redux_amd import _lib as L, ops

dev = torch.device("cuda")
def nchw(t): return t.permute(0, 3, 1, 2).contiguous()
def rel(a, b): return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
def pack(w, cin, cout, compute, transpose=False):
    buf = torch.zeros(ops.packed_bytes(cin, cout, compute), dtype=torch.uint8, device=dev)
    plan = ops.PackPlan(dev); plan.add(w, buf, cin, cout, compute, transpose=transpose); plan.run(); return buf

print("=== (1) D=1 pipelined forward")
for (n, h, w) in [(3, 16, 32), (1, 17, 33), (2, 40, 72), (1, 8, 32)]:
    for cin in (64, 96, 128):
        for cfg in (31, 32, 33):
            g = torch.Generator().manual_seed(1)
            x = torch.randn(n, h, w, cin, generator=g).bfloat16().float()
            wt = (torch.randn(32, cin, 3, 3, generator=g) / math.sqrt(cin * 9)).bfloat16().float()
            b = torch.randn(32, generator=g)
            ref = F.leaky_relu(F.conv2d(nchw(x), wt, b, padding=1), 0.2).bfloat16().float()
            xd = x.to(dev, torch.bfloat16); wp = pack(wt.to(dev), cin, 32, L.BF16); bd = b.to(dev)
            res = []
            for rep in range(2):
                out = torch.zeros(n, h, w, 32, dtype=torch.bfloat16, device=dev)
                with ops.tuned(conv_impl=2, ring_cfg=cfg):
                    ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=32, compute=L.BF16, inp=ops.view(xd), out=ops.view(out), wpacked=wp, bias=bd, act=L.ACT_LRELU)
                res.append(nchw(out.float().cpu()))
            inst = ops.tune_get("last_conv_instance")
            bad = (res[0] != ref)
            rows = bad.any(dim=(0, 1, 3)).nonzero().flatten().tolist()
            chans = bad.any(dim=(0, 2, 3)).nonzero().flatten().tolist()
            print(f"geom {(n,h,w)} cin {cin} cfg {cfg} inst {inst}: same {1 - bad.float().mean().item():.4f} repeatable {torch.equal(res[0], res[1])} bad rows {rows[:20]} bad chans {len(chans)}")

print("=== (2) ring vs reg")
for cn, dt, comp in (("bf16", torch.bfloat16, L.BF16), ("f16", torch.float16, L.F16)):
    n, h, w, cin, cout = 2, 256, 288, 96, 32
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, h, w, cin, generator=g).to(dev, dt)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / 30).to(dev)
    b = torch.randn(cout, generator=g).to(dev)
    wp = pack(wt, cin, cout, comp)
    outs = {}
    for impl in (0, 1):
        for o32 in (False, True):
            out = torch.empty(n, h, w, cout, dtype=torch.float32 if o32 else dt, device=dev)
            with ops.tuned(conv_impl=impl):
                ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=comp, inp=ops.view(x), out=ops.view(out), wpacked=wp, bias=b, act=L.ACT_LRELU)
            outs[(impl, o32)] = out.float().cpu()
    ref = F.leaky_relu(F.conv2d(nchw(x.float().cpu()), wt.cpu().to(dt).float(), b.cpu(), padding=1), 0.2)
    print(cn, "16-bit out: ring==reg", (outs[(0, False)] == outs[(1, False)]).float().mean().item(), " fp32 out: ring==reg", (outs[(0, True)] == outs[(1, True)]).float().mean().item(),
          "rel(ring32, ref)", rel(nchw(outs[(0, True)]), ref), "rel(reg32, ref)", rel(nchw(outs[(1, True)]), ref), "rel(ring32, reg32)", rel(outs[(0, True)], outs[(1, True)]))

print("=== (3) RRDB gradients")
from collections import OrderedDict
from oracle import esrgan as OE
from oracle.precision import BF16_OPS, F16_OPS
from trainner_redux_amd.archs import build_network
for cn, dt, policy in (("bf16", torch.bfloat16, BF16_OPS), ("f16", torch.float16, F16_OPS)):
    kw = dict(scale=1, use_pixel_unshuffle=False, num_blocks=1)
    hp = OE.esrgan_hparams(**kw); sd = OE.init_state_dict(hp, seed=4)
    g = torch.Generator().manual_seed(6)
    x = torch.rand(1, 3, 360, 376, generator=g)
    gout = torch.randn(1, 3, 360, 376, generator=g)
    params = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd.items())
    ref = OE.forward(params, x, hp, policy)
    ref_g = OrderedDict(zip(params.keys(), torch.autograd.grad(ref, list(params.values()), gout)))
    # the same with the L1-style cotangent sign/numel
    got = {}
    for impl in (0, 1):
        net = build_network({"type": "esrgan", **kw}); net.load_state_dict(sd, strict=True); net = net.to(dev); net.compute_dtype = dt; net.train()
        with ops.tuned(conv_impl=impl, wgrad_impl=impl):
            out = net.forward_hip(x.to(dev), train=True)
            net.backward_hip(gout.to(dev))
        got[impl] = (out.cpu(), {k: p.grad.cpu().clone() for k, p in net.state_dict(keep_vars=True).items()})
    print(cn, "fwd: ring-oracle", rel(got[0][0], ref.detach()), "reg-oracle", rel(got[1][0], ref.detach()), "ring-reg", rel(got[0][0], got[1][0]))
    for k in ref_g:
        a, b_, r = got[0][1][k], got[1][1][k], ref_g[k]
        print(f"  {k:40s} ring-oracle {rel(a, r):.2e} reg-oracle {rel(b_, r):.2e} ring-reg {rel(a, b_):.2e} |ref| {r.norm().item():.3e} nan {int(torch.isnan(a).sum())}")
