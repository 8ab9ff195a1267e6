"""Winograd F(4x4,3x3) kernel: correctness against torch fp32 / fp64 on small and ragged shapes (forward with bias + ReLU, input-gradient form with the
ReLU-derivative mask), then timings on the VGG19 layer shapes next to the F(2x2,3x3) kernel.  Run on the GPU box."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops
dev = torch.device("cuda")


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


def check():
    bad = 0
    torch.manual_seed(0)
    for (n, h, w, cin, cout) in [(1, 16, 32, 64, 64), (2, 20, 36, 64, 128), (1, 8, 8, 128, 64), (1, 44, 68, 40, 64), (3, 4, 4, 256, 64), (1, 64, 96, 72, 192), (3, 176, 352, 64, 64), (3, 36, 100, 64, 128)]:
        x = torch.relu(torch.randn(n, cin, h, w, dtype=torch.float64))
        wt = torch.randn(cout, cin, 3, 3, dtype=torch.float64) * math.sqrt(2 / (cin * 9))
        b = torch.randn(cout, dtype=torch.float64)
        ref = torch.relu(F.conv2d(x, wt, b, padding=1))
        xd = x.float().permute(0, 2, 3, 1).contiguous().to(dev)
        out = torch.full((n, h, w, cout + 4), 3.0, device=dev)
        U = ops.pack_winograd43(wt.float().to(dev))
        pooled = torch.full((n, h // 2, w // 2, cout + 4), 5.0, device=dev)
        ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD43, inp=ops.view(xd), out=ops.view(out, 4, cout), wpacked=U, bias=b.float().to(dev), act=L.ACT_RELU,
                    out2=ops.view(pooled, 0, cout))
        got = out[..., 4:].permute(0, 3, 1, 2).cpu()
        e = rel(got, ref)
        ep = rel(pooled[..., :cout].permute(0, 3, 1, 2).cpu(), F.max_pool2d(ref, 2, 2))
        ok = e < 1e-5 and ep < 1e-5 and bool((out[..., :4] == 3.0).all()) and bool((pooled[..., cout:] == 5.0).all())
        # input-gradient form: transposed pack, mask
        gy = torch.randn(n, cout, h, w, dtype=torch.float64)
        m = torch.randn(n, cin, h, w, dtype=torch.float64)
        refg = F.conv_transpose2d(gy, wt, padding=1) * (m > 0)
        Ut = ops.pack_winograd43(wt.float().to(dev), transpose=True) if cin % 64 == 0 else None
        gin = torch.empty(n, h, w, cin, device=dev)
        gyd = gy.float().permute(0, 2, 3, 1).contiguous().to(dev)      # keep the operands alive: a View holds a raw pointer only
        md = m.float().permute(0, 2, 3, 1).contiguous().to(dev)
        if cin % 64 == 0:
            ops.conv3x3(n=n, h=h, w=w, cin=cout, cout=cin, compute=L.F32_WINOGRAD43, inp=ops.view(gyd), out=ops.view(gin), wpacked=Ut, mask=ops.view(md), mask_slope=0.0)
        eg = rel(gin.permute(0, 3, 1, 2).cpu(), refg) if cin % 64 == 0 else 0.0
        print(f"n{n} {h}x{w} {cin}->{cout}: fwd rel-L2 {e:.2e}  dgrad rel-L2 {eg:.2e}  {'ok' if ok and eg < 1e-5 else 'MISMATCH'}")
        bad += not (ok and eg < 1e-5)
    print("check done, mismatches:", bad)
    return bad


def bench():
    shapes = [(16, 1024, 1024, 64, 64), (16, 512, 512, 64, 128), (16, 512, 512, 128, 128), (16, 256, 256, 128, 256), (16, 256, 256, 256, 256),
              (16, 128, 128, 256, 512), (16, 128, 128, 512, 512), (16, 64, 64, 512, 512)]
    for (n, h, w, cin, cout) in shapes:
        x = torch.randn(n, h, w, cin, device=dev)
        wt = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
        out = torch.empty(n, h, w, cout, device=dev)
        U2, U4 = ops.pack_winograd(wt), ops.pack_winograd43(wt)
        f2 = lambda: ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD, inp=ops.view(x), out=ops.view(out), wpacked=U2, act=L.ACT_RELU)
        f4 = lambda: ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD43, inp=ops.view(x), out=ops.view(out), wpacked=U4, act=L.ACT_RELU)
        f4m = lambda: ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD43, inp=ops.view(x), out=ops.view(out), wpacked=U4, mask=ops.view(mk), mask_slope=0.0)
        mk = torch.randn(n, h, w, cout, device=dev)
        res = []
        for f in (f2, f4, f4m):
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): f()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 5)
        flops = 2.0 * n * h * w * 9 * cin * cout
        print(f"n{n} {h}x{w} {cin}->{cout}: F(2,3) {res[0]:7.3f} ms ({flops / res[0] / 1e9:6.1f} TF/s eff)  F(4,3) {res[1]:7.3f} ms ({flops / res[1] / 1e9:6.1f} TF/s eff)  x{res[0] / res[1]:.2f}  masked (dgrad form) {res[2]:7.3f} ms", flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("check", "all"):
        if check() and what == "all": sys.exit(1)
    if what in ("bench", "all"): bench()
