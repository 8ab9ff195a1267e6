"""A/B of the LDS-DMA ring kernel (conv_ring.hip) against the register-staged kernel: bit-exact comparison on slices of dense-block shaped
buffers in every epilogue form of the hot path, ragged sizes included, then timings per shape.  Run on the GPU box."""
import os, sys, math, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops

dev = torch.device("cuda")
torch.manual_seed(0)


def run(impl, cfg, n, h, w, cin, cout, form, dt=torch.bfloat16, cd=L.BF16, up=False, bufs=None):
    ops.tune_set("conv_impl", {"reg": 1, "ring": 2}[impl])        # tr_tune_set (round 3: launches no longer read the environment)
    ops.tune_set("ring_cfg", -1 if cfg is None else cfg)
    x, wbuf, bias, mk, r1, r2 = bufs
    cs = x.shape[-1]
    out_dt = torch.float32 if form in ("res1", "res2") else dt
    ooff = 0 if os.environ.get("OS") else 16                                       # OS: channel stride of the output buffer (slice at 0; default cout + 32, slice at 16)
    out = torch.zeros(n, h, w, int(os.environ.get("OS", cout + 32)), device=dev, dtype=out_dt)
    out2 = torch.zeros(n, h, w, cout, device=dev, dtype=dt)
    kw = {}
    if form == "lrelu": kw = dict(act=L.ACT_LRELU, slope=0.2)
    elif form == "mask": kw = dict(mask=ops.view(mk, 32, cout), mask_slope=0.2)
    elif form == "res1": kw = dict(res1=ops.view(r1), scale1=0.2, out2=ops.view(out2), out2_scale=0.5)
    elif form == "res2": kw = dict(res1=ops.view(r1), scale1=0.2, res2=ops.view(r2), scale2=0.2, res1_scale=0.7, out2=ops.view(out2), out2_scale=0.04)
    f = lambda: ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=cd, inp=ops.view(x, 0, cin), out=ops.view(out, ooff, cout), wpacked=wbuf, bias=bias,
                            in_mode=L.IN_NEAREST2X if up else L.IN_DIRECT, **kw)
    f()
    torch.cuda.synchronize()
    return out, out2, f


def mk_bufs(n, h, w, cin, cout, dt, cd, up=False):
    hi, wi = (h // 2, w // 2) if up else (h, w)
    x = torch.randn(n, hi, wi, int(os.environ.get("XS", cin + 32)), device=dev).to(dt)      # XS: channel stride of the input buffer (default cin + 32)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
    wbuf = torch.zeros(ops.packed_bytes(cin, cout, cd), dtype=torch.uint8, device=dev)
    plan = ops.PackPlan(dev); plan.add(wt, wbuf, cin, cout, cd); plan.run()
    bias = torch.randn(cout, device=dev)
    mk = torch.randn(n, h, w, cout + 64, device=dev).to(dt)
    r1 = torch.randn(n, h, w, cout, device=dev)
    r2 = torch.randn(n, h, w, cout, device=dev)
    return x, wbuf, bias, mk, r1, r2


def check():
    bad = 0
    cases = [(2, 40, 70, 64, 32), (1, 16, 32, 96, 32), (3, 33, 31, 128, 32), (1, 50, 100, 160, 32), (2, 24, 64, 192, 64), (1, 17, 130, 64, 64), (1, 64, 64, 64, 48)]
    for (n, h, w, cin, cout) in cases:
        for dt, cd in ((torch.bfloat16, L.BF16), (torch.float16, L.F16)):
            for up in (False, True):
                if up and (h % 2 or w % 2): continue
                bufs = mk_bufs(n, h, w, cin, cout, dt, cd, up)
                for form in ("plain", "lrelu", "mask", "res1", "res2"):
                    ref, ref2, _ = run("reg", None, n, h, w, cin, cout, form, dt, cd, up, bufs)
                    for cfg in ([33, 0] if cout <= 32 else [0]):
                        o, o2, _ = run("ring", cfg, n, h, w, cin, cout, form, dt, cd, up, bufs)
                        # the ring kernel seeds its accumulators with the bias (the register-staged kernel adds it after the K loop) and the compiler
                        # contracts the residual FMAs differently: fp32 results agree to a few ulps, 16-bit results to one ulp on a small fraction
                        df = (o.float() - ref.float()).abs()
                        tol = 4e-6 * ref.float().abs().clamp_min(1.0) if o.dtype == torch.float32 else 0.0081 * ref.float().abs().clamp_min(1e-3)
                        frac = (df > 0).float().mean().item()
                        d2 = (o2.float() - ref2.float()).abs()
                        ok = bool((df <= tol).all()) and bool((d2 <= 0.0081 * ref2.float().abs().clamp_min(1e-3)).all()) and (o.dtype == torch.float32 or frac < 0.02)
                        if not ok:
                            bad += 1
                            d = (o.float() - ref.float()).abs().max().item()
                            print(f"MISMATCH n{n} {h}x{w} {cin}->{cout} {dt} up{int(up)} {form} cfg{cfg}: max abs diff {d:.3e}, out2 equal {torch.equal(o2, ref2)}")
    print("check done, mismatches:", bad)
    return bad


def bench():
    shapes = [(16, 256, 256, 64, 32), (16, 256, 256, 96, 32), (16, 256, 256, 128, 32), (16, 256, 256, 160, 32), (16, 256, 256, 192, 64),
              (16, 1024, 1024, 64, 64), (16, 512, 512, 64, 64)]
    if os.environ.get("SHAPES"):  # e.g. SHAPES=16x256x256x32x32,16x256x256x64x32
        shapes = [tuple(int(v) for v in sh.split("x")) for sh in os.environ["SHAPES"].split(",")]
    if os.environ.get("N"):      # sub-batch A/B: the same layers at another batch size (does the working set fit the 256 MB Infinity Cache?)
        shapes = [(int(os.environ["N"]),) + sh[1:] for sh in shapes]
    forms = sys.argv[2].split(",") if len(sys.argv) > 2 else ["lrelu", "mask", "res1"]
    for (n, h, w, cin, cout) in shapes:
        bufs = mk_bufs(n, h, w, cin, cout, torch.bfloat16, L.BF16)
        flops = 2.0 * n * h * w * 9 * cin * cout
        for form in forms:
            if form == "res1" and cout == 32: continue
            line = f"n{n} {h}x{w} {cin:3d}->{cout} {form:5s}:"
            for impl, cfg in [("reg", None)] + [("ring", c) for c in ([33, 0] if cout <= 32 else [0])]:
                _, _, f = run(impl, cfg, n, h, w, cin, cout, form, bufs=bufs)
                for _ in range(2): f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                line += f"  {impl}{'' if cfg is None else cfg} {ms:6.3f}ms {flops / ms / 1e9:6.0f}TF"
            print(line, flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what in ("check", "all"):
        if check() and what == "all": sys.exit(1)
    if what in ("bench", "all"): bench()
