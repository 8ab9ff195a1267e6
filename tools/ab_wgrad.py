"""A/B of the wgrad ring kernel (wgrad_ring.hip) against the register-staged wgrad kernel: dW / db agreement on slices of dense-block shaped buffers
(ragged sizes, nearest-2x input), then timings per shape.  Run on the GPU box."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops

dev = torch.device("cuda")
torch.manual_seed(0)


def mk(n, h, w, cin, cout, dt, up=False):
    hi, wi = (h // 2, w // 2) if up else (h, w)
    x = torch.randn(n, hi, wi, cin + 32, device=dev).to(dt)
    gy = (torch.randn(n, h, w, cout + 32, device=dev) / 16).to(dt)
    ws = torch.empty(ops.wgrad_workspace_bytes(n, h, w, cin, cout) // 4 + 16, dtype=torch.float32, device=dev)
    return x, gy, ws


def run(impl, n, h, w, cin, cout, cd, bufs, up=False, accumulate=False):
    ops.tune_set("wgrad_impl", {"reg": 1, "ring": 2}.get(impl, 0))
    x, gy, ws = bufs
    dw = torch.full((cout, cin, 3, 3), 0.5, device=dev)
    db = torch.full((cout,), 0.25, device=dev)
    f = lambda: ops.wgrad3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=cd, inp=ops.view(x, 16, cin), gy=ops.view(gy, 8, cout), dw=dw, db=db, workspace=ws,
                             in_mode=L.IN_NEAREST2X if up else L.IN_DIRECT, accumulate=accumulate)
    f()
    torch.cuda.synchronize()
    return dw, db, f


def check():
    bad = 0
    cases = [(2, 40, 70, 64, 32), (1, 16, 32, 96, 32), (3, 33, 31, 128, 32), (1, 50, 100, 160, 32), (2, 24, 64, 192, 64), (1, 17, 130, 64, 64), (1, 64, 64, 64, 48), (1, 32, 64, 72, 24)]
    for (n, h, w, cin, cout) in cases:
        for dt, cd in ((torch.bfloat16, L.BF16), (torch.float16, L.F16)):
            for up in (False, True):
                if up and (h % 2 or w % 2): continue
                bufs = mk(n, h, w, cin, cout, dt, up)
                for acc in (False, True):
                    rw, rb, _ = run("reg", n, h, w, cin, cout, cd, bufs, up, acc)
                    ow, ob, _ = run("ring", n, h, w, cin, cout, cd, bufs, up, acc)
                    # same products, another summation order (fp32): a few 1e-6 of the gradient's scale
                    scale = rw.abs().max().item()
                    ew = (ow - rw).abs().max().item() / scale
                    eb = (ob - rb).abs().max().item() / max(rb.abs().max().item(), 1e-6)
                    if not (ew < 2e-5 and eb < 2e-5):
                        bad += 1
                        print(f"MISMATCH n{n} {h}x{w} {cin}->{cout} {dt} up{int(up)} acc{int(acc)}: dw {ew:.2e} db {eb:.2e}")
    print("check done, mismatches:", bad)
    return bad


def bench():
    shapes = [(16, 256, 256, 64, 32), (16, 256, 256, 96, 32), (16, 256, 256, 128, 32), (16, 256, 256, 160, 32), (16, 256, 256, 192, 64),
              (16, 1024, 1024, 64, 64), (16, 512, 512, 64, 64)]
    if os.environ.get("SHAPES"):  # e.g. SHAPES=8x64x64x64x32,8x64x64x192x64
        shapes = [tuple(int(v) for v in sh.split("x")) for sh in os.environ["SHAPES"].split(",")]
    for (n, h, w, cin, cout) in shapes:
        bufs = mk(n, h, w, cin, cout, torch.bfloat16)
        flops = 2.0 * n * h * w * 9 * cin * cout
        line = f"n{n} {h}x{w} {cin:3d}->{cout}:"
        for impl in ("reg", "ring"):
            _, _, f = run(impl, n, h, w, cin, cout, L.BF16, bufs)
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            line += f"  {impl} {ms:6.3f}ms {flops / ms / 1e9:6.0f}TF"
        print(line, flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "check"
    if what in ("check", "all"):
        if check() and what == "all": sys.exit(1)
    if what in ("bench", "all"): bench()


def multi():
    """The five weight gradients of a dense block (channel prefixes of one 192-channel x buffer, slices of one 192-channel gradient buffer):
    one fused launch (tr_wgrad3x3_multi) against five separate ones -- agreement, then time per block."""
    nf, gc = 64, 32
    geoms = [(2, 40, 72), (16, 256, 256)]
    if os.environ.get("GEOMS"):   # e.g. GEOMS=8x64x64,2x64x64
        geoms = [tuple(int(v) for v in sh.split("x")) for sh in os.environ["GEOMS"].split(",")]
    for (n, h, w) in geoms:
        xb = torch.randn(n, h, w, nf + 4 * gc, device=dev).bfloat16()
        g = (torch.randn(n, h, w, nf + 4 * gc, device=dev) / 16).bfloat16()
        shapes = [(nf + gc * (c - 1), gc if c < 5 else nf, 0 if c == 5 else nf + gc * (4 - c)) for c in range(1, 6)]
        need = max(ops.wgrad_workspace_bytes(n, h, w, ci, co) for ci, co, _ in shapes)
        wss = [torch.empty(need // 4 + 16, dtype=torch.float32, device=dev) for _ in range(5)]

        def run(fused, impl=2):
            ops.tune_set("wgrad_impl", impl)
            ops.tune_set("wgrad_multi", 1 if fused else 0)
            dws = [torch.full((co, ci, 3, 3), 0.5, device=dev) for ci, co, _ in shapes]
            dbs = [torch.full((co,), 0.25, device=dev) for ci, co, _ in shapes]
            convs = [dict(cin=ci, cout=co, inp=ops.view(xb, 0, ci), gy=ops.view(g, go, co), dw=dws[k], db=dbs[k], workspace=wss[k]) for k, (ci, co, go) in enumerate(shapes)]
            f = lambda: ops.wgrad3x3_multi(convs, n=n, h=h, w=w, compute=L.BF16, accumulate=True)
            f(); torch.cuda.synchronize()
            return dws, dbs, f
        d0, b0, f0 = run(False)
        d1, b1, f1 = run(True)
        bad = 0
        for k in range(5):
            ew = ((d1[k] - d0[k]).double().norm() / (d0[k] - 0.5).double().norm()).item()
            eb = ((b1[k] - b0[k]).double().norm() / (b0[k] - 0.25).double().norm()).item()
            bad += not (ew < 2e-6 and eb < 2e-6)
            print(f"n{n} {h}x{w} conv{k + 1} {shapes[k][0]}->{shapes[k][1]}: dw rel {ew:.2e} db rel {eb:.2e}")
        print("multi check mismatches:", bad)
        if n * h * w < 256 * 512 and not os.environ.get("GEOMS"): continue
        for name, fused, impl in (("register-staged, separate", False, 1), ("ring, separate", False, 2), ("ring, fused", True, 2)):
            _, _, f = run(fused, impl)
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            print(f"  {name}: {e0.elapsed_time(e1) / 10:.3f} ms per dense block (5 wgrads + reductions)")


if len(sys.argv) > 1 and sys.argv[1] == "multi": multi()
