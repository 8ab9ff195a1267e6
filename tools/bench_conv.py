"""Micro-benchmark of tr_conv3x3 / tr_wgrad3x3 on the shapes of the headline config (A/B tool; run on the GPU box)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops

dev = torch.device("cuda")
shapes = [(16, 256, 256, 64, 32), (16, 256, 256, 96, 32), (16, 256, 256, 128, 32), (16, 256, 256, 160, 32), (16, 256, 256, 192, 64),
          (16, 1024, 1024, 64, 64), (16, 512, 512, 64, 64)]
if os.environ.get("N"):      # sub-batch A/B: the same layers at another batch size (does the working set fit the 256 MB Infinity Cache?)
    shapes = [(int(os.environ["N"]),) + sh[1:] for sh in shapes]
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
cds = [("bf16", L.BF16, torch.bfloat16)] if "f32" not in sys.argv else [("f32", L.F32, torch.float32)]
for cdn, cd, dt in cds:
    for (n, h, w, cin, cout) in shapes:
        if cd == L.F32 and h > 512: n = 4
        x = torch.randn(n, h, w, cin, device=dev).to(dt)
        wt = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
        flops = 2.0 * n * h * w * 9 * cin * cout
        if which == "conv":
            buf = torch.zeros(ops.packed_bytes(cin, cout, cd), dtype=torch.uint8, device=dev)
            plan = ops.PackPlan(dev); plan.add(wt, buf, cin, cout, cd); plan.run()
            out = torch.empty(n, h, w, cout, device=dev, dtype=dt)
            kw = dict(act=L.ACT_LRELU)
            if os.environ.get("SIDE") == "mask":        # the dgrad form: LeakyReLU-derivative mask read per output element
                mk = torch.randn(n, h, w, cout, device=dev).to(dt)
                kw = dict(mask=ops.view(mk), mask_slope=0.2)
            elif os.environ.get("SIDE") == "res":       # conv5 / RRDB tail: two residual inputs
                r1, r2 = torch.randn(n, h, w, cout, device=dev).to(dt), torch.randn(n, h, w, cout, device=dev).to(dt)
                kw = dict(res1=ops.view(r1), scale1=0.2, res2=ops.view(r2), scale2=0.2)
            f = lambda: ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=cd, inp=ops.view(x), out=ops.view(out), wpacked=buf, **kw)
        else:
            gy = torch.randn(n, h, w, cout, device=dev).to(dt)
            ws = torch.empty(ops.wgrad_workspace_bytes(n, h, w, cin, cout) // 4, dtype=torch.float32, device=dev)
            dw = torch.empty(cout, cin, 3, 3, device=dev); db = torch.empty(cout, device=dev)
            f = lambda: ops.wgrad3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=cd, inp=ops.view(x), gy=ops.view(gy), dw=dw, db=db, workspace=ws)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        e0.record()
        for _ in range(iters): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        byts = n * h * w * (cin + cout) * x.element_size()
        print(f"{which} {cdn} n{n} {h}x{w} {cin}->{cout}: {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s  {byts / ms / 1e6:7.1f} GB/s(min traffic)")
