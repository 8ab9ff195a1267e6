"""Micro-benchmark of the token-grid (ksize 1) path of tr_conv3x3 / tr_wgrad3x3 on SwinIR-M's Linear shapes (run on the GPU box).
   python tools/bench_linear.py [conv|wgrad] [f32] ;  TRAINNER_LINEAR=conv selects the 3x3-style tiling for A/B."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops

dev = torch.device("cuda")
B, hw = int(os.environ.get("B", "32")), 64 * 64
shapes = [(192, 540, "f32out"), (192, 540, "cdout"), (192, 180, "f32out+res"), (192, 360, "gelu"), (384, 180, "f32out+res"), (544, 180, "f32out"), (192, 360, "cdout"),
          (384, 180, "cdout")]
which = sys.argv[1] if len(sys.argv) > 1 else "conv"
ops.tune_set("linear_impl", 1 if os.environ.get("LINEAR_IMPL") == "reg" else 0)
ops.tune_set("linear_epilogue", 1 if os.environ.get("LINEAR_EPILOGUE") == "generic" else 0)      # A/B: LDS-DMA staged (default) vs register-staged token GEMM
cdn, cd, dt = ("f32", L.F32, torch.float32) if "f32" in sys.argv else ("bf16", L.BF16, torch.bfloat16)
n, h, w = B, hw // 32, 32
for (cin, cout, mode) in shapes:
    cop = (cout + 31) // 32 * 32
    x = torch.randn(n, h, w, cin, device=dev).to(dt)
    wt = torch.randn(cout, cin, device=dev) / math.sqrt(cin)
    flops = 2.0 * n * h * w * cin * cout
    if which == "conv":
        buf = torch.zeros(ops.packed_bytes(cin, cout, cd, 1), dtype=torch.uint8, device=dev)
        plan = ops.PackPlan(dev); plan.add(wt, buf, cin, cout, cd); plan.run()
        odt = dt if "cdout" in mode or mode == "gelu" else torch.float32
        out = torch.zeros(n, h, w, cop, device=dev, dtype=odt)
        kw = {}
        if "res" in mode:
            res = torch.randn(n, h, w, cop, device=dev)
            kw["res1"] = ops.view(res)
        if mode == "gelu":
            pre = torch.zeros(n, h, w, cop, device=dev, dtype=dt)
            kw.update(act=L.ACT_GELU, out2=ops.view(pre))
        f = lambda: ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=cd, inp=ops.view(x), out=ops.view(out), wpacked=buf, ksize=1, **kw)
        byts = n * h * w * (cin * x.element_size() + cop * out.element_size() * (2 if mode == "gelu" else 1) + (cop * 4 if "res" in mode else 0))
    else:
        gy = torch.randn(n, h, w, cop, device=dev).to(dt)
        ws = torch.empty(ops.wgrad_workspace_bytes(n, h, w, cin, cout, 1) // 4 + 16, dtype=torch.float32, device=dev)
        dw = torch.empty(cout, cin, device=dev); db = torch.empty(cout, device=dev)
        f = lambda: ops.wgrad3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=cd, inp=ops.view(x), gy=ops.view(gy), dw=dw, db=db, workspace=ws, ksize=1)
        byts = n * h * w * (cin + cop) * x.element_size()
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"{which} {cdn} tokens {n * h * w} {cin}->{cout} {mode:12s}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s  {byts / ms / 1e6:7.1f} GB/s (min traffic)")
