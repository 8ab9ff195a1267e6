"""VGG19 layer shapes (batch 16 at 1024^2 input): direct fp32 MFMA conv vs the fused Winograd kernel (GPU box)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops
dev = torch.device("cuda")
shapes = [(16, 1024, 1024, 64, 64), (16, 512, 512, 64, 128), (16, 512, 512, 128, 128), (16, 256, 256, 128, 256), (16, 256, 256, 256, 256),
          (16, 128, 128, 256, 512), (16, 128, 128, 512, 512), (16, 64, 64, 512, 512)]
if "--small" in sys.argv: shapes = [(4, s[1], s[2], s[3], s[4]) for s in shapes]
for (n, h, w, cin, cout) in shapes:
    x = torch.randn(n, h, w, cin, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
    out = torch.empty(n, h, w, cout, device=dev)
    buf = torch.zeros(ops.packed_bytes(cin, cout, L.F32), dtype=torch.uint8, device=dev)
    plan = ops.PackPlan(dev); plan.add(wt, buf, cin, cout, L.F32); plan.run()
    U = ops.pack_winograd(wt)
    fd = lambda: ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32, inp=ops.view(x), out=ops.view(out), wpacked=buf, act=L.ACT_RELU)
    fw = lambda: ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD, inp=ops.view(x), out=ops.view(out), wpacked=U, act=L.ACT_RELU)
    res = []
    for f in (fd, fw):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    flops = 2.0 * n * h * w * 9 * cin * cout
    print(f"n{n} {h}x{w} {cin}->{cout}: direct {res[0]:7.3f} ms ({flops / res[0] / 1e9:6.1f} TF/s)  winograd {res[1]:7.3f} ms ({flops / res[1] / 1e9:6.1f} TF/s eff)  x{res[0] / res[1]:.2f}")
