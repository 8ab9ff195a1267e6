"""Per-layer L2-miss traffic of the Winograd F(4x4,3x3) kernel on the VGG19 layer shapes of the headline workload (batch 16, 1024^2 input): run under
   rocprofv3 --kernel-trace --pmc FETCH_SIZE (then WRITE_SIZE) -- python3 tools/w43_traffic.py
and list the dispatches with `python tools/prof_summary.py seq <dir> FETCH_SIZE winograd43`.  Each shape is launched REPS times, in the order printed."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops

dev = torch.device("cuda")
REPS = 2
shapes = [(16, 1024, 1024, 64, 64), (16, 512, 512, 64, 128), (16, 512, 512, 128, 128), (16, 256, 256, 128, 256), (16, 256, 256, 256, 256),
          (16, 128, 128, 256, 512), (16, 128, 128, 512, 512), (16, 64, 64, 512, 512)]
for (n, h, w, cin, cout) in shapes:
    x = torch.randn(n, h, w, cin, device=dev)
    wt = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
    out = torch.empty(n, h, w, cout, device=dev)
    U = ops.pack_winograd43(wt)
    torch.cuda.synchronize()
    for _ in range(REPS):
        ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD43, inp=ops.view(x), out=ops.view(out), wpacked=U, act=L.ACT_RELU)
    torch.cuda.synchronize()
    px = n * h * w
    print(f"n{n} {h}x{w} {cin}->{cout}: algorithmic in {px * cin * 4 / 1e6:.1f} MB  out {px * cout * 4 / 1e6:.1f} MB  packed U {36 * cin * cout * 4 / 1e6:.1f} MB  "
          f"output groups {cout // 64}", flush=True)
    del x, out, U
