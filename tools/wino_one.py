"""One VGG shape through the Winograd kernel, a handful of launches (for rocprofv3 --pmc passes)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops
dev = torch.device("cuda")
n, h, w, cin, cout = 16, 128, 128, 512, 512
x = torch.randn(n, h, w, cin, device=dev)
wt = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
out = torch.empty(n, h, w, cout, device=dev)
U = ops.pack_winograd(wt)
for _ in range(4):
    ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD, inp=ops.view(x), out=ops.view(out), wpacked=U, act=L.ACT_RELU)
torch.cuda.synchronize()
