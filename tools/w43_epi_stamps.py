"""Epilogue detail of the Winograd F(4,3) kernel (build with -DTR_W43_STAMPS=2): raw stamp differences of the first tile of a few workgroups."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops
dev = torch.device("cuda")
n, h, w, cin, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 1024, 1024, 64, 64))]
x = torch.randn(n, h, w, cin, device=dev)
wt = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
out = torch.empty(n, h, w, cout, device=dev)
U = ops.pack_winograd43(wt)
f = lambda: ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD43, inp=ops.view(x), out=ops.view(out), wpacked=U, act=L.ACT_RELU)
for _ in range(3): f()
st = torch.zeros(1024 * 2 * 64, dtype=torch.int64, device=dev)
os.environ["TRAINNER_W43_STAMPS"] = hex(st.data_ptr())
f(); torch.cuda.synchronize()
s = st.cpu().numpy().reshape(1024, 2, 64)
nch = cin // 8
k0 = 1 + 2 + 2 * nch + 1          # index of the first epilogue-detail stamp of tile 0 (start, 2 prologue, 2 per chunk, K-loop exit)
for role in (0, 1):
    rows = np.stack([s[b, role] for b in range(256) if (s[b, role] > 0).sum() == 64]).astype(np.int64)
    d = np.median(np.diff(rows, axis=1), axis=0).astype(int)
    print(f"role {role}: K-loop exit -> after fold+DMA {d[k0 - 1]}; rounds [write->pre-barrier, barrier, finishing, barrier]:",
          [d[k0 + 4 * r: k0 + 4 * r + 4].tolist() for r in range(4)], " tail", d[k0 + 16: k0 + 18].tolist())
