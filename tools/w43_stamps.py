"""In-kernel phase stamps of the Winograd F(4,3) kernel (needs a build with -DTR_W43_STAMPS): waves 0 and 4 stamp, per tile of the persistent
walk, [halo of chunk 0 landed, first V built, (before barrier, after barrier) per chunk, K loop left, epilogue done].  Prints the median
differences between consecutive stamps over the workgroups, in s_memtime ticks."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops
dev = torch.device("cuda")
n, h, w, cin, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 1024, 1024, 64, 64))]
mask = len(sys.argv) > 6 and sys.argv[6] == "mask"
x = torch.randn(n, h, w, cin, device=dev)
wt = torch.randn(cout, cin, 3, 3, device=dev) / math.sqrt(cin * 9)
out = torch.empty(n, h, w, cout, device=dev)
m = torch.randn(n, h, w, cout, device=dev) if mask else None
U = ops.pack_winograd43(wt)
kw = dict(mask=ops.view(m), mask_slope=0.0) if mask else dict(act=L.ACT_RELU)
f = lambda: ops.conv3x3(n=n, h=h, w=w, cin=cin, cout=cout, compute=L.F32_WINOGRAD43, inp=ops.view(x), out=ops.view(out), wpacked=U, **kw)
for _ in range(3): f()
st = torch.zeros(1024 * 2 * 64, dtype=torch.int64, device=dev)
os.environ["TRAINNER_W43_STAMPS"] = hex(st.data_ptr())
f(); torch.cuda.synchronize()
os.environ.pop("TRAINNER_W43_STAMPS")
s = st.cpu().numpy().reshape(1024, 2, 64)
nch = cin // 8
per_tile = 2 + 2 * nch + 2
for role in (0, 1):
    rows = [s[b, role] for b in range(1024) if (s[b, role] > 0).sum() == 64]
    a = np.stack(rows).astype(np.int64)
    d = np.median(np.diff(a, axis=1), axis=0).astype(int)
    print(f"role {role}: {len(rows)} workgroups with 64 stamps; stamps per tile {per_tile}; start -> first stamp of tile 0: {d[0]}")
    body = d[1:]
    for t in range(0, len(body), per_tile):
        seg = body[t:t + per_tile]
        if len(seg) < per_tile: break
        print(f"  tile {t // per_tile}: fill V0 {seg[0]}  chunks (body/barrier) {[(int(seg[1 + 2 * c]), int(seg[2 + 2 * c])) for c in range(min(nch, 12))]}"
              f"  K-loop exit {seg[1 + 2 * nch]}  epilogue {seg[2 + 2 * nch]}  -> next tile's first stamp {seg[3 + 2 * nch] if len(seg) > 3 + 2 * nch else '-'}")
