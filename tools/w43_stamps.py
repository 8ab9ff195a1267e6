"""In-kernel phase stamps of the Winograd F(4,3) kernel (needs a build with -DTR_W43_STAMPS): waves 0 (loader) and 4 stamp kernel start, halo landed,
first V built, [before barrier, after barrier] per chunk, and the epilogue rounds.  Prints median phase durations in s_memtime ticks (10 ns)."""
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
for role in (0, 1):
    rows = []
    for b in range(1024):
        c = s[b, role]; c = c[c > 0]
        if len(c) >= 3 + 2 * min(nch, 28): rows.append(c)
    k = min(len(r) for r in rows)
    a = np.stack([r[:k] for r in rows]).astype(np.int64)
    d = np.median(np.diff(a, axis=1), axis=0).astype(int)
    tot = np.median(a[:, -1] - a[:, 0])
    print(f"role {role}: {len(rows)} workgroups, {k} stamps, median total {tot} ticks")
    print("  start->halo landed:", d[0], " fill V0:", d[1])
    loop = d[2:2 + 2 * min(nch, 28)]
    print("  chunk body (to barrier):", loop[0::2].tolist())
    print("  barrier wait:", loop[1::2].tolist())
    print("  rest (epilogue):", d[2 + 2 * min(nch, 28):].tolist())
# concurrency: how many workgroups overlap in time on average
allst = s[:, 0, :]; t0 = allst[allst > 0].min(); t1 = allst.max()
print("span of stamped workgroups:", t1 - t0, "ticks")
