import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_ring as AB
n, h, w, cin, cout = 2, 24, 64, 192, 64
dt, cd = torch.bfloat16, L.BF16
bufs = AB.mk_bufs(n, h, w, cin, cout, dt, cd, False)
for form in ("res1",):
    ref, ref2, _ = AB.run("reg", None, n, h, w, cin, cout, form, dt, cd, False, bufs)
    o, o2, _ = AB.run("ring", int(sys.argv[1]) if len(sys.argv) > 1 else 1, n, h, w, cin, cout, form, dt, cd, False, bufs)
    d = (o.float() - ref.float()).abs()
    bad = d > 1e-3
    print("bad frac", bad.float().mean().item())
    print("by channel (of 96):", bad.float().mean(dim=(0, 1, 2)).cpu().numpy().round(2))
    print("by row:", bad.float().mean(dim=(0, 2, 3)).cpu().numpy().round(2))
    print("by col:", bad.float().mean(dim=(0, 1, 3)).cpu().numpy().round(2))
    d2 = (o2.float() - ref2.float()).abs() > 1e-2
    print("out2 by channel:", d2.float().mean(dim=(0, 1, 2)).cpu().numpy().round(2))
    idx = bad.nonzero()
    print("n bad", idx.shape[0])
    for i in range(min(12, idx.shape[0])):
        nn, yy, xx, cc = idx[i].tolist()
        print((nn, yy, xx, cc), "ring", o[nn, yy, xx, cc].item(), "ref", ref[nn, yy, xx, cc].item(), "out2 ring/ref", o2[nn, yy, xx, cc - 16].item() if 16 <= cc < 80 else None, ref2[nn, yy, xx, cc - 16].item() if 16 <= cc < 80 else None)
    import collections
    print(collections.Counter((idx[:, 3] // 4 * 4).tolist()))
