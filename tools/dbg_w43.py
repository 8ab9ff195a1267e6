import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops
dev = torch.device("cuda")
torch.manual_seed(0)
n, h, w, cin, cout = 1, 16, 32, 64, 64
wt = torch.randn(cout, cin, 3, 3, dtype=torch.float64) * math.sqrt(2 / (cin * 9))
gy = torch.randn(n, cout, h, w, dtype=torch.float64)
m = torch.randn(n, cin, h, w, dtype=torch.float64)
refg = F.conv_transpose2d(gy, wt, padding=1)
Ut = ops.pack_winograd43(wt.float().to(dev), transpose=True)
gyd = gy.float().permute(0, 2, 3, 1).contiguous().to(dev)
md = m.float().permute(0, 2, 3, 1).contiguous().to(dev)
for use_mask in (False, True):
    gin = torch.zeros(n, h, w, cin, device=dev)
    kw = dict(mask=ops.view(md), mask_slope=0.0) if use_mask else {}
    ops.conv3x3(n=n, h=h, w=w, cin=cout, cout=cin, compute=L.F32_WINOGRAD43, inp=ops.view(gyd), out=ops.view(gin), wpacked=Ut, **kw)
    r = refg * (m > 0) if use_mask else refg
    got = gin.permute(0, 3, 1, 2).cpu().double()
    print("mask", use_mask, "rel", ((got - r).norm() / r.norm()).item())
    # F(2,3) with the same inputs
    U2 = ops.pack_winograd(wt.float().to(dev), transpose=True)
    g2 = torch.zeros(n, h, w, cin, device=dev)
    ops.conv3x3(n=n, h=h, w=w, cin=cout, cout=cin, compute=L.F32_WINOGRAD, inp=ops.view(gyd), out=ops.view(g2), wpacked=U2, **kw)
    print("   F(2,3) rel", ((g2.permute(0, 3, 1, 2).cpu().double() - r).norm() / r.norm()).item())
# forward with the transposed weight tensor explicitly materialised
wtt = wt.flip(2, 3).permute(1, 0, 2, 3).contiguous()
Uf = ops.pack_winograd43(wtt.float().to(dev))
gin = torch.zeros(n, h, w, cin, device=dev)
ops.conv3x3(n=n, h=h, w=w, cin=cout, cout=cin, compute=L.F32_WINOGRAD43, inp=ops.view(gyd), out=ops.view(gin), wpacked=Uf)
print("explicit transpose rel", ((gin.permute(0, 3, 1, 2).cpu().double() - refg).norm() / refg.norm()).item(), "pack equal", torch.equal(Uf, Ut), (Uf - Ut).abs().max().item())
