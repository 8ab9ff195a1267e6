import sys; sys.path.insert(0,'.')
import torch, math
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops
def rel(a,b): return ((a.double()-b.double()).norm()/b.double().norm().clamp_min(1e-30)).item()
dev='cuda'
for (n,h,w,cin,cout) in [(1,4,4,512,512),(1,4,4,256,512),(1,8,8,256,256),(1,4,4,128,512),(1,4,4,512,128),(1,4,4,256,256),(1,4,4,320,64),(1,4,4,64,320)]:
    g=torch.Generator().manual_seed(1)
    wt=torch.randn(cout,cin,3,3,generator=g)/math.sqrt(cin*9)
    gy=torch.randn(n,cout,h,w,generator=g)
    x=torch.randn(n,cin,h,w,generator=g)
    ref=torch.nn.grad.conv2d_input((n,cin,h,w),wt,gy,padding=1)
    buf=torch.zeros(ops.packed_bytes(cout,cin,L.F32),dtype=torch.uint8,device=dev)
    plan=ops.PackPlan(dev); wd=wt.to(dev); plan.add(wd,buf,cout,cin,L.F32,transpose=True); plan.run()
    gyd=gy.permute(0,2,3,1).contiguous().to(dev); out=torch.empty(n,h,w,cin,device=dev)
    ops.conv3x3(n=n,h=h,w=w,cin=cout,cout=cin,compute=L.F32,inp=ops.view(gyd),out=ops.view(out),wpacked=buf)
    print('dgrad',(n,h,w,cin,cout), rel(out.permute(0,3,1,2).cpu(),ref))
    # forward too
    reff=torch.nn.functional.conv2d(x,wt,None,padding=1)
    buf2=torch.zeros(ops.packed_bytes(cin,cout,L.F32),dtype=torch.uint8,device=dev)
    plan2=ops.PackPlan(dev); plan2.add(wd,buf2,cin,cout,L.F32); plan2.run()
    xd=x.permute(0,2,3,1).contiguous().to(dev); o2=torch.empty(n,h,w,cout,device=dev)
    ops.conv3x3(n=n,h=h,w=w,cin=cin,cout=cout,compute=L.F32,inp=ops.view(xd),out=ops.view(o2),wpacked=buf2)
    print('  fwd',(n,h,w,cin,cout), rel(o2.permute(0,3,1,2).cpu(),reff))
