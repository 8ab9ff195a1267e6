import sys, warnings; sys.path.insert(0,'.')
warnings.simplefilter("ignore")
import torch
from torch.nn import functional as F
from oracle import vgg as OV
import trainner_redux_amd
from trainner_redux_amd import ops, _lib as L
from trainner_redux_amd.losses import build_loss
def rel(a,b): return ((a.double()-b.double()).norm()/b.double().norm().clamp_min(1e-30)).item()
sd = OV.synthetic_state_dict(3)
g = torch.Generator().manual_seed(6)
size=32
x = torch.rand(1,3,size,size,generator=g); gt = torch.rand(1,3,size,size,generator=g)
lw={"conv4_1":1.0}
loss = build_loss(dict(type="perceptualloss", loss_weight=1.0, layer_weights=lw)).cuda()
loss.vgg.load_torchvision_state_dict(sd)
# oracle with all intermediates
def feats(img):
    mean=torch.tensor(OV.MEAN).view(1,3,1,1); std=torch.tensor(OV.STD).view(1,3,1,1)
    h=(img-mean)/std; outs={}
    for idx,name in enumerate(OV.VGG19_LAYERS[:OV.VGG19_LAYERS.index("conv4_1")+1]):
        if name.startswith("conv"):
            w,b=sd[f"features.{idx}.weight"],sd[f"features.{idx}.bias"]
            h = F.conv2d(F.pad(h,(1,1,1,1),mode="replicate"),w,b) if idx==0 else F.conv2d(h,w,b,padding=1)
        elif name.startswith("relu"): h=F.relu(h)
        else: h=F.max_pool2d(h,2,2)
        h.retain_grad() if h.requires_grad else None
        outs[name]=h
    return outs
xr=x.clone().requires_grad_(True)
fx=feats(xr); fg=feats(gt)
d=fx["conv4_1"]-fg["conv4_1"].detach()
l=torch.sqrt(d*d+1e-12).mean(); l.backward()
# HIP: monkeypatch ops to capture outputs
log=[]
oc, op, ol = ops.conv3x3, ops.maxpool2x2_bwd, ops.pixel_loss
vgg=loss.vgg
grad = torch.empty(1,3,size,size,device='cuda'); dummy=torch.zeros(8,device='cuda')
vgg.loss_and_grad(x.cuda(), gt.cuda(), layer_weights=lw, kind=loss.kind, eps=1e-12, compute_dtype=torch.float32,
    value_weight=1.0, value_out=dummy, total_weight=0.0, total_out=None, grad_weight=1.0, grad=grad, first=True)
ws=list(vgg._ws.values())[0]
prog=vgg.program
# compare forward activations
k=0
names=[]
for i,opx in enumerate(prog):
    if opx[0]=='conv':
        nm=opx[1]; ref=fx[nm] if not opx[3] else fx[nm.replace('conv','relu')]
        print(i,nm,'act rel', rel(ws['acts'][i].permute(0,3,1,2).float().cpu(), ref.detach()))
    else:
        pn=[n for n in fx if n.startswith('pool')][sum(1 for o in prog[:i] if o[0]=='pool')]
        print(i,pn,'act rel', rel(ws['acts'][i].permute(0,3,1,2).float().cpu(), fx[pn].detach()))
print('final grad rel', rel(grad.cpu(), xr.grad))
# now redo reverse sweep manually step by step comparing with oracle grads
V=ops.view
n=1
pk=vgg._pack(L.F32,'cuda:0' if False else torch.device('cuda',0))
dims=[];hh=ww=size
for o in prog:
    dims.append((hh,ww))
    if o[0]=='pool': hh//=2; ww//=2
gcur=None
for kk in range(len(prog)-1,-1,-1):
    o=prog[kk]; hh,ww=dims[kk]
    if o[0]=='pool':
        src=ws['acts'][kk-1]; gin=torch.zeros_like(src)
        ops.maxpool2x2_bwd(V(src),V(gcur),V(gin),n,hh,ww,True); gcur=gin
        # oracle grad wrt conv output pre-relu? src is post-relu act of conv kk-1 (fused) -> compare with grad of conv output
        nm=prog[kk-1][1]; print(kk,'pool-bwd -> g(',nm,') rel', rel(gcur.permute(0,3,1,2).cpu(), fx[nm].grad)); continue
    _,name,m,fused,tapped=o
    act=ws['acts'][kk]
    if tapped:
        gcur=torch.zeros_like(act); ops.pixel_loss(loss.kind, act, ws['feats_gt'][name], dummy, grad=gcur, grad_weight=1.0)
        print(kk,'loss g rel', rel(gcur.permute(0,3,1,2).cpu(), fx[name].grad))
    if kk==0: break
    prev=prog[kk-1]; inp=ws['acts'][kk-1]; gin=torch.zeros_like(inp)
    mask=V(inp) if (prev[0]=='conv' and prev[3]) else ops.NULL_VIEW
    ops.conv3x3(n=n,h=hh,w=ww,cin=m.cout,cout=m.cin,compute=L.F32,inp=V(gcur),out=V(gin),wpacked=pk[name][1],mask=mask,mask_slope=0.0)
    gcur=gin
    if prev[0]=='conv':
        print(kk,'dgrad',name,'-> g(',prev[1],') rel', rel(gcur.permute(0,3,1,2).cpu(), fx[prev[1]].grad))
    else:
        pn=[n_ for n_ in fx if n_.startswith('pool')][sum(1 for o2 in prog[:kk-1] if o2[0]=='pool')]
        print(kk,'dgrad',name,'-> g(',pn,') rel', rel(gcur.permute(0,3,1,2).cpu(), fx[pn].grad))
print("---- isolate pool3 bwd")
pre = fx['conv3_4'].detach()                       # oracle pre-relu
gp = fx['pool3'].grad                              # oracle grad wrt pool3 out
pr = pre.clone().requires_grad_(True)
F.max_pool2d(F.relu(pr),2,2).backward(gp)
ref_g = pr.grad
src = F.relu(pre).permute(0,2,3,1).contiguous().cuda()
gpd = gp.permute(0,2,3,1).contiguous().cuda()
gin = torch.zeros_like(src)
ops.maxpool2x2_bwd(V(src), V(gpd), V(gin), 1, 8, 8, True)
got = gin.permute(0,3,1,2).cpu()
print('isolated rel', rel(got, ref_g), 'ref vs fx grad', rel(ref_g, fx['conv3_4'].grad))
bad = (got-ref_g).abs() > 1e-9
print('n bad', bad.sum().item(), 'of', bad.numel())
idx = bad.nonzero()[:6]
for (b_,c_,y_,x_) in idx.tolist():
    y0,x0=(y_//2)*2,(x_//2)*2
    print((c_,y_,x_), 'window pre', pre[0,c_,y0:y0+2,x0:x0+2].flatten().tolist(), 'got', got[0,c_,y0:y0+2,x0:x0+2].flatten().tolist(), 'ref', ref_g[0,c_,y0:y0+2,x0:x0+2].flatten().tolist(), 'gp', gp[0,c_,y_//2,x_//2].item())
print("---- sensitivity")
hip_act = ws['acts'][9].permute(0,3,1,2).float().cpu()      # post-relu conv3_4 from HIP
ora_act = F.relu(pre)
def amax(t):
    n_,c_,h_,w_=t.shape
    u=t.view(n_,c_,h_//2,2,w_//2,2).permute(0,1,2,4,3,5).reshape(n_,c_,h_//2,w_//2,4)
    return u.argmax(-1), u
ia,ua=amax(hip_act); ib,ub=amax(ora_act)
flip=(ia!=ib)
print('argmax flips', flip.sum().item(), 'of', flip.numel())
for (b_,c_,y_,x_) in flip.nonzero()[:8].tolist():
    print((c_,y_,x_),'hip',ua[b_,c_,y_,x_].tolist(),'ora',ub[b_,c_,y_,x_].tolist())
print('max abs diff act', (hip_act-ora_act).abs().max().item(), 'act max', ora_act.max().item())
