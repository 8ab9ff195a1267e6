import sys, warnings; sys.path.insert(0,'.')
warnings.simplefilter("ignore")
import torch
from oracle import vgg as OV
import trainner_redux_amd
from trainner_redux_amd.losses import build_loss
def rel(a,b): return ((a.double()-b.double()).norm()/b.double().norm().clamp_min(1e-30)).item()
sd = OV.synthetic_state_dict(3)
loss = build_loss(dict(type="perceptualloss", loss_weight=1.0)).cuda()
loss.vgg.load_torchvision_state_dict(sd)
g = torch.Generator().manual_seed(6)
x = torch.rand(1,3,32,32,generator=g); gt = torch.rand(1,3,32,32,generator=g)
names = loss.vgg.layer_names
for j,nm in enumerate(names):
    xr = x.clone().requires_grad_(True)
    ref = OV.perceptual_loss(sd, xr, gt)
    ref[nm].backward()
    sc = {n:(1.0 if n==nm else 0.0) for n in names}
    grad = torch.empty(1,3,32,32,device='cuda'); dummy=torch.zeros(8,device='cuda')
    loss.vgg.loss_and_grad(x.cuda(), gt.cuda(), layer_weights=loss.layer_weights, kind=loss.kind, eps=1e-12, compute_dtype=torch.float32,
        value_weight=1.0, value_out=dummy, total_weight=0.0, total_out=None, grad_weight=1.0, grad=grad, first=True, layer_grad_scale=sc)
    print(nm, 'val', dummy[j].item(), float(ref[nm]), 'grad rel', rel(grad.cpu(), xr.grad), 'norm', xr.grad.norm().item())
