import sys, warnings; sys.path.insert(0,'.')
warnings.simplefilter("ignore")
import torch
from oracle import vgg as OV
import trainner_redux_amd
from trainner_redux_amd import ops
from trainner_redux_amd.losses import build_loss
def rel(a,b): return ((a.double()-b.double()).norm()/b.double().norm().clamp_min(1e-30)).item()
sd = OV.synthetic_state_dict(3)
g = torch.Generator().manual_seed(6)
for size in (32, 64):
  x = torch.rand(1,3,size,size,generator=g); gt = torch.rand(1,3,size,size,generator=g)
  for lw in ({"conv3_4":1.0},{"conv4_1":1.0},{"conv4_2":1.0},{"conv4_4":1.0}):
    try:
        loss = build_loss(dict(type="perceptualloss", loss_weight=1.0, layer_weights=lw)).cuda()
    except NotImplementedError as e:
        print(lw, 'n/a', e); continue
    loss.vgg.load_torchvision_state_dict(sd)
    xr = x.clone().requires_grad_(True)
    ref = OV.perceptual_loss(sd, xr, gt, layer_weights=lw)
    sum(ref.values()).backward()
    grad = torch.empty(1,3,size,size,device='cuda'); dummy=torch.zeros(8,device='cuda')
    loss.vgg.loss_and_grad(x.cuda(), gt.cuda(), layer_weights=lw, kind=loss.kind, eps=1e-12, compute_dtype=torch.float32,
        value_weight=1.0, value_out=dummy, total_weight=0.0, total_out=None, grad_weight=1.0, grad=grad, first=True)
    print(size, lw, 'grad rel', rel(grad.cpu(), xr.grad))
