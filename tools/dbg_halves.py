"""Debug: gradient of a full-batch accumulate-only step vs the mean of its halves, per parameter (run on the GPU box).
   python tools/dbg_halves.py "--arch mosr --compute f16" 64 64"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trainner_redux_amd.models import build_model

flags, batch, lq = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
sys.argv = ["bench.py", *flags.split(), "--batch", str(batch), "--lq", str(lq)]
args = bench.parse()
torch.manual_seed(1024)
model = build_model(bench.make_opt(args))
if os.environ.get("SCALE"):
    model.scaler_g.scale = float(os.environ["SCALE"])
net = model.net_g
if hasattr(net, "drop_path_rate"): net.drop_path_rate = 0.0
sc, h = args.scale, batch // 2
g = torch.Generator().manual_seed(11)
gt = torch.rand(batch, 3, sc * lq, sc * lq, generator=g).cuda()
lqt = torch.nn.functional.interpolate(gt, scale_factor=1.0 / sc, mode="bicubic", antialias=True).clamp(0, 1)
_, garr = net.flat_parameters()
def grads_of(parts):
    garr.zero_()
    for sl in parts:
        model.feed_data({"lq": lqt[sl], "gt": gt[sl]})
        model.optimize_parameters(1, 0, False)
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone().double() for k, p in net.named_parameters() if p.grad is not None}
print("loss scale", model.scaler_g.scale)
gf = grads_of([slice(0, batch)])
gh = grads_of([slice(0, h), slice(h, batch)])
rows = []
for k in gf:
    half = gh[k] * 0.5
    rows.append(((gf[k] - half).norm() / half.norm().clamp_min(1e-30)).item())
    
names = list(gf)
order = sorted(range(len(rows)), key=lambda i: -rows[i])
for i in order[:14]: print(f"{names[i]:50s} rel {rows[i]:.3e}  |g| {gf[names[i]].norm().item():.3e}")
tot = torch.cat([v.flatten() for v in gf.values()]); toth = torch.cat([v.flatten() * 0.5 for v in gh.values()])
print("total rel", ((tot - toth).norm() / toth.norm()).item())
gf2 = grads_of([slice(0, batch)])
gh2 = grads_of([slice(0, h), slice(h, batch)])
def cmp(a, b, tag):
    r = [((a[k] - b[k]).norm() / b[k].norm().clamp_min(1e-30)).item() for k in a]
    i = max(range(len(r)), key=lambda j: r[j])
    print(tag, "max per-param rel", f"{r[i]:.3e}", names[i], " #params > 1e-3:", sum(x > 1e-3 for x in r))
cmp(gf, gf2, "full vs full again:")
cmp(gh, gh2, "halves vs halves again:")
