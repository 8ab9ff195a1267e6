"""Where the OTF feed's time goes (host timers around synchronised pieces; run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import trainner_redux_amd
from trainner_redux_amd.data import otf
dev = torch.device("cuda")
n, s = 8, 288
gt = torch.rand(n, 3, s, s)
k = torch.rand(n, 21, 21); k /= k.sum((1, 2), keepdim=True)
def T(f, reps=10):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
g = gt.to(dev); kd = k.to(dev)
print("h2d gt                %.2f ms" % T(lambda: gt.to(dev)))
for m in ("bilinear", "bicubic", "nearest-exact", "lanczos"):
    print("resize %-14s %.2f ms" % (m, T(lambda: otf.resize_pt(g, m, size=(72, 72)))))
lq = otf.resize_pt(g, "bicubic", size=(72, 72))
print("filter2d (72)         %.2f ms" % T(lambda: otf.filter2d(lq, kd)))
print("filter2d (288)        %.2f ms" % T(lambda: otf.filter2d(g, kd)))
print("jpeg round trip       %.2f ms" % T(lambda: otf.compress_round_trip(lq, "jpeg", 70)))
print("webp round trip       %.2f ms" % T(lambda: otf.compress_round_trip(lq, "webp", 70)))
print("clamp/round           %.2f ms" % T(lambda: otf.clamp_round_u8(lq)))
print("paired crop           %.2f ms" % T(lambda: otf.paired_random_crop(g, lq, 256, 4)))
pool = otf.PairPool(120)
a, b = otf.paired_random_crop(g, lq, 256, 4)
for _ in range(16): pool(b, a)
print("pair pool (full)      %.2f ms" % T(lambda: pool(b, a)))
