"""Stem kernels (VGG19 conv1_1 forward from the NCHW image, and its input gradient) at the headline shape: time per launch with HIP events.
Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops
dev = torch.device("cuda")
n, h, w = int(os.environ.get("N", 16)), 1024, 1024
img = torch.rand(n, 3, h, w, device=dev)
wt = torch.randn(64, 3, 3, 3, device=dev) / 5
b = torch.randn(64, device=dev)
act = torch.empty(n, h, w, 64, device=dev)
gy = torch.randn(n, h, w, 64, device=dev)
grad = torch.zeros(n, 3, h, w, device=dev)
std = (0.229, 0.224, 0.225)
fw = lambda: ops.conv3x3_stem(img, wt, b, ops.view(act), mean=(0.485, 0.456, 0.406), std=std, pad_mode=L.PAD_REPLICATE, act=L.ACT_RELU)
bw = lambda: ops.conv3x3_dgrad_image(ops.view(gy), wt, n, h, w, grad, pad_mode=L.PAD_REPLICATE, ch_scale=[1 / s for s in std], accumulate=True)
for name, f, gb in (("stem fwd", fw, n * h * w * (12 + 256) / 1e9), ("stem dgrad", bw, n * h * w * (256 + 24) / 1e9)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms:.3f} ms  {gb / ms:.2f} TB/s algorithmic")
