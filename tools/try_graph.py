"""Does hipGraph capture (torch.cuda.graph) work over the C ABI launches?  ESRGAN forward + backward, eager vs replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import trainner_redux_amd
from trainner_redux_amd.archs import build_network
dev = torch.device("cuda")
net = build_network({"type": "esrgan", "scale": 4, "num_blocks": 23}).to(dev).train()
net.compute_dtype = torch.bfloat16
x = torch.rand(8, 3, 64, 64, device=dev)
gy = torch.randn(8, 3, 256, 256, device=dev) * 1e-3
for _ in range(2):
    y = net.forward_hip(x, train=True); net.backward_hip(gy)
torch.cuda.synchronize()
g0 = net.flat_parameters()[1].clone()
def run_eager():
    y = net.forward_hip(x, train=True); net.backward_hip(gy); return y
t = time.perf_counter()
for _ in range(10): run_eager()
t_host = (time.perf_counter() - t) / 10; torch.cuda.synchronize(); t_all = (time.perf_counter() - t) / 10
print(f"eager: host {t_host * 1e3:.1f} ms, wall {t_all * 1e3:.1f} ms per fwd+bwd")
net.flat_parameters()[1].zero_()
y_ref = net.forward_hip(x, train=True).clone(); net.backward_hip(gy); g_ref = net.flat_parameters()[1].clone()
gf, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    net.forward_hip(x, train=True); net.backward_hip(gy)      # warm the per-stream scratch
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(gf):
    y_static = net.forward_hip(x, train=True)
saved = net._saved
with torch.cuda.graph(gb, pool=gf.pool()):
    net.backward_hip(gy)
torch.cuda.synchronize()
net.flat_parameters()[1].zero_()
gf.replay(); gb.replay(); torch.cuda.synchronize()
print("replay output equal:", torch.equal(y_static, y_ref), " grads equal:", torch.equal(net.flat_parameters()[1], g_ref))
t = time.perf_counter()
for _ in range(10): gf.replay(); gb.replay()
t_host = (time.perf_counter() - t) / 10; torch.cuda.synchronize(); t_all = (time.perf_counter() - t) / 10
print(f"graph: host {t_host * 1e3:.1f} ms, wall {t_all * 1e3:.1f} ms per fwd+bwd")
