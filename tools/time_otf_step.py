"""configs[2] step with the OTF feed: host time per phase in the free-running loop, and the loop's wall time (GPU box).
usage: python tools/time_otf_step.py [--graphs]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
graphs = "--graphs" in sys.argv
sys.argv = [sys.argv[0], "--gan", "--otf", "--batch", "8", "--lq", "64"] + (["--graphs"] if graphs else [])
import torch
import bench
args = bench.parse()
import trainner_redux_amd
from trainner_redux_amd.models import build_model
from trainner_redux_amd.data import otf as O
model = build_model(bench.make_opt(args))
n, s, sc = 8, 64, 4
pulse = torch.zeros(21, 21); pulse[10, 10] = 1
k = torch.rand(n, 21, 21); k /= k.sum((1, 2), keepdim=True)
batch = {"gt": torch.rand(n, 3, sc * s + 32, sc * s + 32), "kernel1": k, "kernel2": k, "sinc_kernel": pulse.repeat(n, 1, 1)}
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for i in range(1, 21):
    model.feed_data(batch); model.optimize_parameters(i, 0, True)
acc = collections.defaultdict(float)
orig_crt = O.compress_round_trip
def crt(img, fmt, q, host=None):
    a = time.perf_counter(); r = orig_crt(img, fmt, q, host); acc["codec:" + fmt] += time.perf_counter() - a; acc["n:" + fmt] += 1; return r
O.compress_round_trip = crt
st = model.otf
orig_finish, orig_start = st.finish, st.start
def start(*a):
    t = time.perf_counter(); r = orig_start(*a); acc["start"] += time.perf_counter() - t; return r
def finish(s_):
    t = time.perf_counter(); s_["event"].synchronize(); acc["event wait"] += time.perf_counter() - t
    t = time.perf_counter(); r = orig_finish(s_); acc["finish"] += time.perf_counter() - t; return r
st.start, st.finish = start, finish
N = 40
t0 = sync()
for i in range(21, 21 + N):
    a = time.perf_counter(); model.feed_data(batch); b = time.perf_counter(); model.optimize_parameters(i, 0, True); c = time.perf_counter()
    acc["feed_data"] += b - a; acc["optimize(host)"] += c - b
t1 = sync()
print(f"graphs={graphs}: wall {(t1 - t0) / N * 1e3:.2f} ms/step; host per step:", {k_: round(v / N * (1 if k_.startswith('n:') else 1e3), 2) for k_, v in acc.items()})
lq, gt = model.lq.clone(), model.gt.clone()
t0 = sync()
for i in range(100, 120):
    model.lq, model.gt = lq, gt; model.optimize_parameters(i, 0, True)
t1 = sync(); print(f"optimize only, resident pair: {(t1 - t0) / 20 * 1e3:.2f} ms/step")
h = 0.0
t0 = sync()
for i in range(200, 220):
    model.lq, model.gt = lq, gt
    a = time.perf_counter(); model.optimize_parameters(i, 0, True); h += time.perf_counter() - a
t1 = sync(); print(f"optimize only: host {h / 20 * 1e3:.2f} ms, wall {(t1 - t0) / 20 * 1e3:.2f} ms/step; graph mode: {getattr(model.net_g, '_gs', None) and model.net_g._gs.get('mode')}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(300, 305):
    model.lq, model.gt = lq, gt; model.optimize_parameters(i, 0, True)
pr.disable(); sync()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
