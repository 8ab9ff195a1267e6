"""feed_data vs optimize_parameters wall time for the configs[2] step with the OTF feed (synchronised pieces; GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--gan", "--otf", "--batch", "8", "--lq", "64"]
import torch
import bench
args = bench.parse()
import trainner_redux_amd
from trainner_redux_amd.models import build_model
model = build_model(bench.make_opt(args))
n, s, sc = 8, 64, 4
pulse = torch.zeros(21, 21); pulse[10, 10] = 1
k = torch.rand(n, 21, 21); k /= k.sum((1, 2), keepdim=True)
batch = {"gt": torch.rand(n, 3, sc * s + 32, sc * s + 32), "kernel1": k, "kernel2": k, "sinc_kernel": pulse.repeat(n, 1, 1)}
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for i in range(1, 40):
    t0 = sync(); model.feed_data(batch); t1 = sync(); model.optimize_parameters(i, 0, True); t2 = sync()
    if i % 4 == 0: print(f"step {i}: feed {1e3 * (t1 - t0):.2f} ms, optimize {1e3 * (t2 - t1):.2f} ms")
t0 = sync()
for i in range(40, 60):
    model.feed_data(batch); model.optimize_parameters(i, 0, True)
t1 = sync(); print(f"unsynchronised loop: {(t1 - t0) / 20 * 1e3:.2f} ms/step")
lq, gt = model.lq.clone(), model.gt.clone()
t0 = sync()
for i in range(60, 80):
    model.lq, model.gt = lq, gt; model.optimize_parameters(i, 0, True)
t1 = sync(); print(f"optimize only, resident pair: {(t1 - t0) / 20 * 1e3:.2f} ms/step")
t0 = sync()
model.prefetch(batch)
for i in range(80, 100):
    model.feed_data(batch); model.prefetch(batch); model.optimize_parameters(i, 0, True)
t1 = sync(); print(f"with side-stream prefetch: {(t1 - t0) / 20 * 1e3:.2f} ms/step")
import collections
acc = collections.defaultdict(float)
model.prefetch(batch)
t0 = sync()
for i in range(100, 120):
    a = time.perf_counter(); model.feed_data(batch); b = time.perf_counter(); model.prefetch(batch); c = time.perf_counter()
    model.optimize_parameters(i, 0, True); d = time.perf_counter()
    acc["feed"] += b - a; acc["prefetch"] += c - b; acc["optimize(host)"] += d - c
t1 = sync(); print("host time per step:", {k: round(v / 20 * 1e3, 2) for k, v in acc.items()}, f"total {(t1 - t0) / 20 * 1e3:.2f}")
from trainner_redux_amd.data import otf as O
st = model.otf
acc.clear()
orig_finish, orig_start = st.finish, st.start
def finish(s):
    a = time.perf_counter(); s["event"].synchronize(); b = time.perf_counter()
    lq = st._after_codec_stages(s["out"], s["host"]); c = time.perf_counter()
    lq = O.clamp_round_u8(lq); gt, lq = O.paired_random_crop(s["gt"], lq, st.gt_size, st.scale, rnd=st.rnd); lq, gt = st.pool(lq, gt); d = time.perf_counter()
    acc["ev.sync"] += b - a; acc["codec+h2d"] += c - b; acc["tail"] += d - c
    return lq.contiguous(), gt
st.finish = finish
model.prefetch(batch)
for i in range(120, 140):
    model.feed_data(batch); model.prefetch(batch); model.optimize_parameters(i, 0, True)
sync(); print("finish pieces:", {k: round(v / 20 * 1e3, 2) for k, v in acc.items()})
