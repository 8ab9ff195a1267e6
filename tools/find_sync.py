"""Which torch call inside optimize_parameters synchronises with the device?  (torch.cuda.set_sync_debug_mode; GPU box)"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--gan", "--batch", "8", "--lq", "64"]
import torch
import bench
args = bench.parse()
import trainner_redux_amd
from trainner_redux_amd.models import build_model
model = build_model(bench.make_opt(args))
gt = torch.rand(8, 3, 256, 256, device="cuda"); lq = torch.rand(8, 3, 64, 64, device="cuda")
for i in range(1, 4):
    model.feed_data({"lq": lq, "gt": gt}); model.optimize_parameters(i, 0, True)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
warnings.simplefilter("always")
import traceback
def showwarning(message, category, filename, lineno, file=None, line=None):
    print("SYNC:", message, "\n", "".join(traceback.format_stack(limit=7)[:-1]))
warnings.showwarning = showwarning
model.feed_data({"lq": lq, "gt": gt}); model.optimize_parameters(5, 0, True)
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
# host time of each third of the step, measured with explicit stream queries
import time
t = time.perf_counter(); model.feed_data({"lq": lq, "gt": gt}); model.optimize_parameters(6, 0, True); h = time.perf_counter() - t
q = torch.cuda.current_stream().query(); torch.cuda.synchronize(); w = time.perf_counter() - t
print(f"host returned after {h * 1e3:.1f} ms, stream idle at return: {q}, GPU done after {w * 1e3:.1f} ms")
