"""Which kernel family / instance serves each distinct conv launch of one bench step (run on the GPU box): python tools/conv_families.py "--arch mosr --compute f16"."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from trainner_redux_amd import ops, _lib as L
from trainner_redux_amd.models import build_model
flags = sys.argv[1] if len(sys.argv) > 1 else ""
sys.argv = ["bench.py", *flags.split()]
args = bench.parse()
model = build_model(bench.make_opt(args))
orig = ops.conv3x3
cnt = collections.Counter()
tim = collections.defaultdict(list)
def wrapped(**kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(**kw); e1.record()
    extra = tuple(k for k in ("act", "mask", "res1", "res2", "out2", "sample_scale", "mask_mode", "in_mode", "ksize", "pad_mode") if kw.get(k) is not None and not (hasattr(kw[k], "ptr") and not kw[k].ptr) and kw.get(k) not in (0, L.ACT_NONE))
    cd = {L.BF16: "bf16", L.F16: "f16", L.F32: "f32"}.get(kw["compute"], str(kw["compute"]))
    key = (cd, kw["cin"], kw["cout"], kw["h"], kw["w"], "out32" if kw["out"].dtype == L.F32 else "out16", extra, ops.last_conv_family(), ops.tune_get("last_conv_instance"))
    cnt[key] += 1; tim[key].append((e0, e1))
ops.conv3x3 = wrapped
n, s, sc = args.batch, args.lq, args.scale
gt = torch.rand(n, 3, sc * s, sc * s).cuda(); lq = torch.nn.functional.interpolate(gt, scale_factor=1.0 / sc, mode="bicubic", antialias=True).clamp(0, 1)
for it in (1, 2):
    cnt.clear(); tim.clear()
    model.feed_data({"lq": lq, "gt": gt}); model.optimize_parameters(it, 0, True)
torch.cuda.synchronize()
rows = [(sum(a.elapsed_time(b) for a, b in tim[k]), v, k) for k, v in cnt.items()]
for ms, v, k in sorted(rows, reverse=True)[:int(os.environ.get("TOP", 16))]:
    print(f"{ms:8.3f} ms  x{v:<4d} {k}")
