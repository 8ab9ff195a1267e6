"""In-kernel phase stamps of the ring conv kernel (diagnostics): per workgroup, consumer wave 0 stamps [before barrier, after barrier, after MFMA chunk]
per iteration, loader wave 0 stamps [after issue, after wait, after barrier].  Prints median durations per phase."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import trainner_redux_amd
from trainner_redux_amd import _lib as L, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_ring as AB

dev = torch.device("cuda")
n, h, w, cin, cout = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (16, 256, 256, 64, 32))]
form = sys.argv[6] if len(sys.argv) > 6 else "lrelu"
cfg = int(sys.argv[7]) if len(sys.argv) > 7 else 10
bufs = AB.mk_bufs(n, h, w, cin, cout, torch.bfloat16, L.BF16)
_, _, f = AB.run("ring", cfg, n, h, w, cin, cout, form, bufs=bufs)
ops.tune_set("conv_impl", 2); ops.tune_set("ring_cfg", cfg)
for _ in range(3): f()
st = torch.zeros(1024 * 2 * 256, dtype=torch.int64, device=dev)   # the kernel stamps workgroups below 1024 only
ops.tune_set("stamps", st.data_ptr())          # honoured by builds with -DTR_RING_DEV only (TRAINNER_DEV_BUILD=1 python trainner-redux_amd/build.py --force)
f(); torch.cuda.synchronize()
ops.tune_set("stamps", 0)
s = st.cpu().numpy().reshape(1024, 2, 256)[:256]
nchunks = cin // 32
for b in (0, 100, 255):
    c = s[b, 0]; c = c[c > 0]; lo = s[b, 1]; lo = lo[lo > 0]
    t0 = min(c[0], lo[0])
    print(f"WG {b}: consumer stamps {len(c)}, loader stamps {len(lo)}; total {(max(c[-1], lo[-1]) - t0)} ticks")
    print("  consumer (rel):", (c[:40] - t0).tolist())
    print("  loader   (rel):", (lo[:40] - t0).tolist())
# consumer pattern per iteration (after the D-1 lead-in): [pre-barrier, post-barrier, post-mfma]
allc = []
for b in range(256):
    c = s[b, 0]; c = c[c > 0]
    if len(c) < 12: continue
    d = np.diff(c)
    allc.append(d[:60])
m = min(len(x) for x in allc)
med = np.median(np.stack([x[:m] for x in allc]), axis=0)
print("median consumer deltas:", med.astype(int).tolist())
alll = []
for b in range(256):
    c = s[b, 1]; c = c[c > 0]
    if len(c) < 12: continue
    alll.append(np.diff(c)[:60])
m = min(len(x) for x in alll)
print("median loader deltas:", np.median(np.stack([x[:m] for x in alll]), axis=0).astype(int).tolist())
