"""Adds / replaces one workload entry of profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of a bench.py workload.
   python tools/make_traffic_json.py <workload key, bench.py::workload_key> <fetch_dir> <write_dir> [source note]
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: counters are in KB and FETCH_SIZE reports half of wide coalesced streaming
reads on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from prof_summary import pmc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl, fe, wr = sys.argv[1], pmc(sys.argv[2], "FETCH_SIZE"), pmc(sys.argv[3], "WRITE_SIZE")
groups = {}
for k in fe:
    m = re.search(r"conv3x3_kernel_reg<(\d), (\d), (\d), (\d)>", k)
    mr = re.search(r"conv_ring_kernel<(\d), (\d),", k)
    ml = re.search(r"linear(?:_dma)?_kernel<(\d), (\d)", k)      # token GEMM, register-staged or LDS-DMA staged: one group
    name = lambda ct: "f32" if ct == 0 else "bf16" if ct == 1 else "f16"  # noqa: E731
    if "winograd43_f32_kernel" in k:
        key = "f32/winograd43"
    elif "winograd_f32_kernel" in k:
        key = "f32/winograd"
    elif mr:     # the LDS-DMA ring kernels: same grouping as bench.py's per-launch timers (compute dtype / output-group width)
        key = name(int(mr.group(1))) + ("/cg64" if int(mr.group(2)) == 2 else "/cg32")
    elif ml:
        key = name(int(ml.group(1))) + "/cg64/k1"
    elif m:
        key = name(int(m.group(1))) + ("/cg64" if int(m.group(2)) == 2 else "/cg32") + ("" if int(m.group(4)) == 3 else f"/k{m.group(4)}")
    else:
        continue
    g = groups.setdefault(key, [0, 0.0, 0.0])
    g[0] += fe[k][0]
    g[1] += fe[k][1]
    g[2] += wr.get(k, [0, 0.0])[1]
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path)) if os.path.exists(path) else {"workloads": {}}
entry = {"_source": sys.argv[4] if len(sys.argv) > 4 else ""}
for key, (n, f, w) in groups.items():
    entry[key] = {"fetch_kb_per_launch": f / n, "write_kb_per_launch": w / n, "traffic_bytes_per_launch": (2 * f + w) / n * 1024, "launches_profiled": n}
tj.setdefault("workloads", {})[wl] = entry
json.dump(tj, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
