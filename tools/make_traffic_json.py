"""profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the default bench workload.
   python tools/make_traffic_json.py <fetch_dir> <write_dir> > profiles/traffic.json
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: counters are in KB and FETCH_SIZE reports half of wide coalesced streaming
reads on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import json, re, sys
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from prof_summary import pmc

fe, wr = pmc(sys.argv[1], "FETCH_SIZE"), pmc(sys.argv[2], "WRITE_SIZE")
out = {"_comment": "HBM bytes per launch from rocprofv3 PMC passes (profiles/r02_pmc_traffic_headline.txt): (2*FETCH_SIZE + WRITE_SIZE)*1024, the gfx950 "
                   "FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section); workload = bench.py default (ESRGAN l1+perceptual bf16 256->1024 batch 16)",
       "workload": "l1+perceptual/bf16/batch16/lq256/blocks23"}
groups = {}
for k in fe:
    m = re.search(r"conv3x3_kernel_reg<(\d), (\d), (\d), 3>", k)
    mr = re.search(r"conv_ring_kernel<(\d), (\d),", k)
    if "winograd43_f32_kernel" in k:
        key = "f32/winograd43"
    elif "winograd_f32_kernel" in k:
        key = "f32/winograd"
    elif mr:     # the LDS-DMA ring kernels: same grouping as bench.py's per-launch timers (compute dtype / output-group width)
        ct, nt = int(mr.group(1)), int(mr.group(2))
        key = ("f32" if ct == 0 else "bf16" if ct == 1 else "f16") + ("/cg64" if nt == 2 else "/cg32")
    elif not m:
        continue
    else:
        ct, nt = int(m.group(1)), int(m.group(2))
        key = ("f32" if ct == 0 else "bf16" if ct == 1 else "f16") + ("/cg64" if nt == 2 else "/cg32")
    g = groups.setdefault(key, [0, 0.0, 0.0])
    g[0] += fe[k][0]
    g[1] += fe[k][1]
    g[2] += wr.get(k, [0, 0.0])[1]
for key, (n, f, w) in groups.items():
    out[key] = {"fetch_kb_per_launch": f / n, "write_kb_per_launch": w / n, "traffic_bytes_per_launch": (2 * f + w) / n * 1024, "launches_profiled": n}
print(json.dumps(out, indent=1))
