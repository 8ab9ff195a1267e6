"""Runs bench.py on the secondary workloads (BASELINE configs[2..4] and the A/B forms of the headline) and writes profiles-style JSON:
   python tools/bench_variants.py gpurun_out/variants.json          (on the GPU box; each workload is a child process of its own)"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--no-cpu-baseline", "--no-variants", "--no-loader"]
RUNS = [("--arch swinir_m", 20, 5), ("--arch span --compute f16", 20, 5), ("--arch mosr --compute f16", 20, 5), ("--gan --batch 8 --lq 64", 10, 4),
        ("--gan --otf --batch 8 --lq 64", 10, 4), ("--compute f16", 4, 2), ("--losses l1+perceptualfp16", 4, 2), ("--losses l1", 4, 2)]
out = []
for flags, steps, warm in RUNS:
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), *flags.split(), "--steps", str(steps), "--warmup", str(warm), *COMMON]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    if r.returncode or not line:
        print("FAILED", flags, r.stderr[-400:], flush=True)
        out.append({"flags": flags, "error": r.stderr[-400:]})
        continue
    d = json.loads(line[-1])
    rf = d.get("roofline") or {}
    rf.pop("all", None)
    out.append({"flags": flags, "workload": d["config"]["workload"], "value": round(d["value"], 1), "unit": d["unit"], "ms_per_step": round(d["ms_per_step"], 2),
                "dtype": d["dtype"], "roofline": rf})
    # the same run without the per-launch HIP events of the roofline record (two events around every conv launch: a few % at 256 -> 1024, a third of the
    # step at the launch-bound 64 -> 256 tiles)
    r2 = subprocess.run(cmd + ["--no-roofline"], capture_output=True, text=True, timeout=900)
    l2 = [l for l in r2.stdout.strip().splitlines() if l.startswith("{")]
    if r2.returncode == 0 and l2:
        d2 = json.loads(l2[-1])
        out[-1]["value_without_event_timers"] = round(d2["value"], 1)
        out[-1]["ms_per_step_without_event_timers"] = round(d2["ms_per_step"], 2)
    print(flags, "->", out[-1]["value"], d["unit"], out[-1]["ms_per_step"], "ms/step; without the event timers", out[-1].get("value_without_event_timers"), flush=True)
json.dump(out, open(sys.argv[1], "w"), indent=1)
