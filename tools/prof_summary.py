"""Condenses rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/."""
import csv, glob, re, sys, collections

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(ConvArgs\)|\(WgArgs\)", "", n)
    return n[:80]

def kernel_stats(d):
    f = (glob.glob(f"{d}/*/*kernel_stats.csv") + glob.glob(f"{d}/*kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = [f"total GPU kernel time {tot / 1e6:.1f} ms", f"{'kernel':80s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}"]
    for r in rows[:18]:
        out.append(f"{short(r['Name']):80s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")
    return "\n".join(out)

def pmc(d, counter):
    f = (glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg

def pmc_all(d):
    """every counter of a --pmc pass, averaged per (kernel, grid size, LDS bytes)"""
    f = (glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        key = (short(r["Kernel_Name"])[:48], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        agg[key]["_us"] += 0.0
        cnt[key].add(r["Dispatch_Id"])
        agg[key]["_t0"] = min(agg[key].get("_t0", 1e30) or 1e30, float(r["Start_Timestamp"]))
    out = []
    for key, cs in agg.items():
        n = len(cnt[key])
        names = [c for c in cs if not c.startswith("_")]
        out.append(f"{key[0]:48s} grid {key[1]:>8s} lds {key[2]:>6s} launches {n:4d} | " + "  ".join(f"{c}={cs[c] / n:.3e}" for c in sorted(names)))
    return "\n".join(out)

def mfma_busy(d):
    """MFMA utilisation per kernel from a `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE ...` pass:
    util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)  (the gfx94x MfmaUtil formula; rocprofv3 sums GRBM over the XCDs);
    clock = GRBM_GUI_ACTIVE / 8 / kernel wall time (MI355X_MICROARCH.md, DVFS give-back)."""
    f = (glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen, dur = collections.defaultdict(set), collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])[:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen[k]:
            seen[k].add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    out = [f"{'kernel':70s} {'calls':>6s} {'avg_us':>9s} {'MFMA_BUSY/launch':>17s} {'GUI_ACTIVE/8':>13s} {'mfma_util':>9s} {'clock_GHz':>9s} {'MFMA_MOPS/launch':>17s}"]
    for k in sorted(agg, key=lambda k: -dur[k])[:16]:
        c, n = agg[k], len(seen[k])
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        mops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) + c.get("SQ_INSTS_VALU_MFMA_MOPS_F16", 0.0)
        out.append(f"{k:70s} {n:6d} {dur[k] / n:9.1f} {mf / n:17.3e} {gui / n:13.3e} {(mf / (gui * 1024) if gui else 0):9.3f} {(gui / dur[k] / 1e3 if dur[k] else 0):9.2f} {mops / n:17.3e}")
    return "\n".join(out)


def pmc_seq(d, counter, pattern):
    """one line per dispatch, in dispatch order, of the kernels whose name contains `pattern` (per-layer reading of a scripted launch sequence)"""
    f = (glob.glob(f"{d}/*/*counter_collection.csv") + glob.glob(f"{d}/*counter_collection.csv"))[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and pattern in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return "\n".join(f"dispatch {r['Dispatch_Id']:>6s} {short(r['Kernel_Name'])[:60]:60s} {counter}={float(r['Counter_Value']):14.1f} "
                     f"us={(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f}" for r in rows)


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "seq":
        print(pmc_seq(sys.argv[2], sys.argv[3], sys.argv[4]))
    elif cmd == "mfma":
        print(mfma_busy(sys.argv[2]))
    elif cmd == "pmcall":
        print(pmc_all(sys.argv[2]))
    elif cmd == "stats":
        print(kernel_stats(sys.argv[2]))
    else:
        fe, wr = pmc(sys.argv[2], "FETCH_SIZE"), pmc(sys.argv[3], "WRITE_SIZE")
        print(f"{'kernel':80s} {'launches':>8s} {'FETCH_SIZE/launch':>18s} {'WRITE_SIZE/launch':>18s} {'avg_us(profiled)':>16s}")
        for k in sorted(fe, key=lambda k: -fe[k][2])[:12]:
            w = wr.get(k, [1, 0.0, 0.0])
            print(f"{k:80s} {fe[k][0]:8d} {fe[k][1] / fe[k][0]:18.1f} {w[1] / max(w[0], 1):18.1f} {fe[k][2] / fe[k][0]:16.1f}")
