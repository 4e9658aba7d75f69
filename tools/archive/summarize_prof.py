"""Summarise rocprofv3 output: per-kernel average duration (kernel trace) and per-kernel PMC sums."""
import csv, glob, os, sys, collections
root = sys.argv[1]
def short(n):
    for p in ("void (anonymous namespace)::", "(anonymous namespace)::", "void "):
        n = n.replace(p, "")
    n = n.split("(")[0]
    return n[:70]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", os.path.relpath(f, root))
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print("  %-70s calls %6s  avg %10.1f us  total %10.1f us  %5s%%" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3, r["Percentage"]))
for f in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    print("== pmc", os.path.relpath(f, root))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    for k in sorted(agg, key=lambda k: -cnt[k])[:40]:
        print("  %-60s n=%4d " % (k, cnt[k]) + " ".join("%s=%.4g" % (c, v / cnt[k]) for c, v in sorted(agg[k].items())))

# per-kernel HBM traffic per launch (bytes) for bench.py's roofline.traffic: FETCH_SIZE / WRITE_SIZE are in KiB;
# on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section), so reads are
# doubled; WRITE_SIZE is taken as is (uncalibrated).
import json
tr = {}
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, "pmc_" + cname, "**", "*counter_collection.csv"), recursive=True):
        agg = collections.defaultdict(float); cnt = collections.Counter(); seen = set()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            k = short(r["Kernel_Name"])
            agg[k] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
        for k in agg:
            tr.setdefault(k, {})[cname + "_KiB_per_launch"] = agg[k] / cnt[k]
for k, d in tr.items():
    d["hbm_bytes_per_launch"] = int(1024 * (2.0 * d.get("FETCH_SIZE_KiB_per_launch", 0.0) + d.get("WRITE_SIZE_KiB_per_launch", 0.0)))
# MFMA utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES (summed over SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            continue
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); cnt[k] += 1
    for k, d in agg.items():
        if d.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
            e = tr.setdefault(k, {})
            e["mfma_busy_cycles_per_launch"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / cnt[k]
            e["gui_active_cycles_per_launch"] = d["GRBM_GUI_ACTIVE"] / cnt[k]
            e["mfma_util"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
json.dump(tr, open(os.path.join(root, "traffic.json"), "w"), indent=1, sort_keys=True)
