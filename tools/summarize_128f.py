"""Joins the rocprofv3 CSVs of `tools/prof_128f.py` with its calls.json: every row of the table is
ONE C-ABI call at 128 frames per launch -- its kernels, their summed duration (mean over the measured passes), executed
and nominal flops, algorithmic and counter bytes, and the two roofline fractions -- re-derivable from the CSVs by
division.        python tools/summarize_128f.py RUN_DIR > profiles/r05_rooflines_128f.txt
RUN_DIR holds calls.json, trace/ (--kernel-trace --stats) and optionally pmc_*/ (one --pmc pass each)."""
import collections
import csv
import glob
import json
import os
import sys

HBM, MFMA16, VALU32 = 8000.0, 2500.0, 157.3       # GB/s, TFLOP/s dense bf16/fp16, TFLOP/s fp32 (MI355X_MICROARCH.md)


def short(n):
    for p in ("void (anonymous namespace)::", "(anonymous namespace)::", "void "):
        n = n.replace(p, "")
    return n.split("(")[0][:70]


def passes_of(rows, name_key, marker, npass):
    """rows (in dispatch order) -> the last `npass` passes, each a list of call groups (lists of rows).  A pass starts after
    a run of >= 5 marker rows; single markers separate the calls."""
    out, cur_pass, cur, run = [], None, None, 0
    for r in rows:
        if marker in r[name_key]:
            run += 1
            if run >= 5:
                if cur_pass is not None and cur_pass and run == 5:
                    out.append(cur_pass)
                cur_pass, cur = [], None
            elif cur_pass is not None:
                cur = []
                cur_pass.append(cur)
            continue
        if run >= 5 and cur_pass is not None:
            cur = []
            cur_pass.append(cur)
        run = 0
        if cur is not None:
            cur.append(r)
    if cur_pass:
        out.append(cur_pass)
    out = [[g for g in p if g] for p in out]
    # what follows a pass's closing marker (the host reading the plan headers back: runtime copy kernels) is no call's
    out = [[g for g in p if not all("__amd_rocclr" in r[name_key] for r in g)] for p in out]
    return out[-npass:]


def main():
    root = sys.argv[1]
    meta = json.load(open(os.path.join(root, "calls.json")))
    calls, npass, marker = meta["calls_per_pass"], meta["passes"], meta["marker_kernel"]
    f = glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    ps = passes_of(rows, "Kernel_Name", marker, npass)
    assert ps and all(len(p) == len(calls) for p in ps), ([len(p) for p in ps], len(calls))
    # per call: mean over passes of the summed kernel durations; kernel names
    dur = [0.0] * len(calls)
    names = [collections.Counter() for _ in calls]
    per_kernel = collections.defaultdict(list)
    for p in ps:
        for i, g in enumerate(p):
            for r in g:
                us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                dur[i] += us / len(ps)
                names[i][short(r["Kernel_Name"])] += 1
                per_kernel[short(r["Kernel_Name"])].append(us)
    # counters: FETCH_SIZE / WRITE_SIZE (KiB), SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE per call
    ctr = [collections.defaultdict(float) for _ in calls]
    kctr = collections.defaultdict(lambda: collections.defaultdict(float))      # kernel -> counter -> sum over launches
    klaunch = collections.defaultdict(lambda: collections.defaultdict(int))     # kernel -> counter -> launches seen
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        crow = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        seen, disp = set(), []
        for r in crow:                                   # one representative row per dispatch for the marker cut ...
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                disp.append(r)
        by_disp = collections.defaultdict(list)
        for r in crow:
            by_disp[r["Dispatch_Id"]].append(r)
        cps = passes_of(disp, "Kernel_Name", marker, npass)
        if not cps or any(len(p) != len(calls) for p in cps):
            print("# counter file %s does not align with calls.json, skipped" % os.path.relpath(f, root))
            continue
        for p in cps:
            for i, g in enumerate(p):
                for d in g:
                    seen_c = set()
                    for r in by_disp[d["Dispatch_Id"]]:
                        ctr[i][r["Counter_Name"]] += float(r["Counter_Value"]) / len(cps)
                        k = short(r["Kernel_Name"])
                        kctr[k][r["Counter_Name"]] += float(r["Counter_Value"])
                        if r["Counter_Name"] not in seen_c:
                            seen_c.add(r["Counter_Name"])
                            klaunch[k][r["Counter_Name"]] += 1
    print("# one row = one C-ABI call of a backbone pass over %d frames (data=%s); us = sum of its kernels' durations, mean of %d passes"
          % (meta["frames_per_launch"], meta["data"], len(ps)))
    print("# peaks: HBM %.0f GB/s, MFMA bf16/fp16 dense %.0f TFLOP/s, fp32 VALU %.1f TFLOP/s; counter bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB, gfx950 note of MI355X_MICROARCH.md)" % (HBM, MFMA16, VALU32))
    hdr = "%-24s %-62s %9s %10s %10s %7s %7s %9s %9s %7s %7s  %s" % ("call", "label", "us", "GF nominal", "GF exec", "f.nom", "f.exec", "MB alg", "MB ctr", "f.hbm", "mfmabusy", "kernels")
    print(hdr)
    tot = collections.defaultdict(float)
    for i, c in enumerate(calls):
        us = dur[i]
        gf, gfe = c["flops"] / 1e9, c.get("flops_executed", 0.0) / 1e9
        mb = c["bytes"] / 1e6
        mbc = (2.0 * ctr[i].get("FETCH_SIZE", 0.0) + ctr[i].get("WRITE_SIZE", 0.0)) * 1024 / 1e6 if ("FETCH_SIZE" in ctr[i] or "WRITE_SIZE" in ctr[i]) else None
        peak = VALU32 if ("fps" in c["call"] or "square_dist" in c["call"]) else MFMA16
        fnom = gf / us * 1e3 / peak if us > 0 and gf else None     # GFLOP / us = 1000 TFLOP/s
        fex = gfe / us * 1e3 / peak if us > 0 and gfe else None
        fh = (mb / us * 1000.0) / HBM if us > 0 else None          # MB / us = 1000 GB/s
        busy = ctr[i]["SQ_VALU_MFMA_BUSY_CYCLES"] / (ctr[i]["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if ctr[i].get("GRBM_GUI_ACTIVE") else None
        fmt = lambda v, f="%.3f": "-" if v is None else f % v
        print("%-24s %-62s %9.1f %10.1f %10s %7s %7s %9.1f %9s %7s %7s  %s" %
              (c["call"][:24], c["label"][:62], us, gf, fmt(gfe if gfe else None, "%.1f"), fmt(fnom), fmt(fex), mb, fmt(mbc, "%.1f"), fmt(fh), fmt(busy),
               ", ".join("%s x%d" % (k, v // len(ps)) for k, v in names[i].items())))
        tot["us"] += us
        if c["call"] in ("sa_group_mlp_max_layer", "sa_group_mlp_max", "sa_group_mlp_plan2", "sa_group_mlp_plan"):
            tot["mlp_us"] += us; tot["mlp_gf"] += gf; tot["mlp_gfe"] += gfe
            tot["mlp_busy"] += ctr[i].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); tot["mlp_cap"] += ctr[i].get("GRBM_GUI_ACTIVE", 0.0) / 8.0 * 1024.0
            tot["rows_eval"] += c.get("rows_evaluated", 0); tot["rows_dist"] += c.get("rows_distinct", 0)
    print("# pass total %.1f us (single stream, kernels only; the layer-1 D-FPS holds 128 of the 256 CUs)" % tot["us"])
    if tot["mlp_us"]:
        print("# grouped MLP incl. plans: %.1f us, nominal %.1f GFLOP -> %.3f of the bf16 peak, executed %.1f GFLOP -> %.3f; rows evaluated / distinct %.2f%s"
              % (tot["mlp_us"], tot["mlp_gf"], tot["mlp_gf"] / tot["mlp_us"] * 1e3 / MFMA16, tot["mlp_gfe"], tot["mlp_gfe"] / tot["mlp_us"] * 1e3 / MFMA16,
                 tot["rows_eval"] / max(tot["rows_dist"], 1), ("; MFMA busy %.3f" % (tot["mlp_busy"] / tot["mlp_cap"])) if tot["mlp_cap"] else ""))
    # per kernel (bench.py reads this for roofline.traffic / roofline_grouped_mlp.pmc): HBM bytes per launch = 2 x FETCH_SIZE +
    # WRITE_SIZE (KiB; gfx950 note of MI355X_MICROARCH.md), MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
    tr = {}
    for k, cs in kctr.items():
        e = {}
        per = lambda c: cs[c] / max(klaunch[k][c], 1)
        if "FETCH_SIZE" in cs or "WRITE_SIZE" in cs:
            e["FETCH_SIZE_KiB_per_launch"] = per("FETCH_SIZE") if "FETCH_SIZE" in cs else 0.0
            e["WRITE_SIZE_KiB_per_launch"] = per("WRITE_SIZE") if "WRITE_SIZE" in cs else 0.0
            e["hbm_bytes_per_launch"] = int(1024 * (2.0 * e["FETCH_SIZE_KiB_per_launch"] + e["WRITE_SIZE_KiB_per_launch"]))
        if cs.get("GRBM_GUI_ACTIVE", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
            e["mfma_busy_cycles_per_launch"] = per("SQ_VALU_MFMA_BUSY_CYCLES")
            e["gui_active_cycles_per_launch"] = per("GRBM_GUI_ACTIVE")
            e["mfma_util"] = round(cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (cs["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        for c in ("SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY"):
            if c in cs:
                e[c + "_per_launch"] = per(c)
        if e and marker not in k:
            tr[k] = e
    json.dump(tr, open(os.path.join(root, "traffic.json"), "w"), indent=1, sort_keys=True)
    with open(os.path.join(root, "kernel_stats_128f.csv"), "w") as o:
        o.write("kernel,launches_per_pass,avg_us,min_us,max_us,total_us_per_pass\n")
        for k, v in sorted(per_kernel.items(), key=lambda kv: -sum(kv[1])):
            o.write("%s,%.2f,%.2f,%.2f,%.2f,%.2f\n" % (k.replace(",", ";"), len(v) / len(ps), sum(v) / len(v), min(v), max(v), sum(v) / len(ps)))


main()
