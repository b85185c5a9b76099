"""Condense a tools/profile.sh output directory into one JSON: per-kernel average duration from the
kernel-stats CSV and per-launch counter averages (dominant kernel only) from the --pmc passes."""
import csv, glob, json, os, sys

root = sys.argv[1]
res = {"kernels": [], "counters": {}}
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        res["kernels"].append({"name": r["Name"][:100], "calls": int(r["Calls"]),
                               "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])})
res["kernels"].sort(key=lambda k: -k["pct"])
dom = res["kernels"][0]["name"].split("(")[0] if res["kernels"] else None
res["dominant"] = dom
# per-launch durations of the dominant kernel (the first launch of a process, or one after an idle gap, runs at a
# lower clock: the median is what the HIP-event timings of the same program report)
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_trace.csv"), recursive=True):
    d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f))
               if dom and r["Kernel_Name"].startswith(dom))
    if d:
        res["dominant_launch_ns"] = {"launches": len(d), "min": d[0], "median": d[len(d) // 2] if len(d) % 2 else
                                     (d[len(d) // 2 - 1] + d[len(d) // 2]) / 2, "max": d[-1]}
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        if dom and not r["Kernel_Name"].startswith(dom):
            continue
        k = (r["Counter_Name"], r["Dispatch_Id"])
        acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
    per = {}
    for (c, d), v in acc.items():
        per.setdefault(c, []).append(v)
    for c, v in per.items():
        res["counters"][c] = {"per_launch_avg": sum(v) / len(v), "launches": len(v)}
c = res["counters"]
if "FETCH_SIZE" in c:   # KB units; gfx950 FETCH_SIZE counts 64 B as 32 B -> x2 (MI355X_MICROARCH.md, HBM section)
    res["hbm_read_bytes_per_launch"] = c["FETCH_SIZE"]["per_launch_avg"] * 1024 * 2
if "WRITE_SIZE" in c:
    res["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"]["per_launch_avg"] * 1024
print(json.dumps(res, indent=1))
