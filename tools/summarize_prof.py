"""Summarise rocprofv3 CSV outputs (kernel stats + optional PMC pass) into small JSON files for profiles/."""
import csv
import json
import os
import sys
from collections import defaultdict

out_dir, tag = sys.argv[1], sys.argv[2]
res = {}
stats = os.path.join("gpurun_out", "prof", "bench_kernel_stats.csv")
if os.path.exists(stats):
    rows = []
    for r in csv.DictReader(open(stats)):
        if r["Name"].startswith(("void ss::", "ss::")):
            rows.append({"kernel": r["Name"].split("(")[0][:120], "calls": int(r["Calls"]),
                         "total_ms": float(r["TotalDurationNs"]) / 1e6, "avg_us": float(r["AverageNs"]) / 1e3,
                         "pct": float(r["Percentage"])})
    res["kernel_stats"] = rows
pmc = []
for root, _, files in os.walk(os.path.join("gpurun_out", "pmc")):
    for f in files:
        if f.endswith("counter_collection.csv"):
            pmc.append(os.path.join(root, f))
if pmc:
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for path in pmc:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0][:120]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
    summ = {}
    for k in agg:
        if "ss::" not in k:
            continue
        summ[k] = {c: agg[k][c] / cnt[k][c] for c in agg[k]}
        summ[k]["dispatches"] = max(cnt[k].values())
    res["pmc_avg_per_dispatch"] = summ
    # HBM bytes per launch of every decode-GEMV kernel symbol.  MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are
    # in KiB, and FETCH_SIZE reports 1/2 of a wide coalesced read stream on gfx950 (hence the x2).
    gem = {}
    for k, v in summ.items():
        if k.startswith("void ss::gemv") and "FETCH_SIZE" in v:
            norm = k.replace("void ", "").replace("ss::", "").replace(" ", "")
            gem[norm] = {"hbm_bytes_per_launch": round(v["FETCH_SIZE"] * 1024 * 2 + v.get("WRITE_SIZE", 0.0) * 1024),
                         "fetch_bytes_per_launch_corrected_x2": round(v["FETCH_SIZE"] * 1024 * 2),
                         "write_bytes_per_launch": round(v.get("WRITE_SIZE", 0.0) * 1024),
                         "dispatches": v["dispatches"]}
    res["gemv_hbm_traffic"] = gem
os.makedirs(out_dir, exist_ok=True)
json.dump(res, open(os.path.join(out_dir, tag + ".json"), "w"), indent=1)
print(json.dumps(res)[:600])
