"""Round-3 profile summaries -> profiles/ (run on the GPU box by tools/gpu_r3_c.sh, inputs under gpurun_out/r3/): the shipped
schedule — image-token block decode ON, prompts and blocks of the lock-step stories as stacked forwards.

  r3/stats_overlap   rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch1`
  r3/stats_serial    the same with --no-overlap (MLLM half and render back to back: isolated kernel durations)
  r3/fetch, r3/write rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (+ --kernel-trace only) of `bench.py --steps 1 --warmup 0 ...`
  r3/k_<set>         the counter passes of tools/pmc_kernels.py (dominant MFMA kernels, cases separated by a marker kernel)
"""
import collections
import csv
import glob
import json
import os
import sys

R = os.path.join("gpurun_out", "r3")
OUT = sys.argv[1] if len(sys.argv) > 1 else os.path.join("gpurun_out", "summary")
os.makedirs(OUT, exist_ok=True)
SPG = 4


def norm(k):
    return k.split("(")[0].replace("void ", "").replace("ss::", "").replace(" ", "")


def stats(tag):
    f = glob.glob(os.path.join(R, tag, "**", "*kernel_stats.csv"), recursive=True)
    rows = []
    if f:
        for r in csv.DictReader(open(f[0])):
            rows.append({"kernel": norm(r["Name"])[:140], "calls": int(r["Calls"]), "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3),
                         "avg_us": round(float(r["AverageNs"]) / 1e3, 3), "pct": float(r["Percentage"])})
    return rows


def write_stats_csv(rows, path):
    with open(path, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct"])
        for r in rows[:60]:
            w.writerow([r["kernel"], r["calls"], r["total_ms"], r["avg_us"], r["pct"]])


def counters(tag):
    """kernel -> counter -> (mean, n) over dispatches"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(R, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[norm(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def per_case(tag):
    """pmc_kernels.py passes: split the dispatch sequence at the marker kernel -> list of {counter: mean} per case."""
    cases = []
    for f in glob.glob(os.path.join(R, tag, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        if rows and "Dispatch_Id" in rows[0]:
            rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        cur = None
        last_marker = None
        seq = []
        for r in rows:
            k = norm(r["Kernel_Name"])
            if "softmax_rows" in k:
                if r.get("Dispatch_Id") != last_marker:      # one row per counter for the same dispatch
                    cur = collections.defaultdict(list)
                    seq.append(cur)
                    last_marker = r.get("Dispatch_Id")
                continue
            if cur is None or not ("gemm_sp_kernel" in k or "flash_attn" in k or "gemm_glds" in k):
                continue
            cur[r["Counter_Name"]].append(float(r["Counter_Value"]))
            cur["_kernel"] = k
        cases.append(seq)
    merged = []
    for seq in cases:
        for i, c in enumerate(seq):
            while len(merged) <= i:
                merged.append({})
            for name, v in c.items():
                merged[i][name] = v if name == "_kernel" else sum(v[1:] or v) / max(len(v[1:] or v), 1)   # first launch = cold instruction cache
    return merged


res = {"stories_per_gpu": SPG}
ov, se = stats("stats_overlap"), stats("stats_serial")
if ov:
    write_stats_csv(ov, os.path.join(OUT, "round3_bench_kernel_stats.csv"))
if se:
    write_stats_csv(se, os.path.join(OUT, "round3_bench_no_overlap_kernel_stats.csv"))
res["gemv_avg_us_under_render"] = {r["kernel"]: r["avg_us"] for r in ov if r["kernel"].startswith("gemv")}
res["gemv_avg_us_isolated"] = {r["kernel"]: r["avg_us"] for r in se if r["kernel"].startswith("gemv")}
fe, wr = counters("fetch"), counters("write")
gem = {}
for k in fe:
    if k.startswith("gemv") and "FETCH_SIZE" in fe[k]:
        f = sum(fe[k]["FETCH_SIZE"]) / len(fe[k]["FETCH_SIZE"])
        w = sum(wr[k]["WRITE_SIZE"]) / len(wr[k]["WRITE_SIZE"]) if k in wr and "WRITE_SIZE" in wr[k] else 0.0
        gem[k] = {"hbm_bytes_per_launch": round(2 * f * 1024 + w * 1024), "fetch_bytes_per_launch_corrected_x2": round(2 * f * 1024),
                  "write_bytes_per_launch": round(w * 1024), "dispatches": len(fe[k]["FETCH_SIZE"])}
res["gemv_hbm_traffic"] = gem
# per-kernel counter cases of the dominant MFMA kernels: those kernels did not change in round 3 -> carried over from the
# round-2 collection (tools/pmc_kernels.py, profiles/round2_pmc_summary.json), marked as such
try:
    r2 = json.load(open(os.path.join("profiles", "round2_pmc_summary.json")))
    res["gemm_hbm_traffic"] = r2.get("gemm_hbm_traffic", {})
    res["gemm_hbm_traffic_source"] = "profiles/round2_pmc_summary.json (kernels unchanged since that collection)"
except Exception:
    res["gemm_hbm_traffic"] = {}
res["note"] = ("rocprofv3 on MI355X (tools/gpu_r3_c.sh; image-token block decode ON = the shipped schedule): every counter set is its own pass with --kernel-trace only; "
               "HBM bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md: FETCH_SIZE is KiB and counts half of a "
               "wide coalesced read stream on gfx950); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES/32 * 1024 SIMDs); "
               "per-kernel cases of tools/pmc_kernels.py average launches 2.. of 4 (rotating weights). gemv_avg_us_under_render = "
               "kernel-trace average with the SDXL render running on the other stream (shipped schedule); "
               "gemv_avg_us_isolated = the same command with --no-overlap.")
json.dump(res, open(os.path.join(OUT, "round3_pmc_summary.json"), "w"), indent=1)
print(json.dumps(res)[:1500])
