"""rocprofv3 counter CSVs of tools/pmc_round5.sh -> profiles/round5_pmc_summary.json (per kernel: launches, mean duration, MFMA busy =
SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024 SIMDs), VALU-active share, effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration,
HBM-side bytes per launch = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 — the same derivations as tools/summarize_round4.py)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5pmc"
out = {}
for wl in ("attn", "split", "gemv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for d in sorted(glob.glob(os.path.join(root, wl + "_*"))):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                name = r["Kernel_Name"]
                if "ss::" not in name:
                    continue
                key = name.split("(")[0].replace("void ss::", "")[:70] + " grid=" + str(r.get("Grid_Size", ""))
                agg[key][r["Counter_Name"]][0] += 1
                agg[key][r["Counter_Name"]][1] += float(r["Counter_Value"])
                if r.get("Start_Timestamp") and r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE"):
                    dur[key][0] += 1
                    dur[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    res = {}
    for key, cs in agg.items():
        g = lambda n: (cs[n][1] / cs[n][0]) if n in cs and cs[n][0] else None   # noqa: E731
        rec = {"launches_per_pass": max(v[0] for v in cs.values())}
        us = dur[key][1] / dur[key][0] if dur[key][0] else None
        if us:
            rec["profiled_us"] = round(us, 1)
        if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
            rec["mfma_busy_frac"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("SQ_BUSY_CYCLES") / 32.0 * 1024.0), 4)
        if g("SQ_ACTIVE_INST_VALU") and g("SQ_WAVE_CYCLES"):
            rec["sq_active_inst_valu_per_wave_cycle"] = round(g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), 4)
        if g("GRBM_GUI_ACTIVE") and us:
            rec["effective_clock_ghz"] = round(g("GRBM_GUI_ACTIVE") / 8.0 / (us * 1e3), 3)
        if g("FETCH_SIZE") is not None:
            rec["hbm_bytes_per_launch"] = int(2 * g("FETCH_SIZE") * 1024 + (g("WRITE_SIZE") or 0.0) * 1024)
        res[key] = rec
    out[wl] = res
out["method"] = ("tools/pmc_round5.sh: every counter set in its own `rocprofv3 --pmc ... --kernel-trace` pass over tools/pmc5_workloads.py; "
                 "derivations as in profiles/round4_pmc_summary.json")
json.dump(out, open("gpurun_out/round5_pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
