"""Round-4 profile summaries (run on the GPU box by tools/gpu_round4.sh; inputs under gpurun_out/r4/) -> gpurun_out/summary/:
  stats : rocprofv3 --kernel-trace --stats of the bench command, shipped schedule and --no-overlap -> round4_bench*_kernel_stats.csv
  pmc   : round4_pmc_summary.json — per case of the torch-free micro-benchmark (dominant MFMA kernels ON THE SHIPPED TILE TABLE):
          kernel, profiled duration, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA busy fraction, wait / active
          fractions, LDS conflicts, HBM-side bytes (2 x FETCH_SIZE KiB + WRITE_SIZE KiB: MI355X_MICROARCH.md), L2 hit rate; the two
          self-attention shapes; decode-GEMV traffic of `bench.py --mllm-only`; and the sha256 of seedstory/tune_gfx950.json the
          counters belong to (bench.py refuses a `traffic` figure whose table hash differs from the shipped one)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

R = os.path.join("gpurun_out", "r4")
OUT = os.path.join("gpurun_out", "summary")
os.makedirs(OUT, exist_ok=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def norm(k):
    return k.split("(")[0].replace("void ", "").replace("ss::", "").replace(" ", "")


def table_sha16():
    return hashlib.sha256(open(os.path.join(ROOT, "seed-story_amd", "seedstory", "tune_gfx950.json"), "rb").read()).hexdigest()[:16]


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


def dispatches(tag, want):
    """One pass: ordered list of (kernel, {counter: value}, duration_us) of the dispatches whose kernel name contains `want`."""
    cc = glob.glob(os.path.join(R, tag, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(R, tag, "**", "*kernel_trace.csv"), recursive=True)
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by = collections.OrderedDict()
    if cc:
        rows = sorted(csv.DictReader(open(cc[0])), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            k = norm(r["Kernel_Name"])
            if not any(w in k for w in want):
                continue
            d = by.setdefault(r["Dispatch_Id"], {"kernel": k, "c": {}, "us": dur.get(r["Dispatch_Id"])})
            d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    return list(by.values())


def mean(v):
    return sum(v) / len(v) if v else None


def derive(c, us):
    out = {}
    g = c.get
    if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
        out["mfma_busy_frac"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("SQ_BUSY_CYCLES") / 32.0 * 1024.0), 4)
    if g("SQ_WAVE_CYCLES"):
        for n, k in (("sq_wait_any_per_wave_cycle", "SQ_WAIT_ANY"), ("sq_wait_inst_any_per_wave_cycle", "SQ_WAIT_INST_ANY"),
                     ("sq_active_inst_any_per_wave_cycle", "SQ_ACTIVE_INST_ANY"), ("sq_active_inst_valu_per_wave_cycle", "SQ_ACTIVE_INST_VALU")):
            if g(k) is not None:
                out[n] = round(g(k) / g("SQ_WAVE_CYCLES"), 4)
    if g("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_per_active"] = round((g("SQ_LDS_BANK_CONFLICT") or 0.0) / g("SQ_LDS_IDX_ACTIVE"), 4)
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        out["hbm_bytes_per_launch"] = round((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024)
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        out["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
    if g("GRBM_GUI_ACTIVE") and us:
        out["profiled_us"] = round(us, 1)
        out["effective_clock_ghz"] = round(g("GRBM_GUI_ACTIVE") / 8.0 / (us * 1e3), 3)
        out["grbm_cycles_per_xcd"] = round(g("GRBM_GUI_ACTIVE") / 8.0)
    return out


def cases_from_ubench():
    """gemm_ubench in PMC mode launches, per case with c cfgs: 1 reference + c checks + 3 per cfg (these are used)."""
    spec = open(os.path.join(R, "cases.txt")).read().split()
    names = ["ff1_16384x10240x1280_geglu", "ff1_8192x10240x1280_geglu", "qkv_8192x3840x1280", "n1280res_8192x1280x1280", "ff2res_8192x1280x5120",
             "conv3x3_1280to1280_32x32_b8_rowvec", "gemm_8192cubed"]
    merged = collections.OrderedDict()
    for tag in sorted(glob.glob(os.path.join(R, "k_*"))):
        seq = dispatches(os.path.basename(tag), ("gemm_sp_kernel", "gemm_w4_kernel", "gemm_glds", "gemm_kernel"))
        pos = 0
        for name, sp in zip(names, spec):
            cfgs = sp.split(":")[1].split(",")
            n = 1 + 4 * len(cfgs)
            chunk = seq[pos:pos + n]
            pos += n
            for ci, cfg in enumerate(cfgs):
                ds = chunk[1 + len(cfgs) + 3 * ci: 1 + len(cfgs) + 3 * ci + 3]
                if not ds:
                    continue
                key = name if len(cfgs) == 1 else "%s_cfg%s" % (name, cfg.split("/")[0])
                e = merged.setdefault(key, {"kernel": ds[-1]["kernel"], "cfg": cfg, "c": {}, "us": []})
                for d in ds[1:] or ds:
                    for k, v in d["c"].items():
                        e["c"].setdefault(k, []).append(v)
                    if d["us"]:
                        e["us"].append(d["us"])
    out = collections.OrderedDict()
    for key, e in merged.items():
        c = {k: mean(v) for k, v in e["c"].items()}
        rec = {"kernel": e["kernel"], "cfg_swz": e["cfg"]}
        rec.update(derive(c, mean(e["us"])))
        out[key] = rec
    return out


def attention_cases():
    merged = [collections.defaultdict(list), collections.defaultdict(list)]
    uss = [[], []]
    kern = None
    for tag in sorted(glob.glob(os.path.join(R, "a_*"))):
        seq = dispatches(os.path.basename(tag), ("flash_attn",))
        for i in range(2):
            for d in seq[3 * i + 1: 3 * i + 3]:
                kern = d["kernel"]
                for k, v in d["c"].items():
                    merged[i][k].append(v)
                if d["us"]:
                    uss[i].append(d["us"])
    out = {}
    for i, name in enumerate(("self_attention_4096tok_8x10x64", "self_attention_1024tok_8x20x64")):
        if merged[i]:
            rec = {"kernel": kern}
            rec.update(derive({k: mean(v) for k, v in merged[i].items()}, mean(uss[i])))
            out[name] = rec
    return out


def counters(tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(R, tag, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[norm(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def gemv_traffic():
    fe, wr = counters("fetch"), counters("write")
    gem = {}
    for k in fe:
        if k.startswith("gemv") and "FETCH_SIZE" in fe[k]:
            f = mean(fe[k]["FETCH_SIZE"])
            w = mean(wr[k]["WRITE_SIZE"]) if k in wr and "WRITE_SIZE" in wr[k] else 0.0
            gem[k] = {"hbm_bytes_per_launch": round(2 * f * 1024 + w * 1024), "fetch_bytes_per_launch_corrected_x2": round(2 * f * 1024),
                      "write_bytes_per_launch": round(w * 1024), "dispatches": len(fe[k]["FETCH_SIZE"])}
    return gem


mode = sys.argv[1] if len(sys.argv) > 1 else "pmc"
if mode == "final":
    # the round's last call: the shipped default moved to 8 stories per GPU (MFMA-form decode GEMV, UNet batch 16) after the `pmc` /
    # `stats` records above were taken: fold the new passes INTO the committed record (profiles/round4_pmc_summary.json)
    base = json.load(open(os.path.join(ROOT, "profiles", "round4_pmc_summary.json")))
    ov = stats("stats_overlap")
    if ov:
        write_stats_csv(ov, os.path.join(OUT, "round4_bench_8stories_kernel_stats.csv"))
        base["gemv_avg_us_under_render_8_slots"] = {r["kernel"]: r["avg_us"] for r in ov if r["kernel"].startswith("gemv")}
        base["gemv_calls_under_render_8_slots"] = {r["kernel"]: r["calls"] for r in ov if r["kernel"].startswith("gemv")}
        if base.get("stories_per_gpu") == 8:
            base["gemv_avg_us_under_render"] = base["gemv_avg_us_under_render_8_slots"]
            base["gemv_calls_under_render"] = base["gemv_calls_under_render_8_slots"]
    se = stats("stats_serial")
    if se:
        write_stats_csv(se, os.path.join(OUT, "round4_bench_8stories_no_overlap_kernel_stats.csv"))
        base["gemv_avg_us_isolated_8_slots"] = {r["kernel"]: r["avg_us"] for r in se if r["kernel"].startswith("gemv")}
        base["gemv_calls_isolated_8_slots"] = {r["kernel"]: r["calls"] for r in se if r["kernel"].startswith("gemv")}
    gem = gemv_traffic()
    if gem:
        if base.get("stories_per_gpu") != 8:       # keep the 4-slot records (LDS-staged dot-product form) under their own keys
            base["gemv_4_slots"] = {"gemv_hbm_traffic": base.get("gemv_hbm_traffic"), "gemv_avg_us_under_render": base.get("gemv_avg_us_under_render"),
                                    "gemv_avg_us_isolated": base.get("gemv_avg_us_isolated")}
        base["stories_per_gpu"] = 8
        base["gemv_hbm_traffic"] = gem
        base["gemv_avg_us_under_render"] = base.get("gemv_avg_us_under_render_8_slots", {})
        base["gemv_calls_under_render"] = base.get("gemv_calls_under_render_8_slots", {})
        base["gemv_hbm_traffic_source"] = ("tools/gemv_pmc.py: the decode token's 129-launch GEMV mix at 8 slots per sweep over rotating weight "
                                           "copies (FETCH_SIZE and WRITE_SIZE in separate passes)")
        try:
            base["gemv_pmc_run"] = json.loads([l for l in open(os.path.join("gpurun_out", "r4_fetch.log")) if l.startswith('{"gemv_pmc')][-1])
        except Exception:
            pass
    g = cases_from_ubench()
    for k, v in g.items():
        if k == "ff1_16384x10240x1280_geglu":
            v["algorithmic_bytes_per_launch"] = 2 * (16384 * 1280 + 10240 * 1280 + 16384 * 5120)
            if v.get("hbm_bytes_per_launch"):
                v["overfetch_ratio"] = round(v["hbm_bytes_per_launch"] / v["algorithmic_bytes_per_launch"], 2)
        base.setdefault("gemm_hbm_traffic", {})[k] = v
    base["final_call_note"] = ("tools/gpu_round4.sh final: kernel-trace --stats of the bench command at the shipped default (8 stories per GPU), the "
                               "ff1 GEGLU GEMM of UNet batch 16 (the line's dominant kernel) and the 8-slot decode GEMV traffic were added to this "
                               "record by the round's last GPU call; the other GEMM / conv / attention rows are the earlier `pmc` stage's")
    json.dump(base, open(os.path.join(OUT, "round4_pmc_summary.json"), "w"), indent=1)
    print(json.dumps({k: base[k] for k in ("gemv_hbm_traffic", "gemv_avg_us_under_render") if k in base}))
    print(json.dumps(g))
elif mode == "stats":
    ov, se = stats("stats_overlap"), stats("stats_serial")
    if ov:
        write_stats_csv(ov, os.path.join(OUT, "round4_bench_kernel_stats.csv"))
    if se:
        write_stats_csv(se, os.path.join(OUT, "round4_bench_no_overlap_kernel_stats.csv"))
    gem = {"gemv_avg_us_under_render": {r["kernel"]: r["avg_us"] for r in ov if r["kernel"].startswith("gemv")},
           "gemv_avg_us_isolated": {r["kernel"]: r["avg_us"] for r in se if r["kernel"].startswith("gemv")}}
    json.dump(gem, open(os.path.join(OUT, "round4_gemv_kernel_trace_avg.json"), "w"), indent=1)
    print(json.dumps(gem))
else:
    slots = 8
    try:        # slots per sweep of the --mllm-only command the FETCH_SIZE pass profiled (its JSON line)
        line = [l for l in open(os.path.join("gpurun_out", "r4_fetch.log")) if l.startswith('{"metric')][-1]
        slots = int(json.loads(line)["roofline"]["mllm_decode_gemv"]["slots_per_sweep"])
    except Exception:
        pass
    res = {"stories_per_gpu": slots, "tile_table_sha16": table_sha16()}
    try:
        res.update(json.load(open(os.path.join(OUT, "round4_gemv_kernel_trace_avg.json"))))
    except Exception:
        pass
    fe, wr = counters("fetch"), counters("write")
    gem = {}
    for k in fe:
        if k.startswith("gemv") and "FETCH_SIZE" in fe[k]:
            f = mean(fe[k]["FETCH_SIZE"])
            w = mean(wr[k]["WRITE_SIZE"]) if k in wr and "WRITE_SIZE" in wr[k] else 0.0
            gem[k] = {"hbm_bytes_per_launch": round(2 * f * 1024 + w * 1024), "fetch_bytes_per_launch_corrected_x2": round(2 * f * 1024),
                      "write_bytes_per_launch": round(w * 1024), "dispatches": len(fe[k]["FETCH_SIZE"])}
    res["gemv_hbm_traffic"] = gem
    g = cases_from_ubench()
    alg = {"ff1_16384x10240x1280_geglu": 2 * (16384 * 1280 + 10240 * 1280 + 16384 * 5120),
           "ff1_8192x10240x1280_geglu": 2 * (8192 * 1280 + 10240 * 1280 + 8192 * 5120), "qkv_8192x3840x1280": 2 * (8192 * 1280 + 3840 * 1280 + 8192 * 3840),
           "n1280res_8192x1280x1280": 2 * (8192 * 1280 + 1280 * 1280 + 2 * 8192 * 1280), "ff2res_8192x1280x5120": 2 * (8192 * 5120 + 1280 * 5120 + 2 * 8192 * 1280),
           "conv3x3_1280to1280_32x32_b8_rowvec": 2 * (8192 * 1280 + 1280 * 11520 + 8192 * 1280)}
    for k, v in g.items():
        if k in alg:
            v["algorithmic_bytes_per_launch"] = alg[k]
            if v.get("hbm_bytes_per_launch"):
                v["overfetch_ratio"] = round(v["hbm_bytes_per_launch"] / alg[k], 2)
    res["gemm_hbm_traffic"] = g
    res["attention"] = attention_cases()
    res["round2_overfetch_for_comparison"] = {"ff1": 3.1, "conv3x3_1280to1280": 5.8}
    res["note"] = ("rocprofv3 on MI355X (tools/gpu_round4.sh pmc): every counter set is its own pass with --kernel-trace only; GEMM / conv cases "
                   "run the tile (cfg/xcd group) the SHIPPED table holds for the shape (tile_table_sha16), launches 2-3 of 3 over rotating "
                   "weights; HBM-side bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (MI355X_MICROARCH.md gfx950 correction; Infinity-Cache hits are "
                   "counted, so this is fabric-side traffic); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES/32 * 1024 SIMDs); "
                   "effective clock = GRBM_GUI_ACTIVE / 8 XCDs / the profiled dispatch duration of the same pass.  gemm_8192cubed: the 8-wave "
                   "16x16x32 tile (cfg 60) against the 4-wave AGPR-accumulator 32x32x16 tile (cfg 91) — fewer cycles, same wall time: the "
                   "chip clocks by its power budget.")
    json.dump(res, open(os.path.join(OUT, "round4_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(res)[:3000])
