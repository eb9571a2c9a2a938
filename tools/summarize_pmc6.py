"""Round-6 PMC summary of the dominant MFMA kernels on the SHIPPED tile table.

    python tools/summarize_pmc6.py --cases          -> the case list of tools/pmc_round6.sh (cfg / XCD group read from the table)
    python tools/summarize_pmc6.py gpurun_out/r6pmc -> gpurun_out/round6_pmc_summary.json

Per case: kernel, profiled duration, effective clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), MFMA busy =
SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES / 32 * 1024 SIMDs), wait / active shares of the wave cycles, LDS conflicts, L2 hit rate and
HBM-side bytes per launch = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB (the derivations of profiles/round4_pmc_summary.json, as
MI355X_MICROARCH.md prescribes for gfx950).  The GEMV / attention records of the round-4 / round-5 files are carried over unchanged
(those kernels did not change) so that bench.py finds one file."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "seed-story_amd", "seedstory", "tune_gfx950.json")

# name, M, N, K, epi (16 = GEGLU), residual, conv geometry (B, H, W, Cin, Cout) or None
CASES = [
    ("ff1_16384x10240x1280_geglu", 16384, 10240, 1280, 16, 0, None),
    ("qkv_16384x3840x1280", 16384, 3840, 1280, 0, 0, None),
    ("n1280res_16384x1280x1280", 16384, 1280, 1280, 0, 1, None),
    ("ff2res_16384x1280x5120", 16384, 1280, 5120, 0, 1, None),
    ("ff1_65536x5120x640_geglu", 65536, 5120, 640, 16, 0, None),
    ("conv3x3_16x32x32_1280to1280", 16384, 1280, 11520, 0, 0, (16, 32, 32, 1280, 1280)),
    ("conv3x3_16x64x64_640to640", 65536, 640, 5760, 0, 0, (16, 64, 64, 640, 640)),
    ("gemm_8192cubed", 8192, 8192, 8192, 0, 0, None),
]


def table_entry(M, N, K, conv):
    for r in json.load(open(TABLE))["entries"]:
        if r[0] == 1 and r[1] == M and r[2] == N and r[3] == K and ((conv is None and r[4] == 0) or (conv is not None and r[4] == conv[3] and r[6] == conv[1])):
            return r[8], r[9]
    return (54, 8) if conv is None else (56, 8)


def algorithmic_bytes(M, N, K, epi, res, conv):
    a = (conv[0] * conv[1] * conv[2] * conv[3]) if conv else M * K
    out = M * N // 2 if epi & 16 else M * N
    return 2 * (a + N * K + out + (M * N if res else 0))


def cases():
    for name, M, N, K, epi, res, conv in CASES:
        cfg, swz = table_entry(M, N, K, conv)
        if conv:
            arg = "c%d,%d,%d,%d,%d,1,0,1:%d/%d" % (conv + (cfg, swz))
        else:
            arg = "%d,%d,%d,%d,%d:%d/%d" % (M, N, K, epi, res, cfg, swz)
        print(name, arg)


def main(root):
    out = {}
    tab = {}
    for line in open(os.path.join(root, "cases.txt")):
        n, a = line.split()
        tab[n] = a
    for name, M, N, K, epi, res, conv in CASES:
        cs = collections.defaultdict(list)
        durs, kern = [], None
        for d in sorted(glob.glob(os.path.join(root, name + "_*"))):
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    if "gemm_pp_kernel" not in k and "gemm_sp_kernel" not in k and "gemm_glds" not in k:
                        continue
                    kern = k.split("(")[0].replace("void ss::", "").replace("ss::", "")
                    cs[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",) and r.get("Start_Timestamp"):
                        durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if not cs:
            continue
        # the first launch of a pass is the correctness launch of the harness (cold): drop it where there are several
        g = lambda n: (sum(cs[n][1:]) / len(cs[n][1:]) if len(cs[n]) > 1 else (cs[n][0] if cs[n] else None))   # noqa: E731
        us = sum(durs[1:]) / len(durs[1:]) if len(durs) > 1 else (durs[0] if durs else None)
        rec = {"kernel": kern, "cfg_swz": tab[name].split(":")[1], "launches_per_pass": len(cs.get("GRBM_GUI_ACTIVE", []))}
        if us:
            rec["profiled_us"] = round(us, 1)
            rec["profiled_tflops"] = round(2.0 * M * N * K / (us * 1e-6) / 1e12, 1)
        if g("SQ_BUSY_CYCLES") and g("SQ_VALU_MFMA_BUSY_CYCLES"):
            rec["mfma_busy_frac"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("SQ_BUSY_CYCLES") / 32.0 * 1024.0), 4)
        if g("SQ_WAVE_CYCLES"):
            for c, nm in (("SQ_WAIT_ANY", "sq_wait_any_per_wave_cycle"), ("SQ_WAIT_INST_ANY", "sq_wait_inst_any_per_wave_cycle"),
                          ("SQ_ACTIVE_INST_ANY", "sq_active_inst_any_per_wave_cycle"), ("SQ_ACTIVE_INST_VALU", "sq_active_inst_valu_per_wave_cycle")):
                if g(c) is not None:
                    rec[nm] = round(g(c) / g("SQ_WAVE_CYCLES"), 4)
        if g("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_per_active"] = round((g("SQ_LDS_BANK_CONFLICT") or 0.0) / g("SQ_LDS_IDX_ACTIVE"), 4)
        if g("GRBM_GUI_ACTIVE") and us:
            rec["effective_clock_ghz"] = round(g("GRBM_GUI_ACTIVE") / 8.0 / (us * 1e3), 3)
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
            rec["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
        if g("FETCH_SIZE") is not None:
            rec["hbm_bytes_per_launch"] = int(2 * g("FETCH_SIZE") * 1024 + (g("WRITE_SIZE") or 0.0) * 1024)
            rec["algorithmic_bytes_per_launch"] = algorithmic_bytes(M, N, K, epi, res, conv)
            rec["overfetch_ratio"] = round(rec["hbm_bytes_per_launch"] / rec["algorithmic_bytes_per_launch"], 2)
        out[name] = rec
    base = {}
    for prev in ("round4_pmc_summary.json",):
        pth = os.path.join(ROOT, "profiles", prev)
        if os.path.exists(pth):
            base = json.load(open(pth))
    base["gemm_hbm_traffic_round4"] = base.get("gemm_hbm_traffic")
    base["gemm_hbm_traffic"] = out
    base["tile_table_sha16"] = hashlib.sha256(open(TABLE, "rb").read()).hexdigest()[:16]
    base["gemm_pmc_run"] = ("round 6: tools/pmc_round6.sh — every counter set in its own `rocprofv3 --pmc ... --kernel-trace` pass over tools/gemm_ubench "
                            "(UBENCH_PMC: 3 launches per case over rotating weights, the first dropped), each case on the tile the shipped table holds "
                            "for its shape; GEMV / attention records carried over from round 4 / 5 (kernels unchanged)")
    json.dump(base, open(os.path.join("gpurun_out", "round6_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--cases":
        cases()
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r6pmc")
