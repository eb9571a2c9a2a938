import csv, sys, collections
f = sys.argv[1]
n_fwd = int(sys.argv[2]) if len(sys.argv) > 2 else 3
agg = collections.defaultdict(lambda: [0, 0.0])
tot = 0.0
rows_all = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows_all) if "softmax_rows_kernel" in r["Kernel_Name"]]
if marks:          # tools/unet_trace.py drops a marker kernel after its autotune pass
    rows_all = rows_all[marks[-1] + 1:]
for r in rows_all:
    name = r["Kernel_Name"]
    if "ss::" not in name:
        continue
    short = name.split("(")[0].replace("void ss::", "").replace("ss::bf16_t, ", "")[:60]
    grid = (r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"))
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    k = (short, grid)
    agg[k][0] += 1
    agg[k][1] += dur
    tot += dur
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("total kernel ms per forward: %.2f" % (tot / 1e3 / n_fwd))
for (short, grid), (c, d) in rows[:45]:
    print("%-62s grid=%-16s calls/fwd=%5.1f avg_us=%8.1f ms/fwd=%6.2f" % (short, str(grid), c / n_fwd, d / c, d / 1e3 / n_fwd))
