import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "gemm_glds" not in n and "flash_attn2" not in n:
            continue
        key = (n.split("(")[0].replace("void ss::", "")[:48], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, d in agg.items():
    print(key)
    for c, v in sorted(d.items()):
        print("    %-28s %14.0f" % (c, sum(v) / len(v)))
