"""Average the per-dispatch counters of rocprofv3 --pmc passes per (kernel symbol, grid): MFMA busy fraction, wait / active
fractions of the wave cycles, effective clock, LDS conflicts, HBM-side bytes (gfx950 FETCH_SIZE correction), L2 hit rate.
    python tools/pmc_summary.py <kernel-name substrings, comma separated> <counter_collection.csv> ..."""
import collections
import csv
import re
import sys

pats = sys.argv[1].split(",")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
order = []
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if not any(p in n for p in pats):
            continue
        short = re.sub(r"void ss::|ss::bf16_t, ", "", n.split("(")[0])[:80]
        key = (short, r.get("Grid_Size") or r.get("Grid_Size_X"))
        if key not in order:
            order.append(key)
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in order:
    d = agg[key]
    print(key)
    for c, v in sorted(d.items()):
        print("    %-28s %16.0f   (n=%d)" % (c, sum(v) / len(v), len(v)))
    g = lambda c: (sum(d[c]) / len(d[c])) if c in d else None
    wc, mf, busy = g("SQ_WAVE_CYCLES"), g("SQ_VALU_MFMA_BUSY_CYCLES"), g("SQ_BUSY_CYCLES")
    if busy and mf:
        # SQ_BUSY_CYCLES is summed over the 32 shader engines; 1024 SIMDs; MFMA_BUSY counts cycles (not quad-cycles)
        print("    -> MFMA busy fraction ~ %.3f (MFMA_BUSY / (BUSY/32 * 1024))" % (mf / (busy / 32.0 * 1024.0)))
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_INST_CYCLES_VMEM",
                  "SQ_WAIT_INST_LDS"):
            if g(c) is not None:
                print("    -> %-22s / WAVE_CYCLES = %.3f" % (c, g(c) / wc))
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        print("    -> LDS bank conflict / idx active = %.3f" % (g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        print("    -> HBM-side bytes per launch ~ %.1f MB (2*FETCH_SIZE KiB + WRITE_SIZE KiB; gfx950 FETCH correction)" %
              ((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e6))
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        print("    -> L2 hit rate %.3f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
