"""Summarise a rocprofv3 --pmc counter_collection.csv for kernels matching a substring."""
import csv, collections, sys, glob
path = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
pat = sys.argv[2]
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if pat in r["Kernel_Name"]:
        agg[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
if not agg:
    print("no kernel matching", pat); sys.exit(0)
d, v = list(agg.items())[-1]
print("dispatch", d, "of", len(agg))
for k, x in sorted(v.items()):
    print("  %-28s %.4g" % (k, x))
wc = v.get("SQ_WAVE_CYCLES")
if wc:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM"):
        if k in v: print("  %s / WAVE_CYCLES = %.3f" % (k, v[k] / wc))
if v.get("SQ_LDS_IDX_ACTIVE"):
    print("  LDS bank-conflict cycles / LDS active = %.3f" % (v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]))
