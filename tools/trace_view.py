"""Print a compact timeline of one fit from a rocprofv3 kernel trace CSV (start offset us, duration us, queue, name)."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last gram_kernel with lower-only (fit) marks the start of a fit: take the LAST fit in the trace
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("gram_kernel")]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 0 = last fit in the trace, 1 = the one before, ...
start = idx[-2 - 2 * back] if len(idx) >= 2 + 2 * back else idx[-1]   # fit gram then predict gram alternate
t0 = int(rows[start]["Start_Timestamp"])
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 120
short = {"leaf_signal_wait": "LEAFWAIT", "leaf_worker": "WORKER", "gemm_nt_f64_kernel": "gemm", "potrf128_kernel": "POTRF", "trtri128_kernel": "TRTRI", "trsv": "trsv", "gram": "gram"}
n = 0
for r in rows[start:]:
    name = r["Kernel_Name"]
    tag = next((v for k, v in short.items() if k in name), name[:20])
    if "gemm_nt" in name:
        tag += name[name.index("<"):name.index(">") + 1]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%9.1f +%8.1f us  q%-3s grid %-7s %s" % (s / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r.get("Grid_Size_X", "?") + "x" + r.get("Grid_Size_Y", "?"), tag))
    n += 1
    if n >= lim or "trsv" in name:
        break
