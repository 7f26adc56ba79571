"""Timeline of one gpx_fit from a rocprofv3 kernel trace (rocpd database): per outer panel the bulk launch, the chain of
leaves and the main stream's idle time.  usage: fit_timeline.py DB [fit_index]"""
import re
import sqlite3
import sys


def load(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, d.start, d.end, d.queue_id, d.grid_size_x/d.workgroup_size_x, d.grid_size_y from %s d join %s s "
         "on d.kernel_id=s.id order by d.start" % (disp, sym))
    return list(c.execute(q))


def short(n):
    if "potrf" in n:
        return "LEAF"
    if "persistent" in n:
        return "G44P"
    if "trap_signal" in n:
        return "TRAP"
    m = re.search(r"gemm_nt_f64_kernelILi(\d)ELi(\d)ELb(\d)", n)
    if m:
        return "G%s%s%s" % (m.group(1), m.group(2), "L" if m.group(3) == "1" else "")
    return n[3:20]


def main():
    rows = load(sys.argv[1])
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    seeds = [i for i, r in enumerate(rows) if "ts_seed" in r[0]]
    i1 = seeds[k]
    i0 = max(i for i in range(i1) if "gram_kernel" in rows[i][0])
    # the fit assembles the first panel's columns in a launch of its own just before the rest of the Gram matrix
    for i in range(i0 - 1, max(i0 - 6, -1), -1):
        if "gram_kernel" in rows[i][0] and rows[i0][1] - rows[i][2] < 200000:
            i0 = i
    t0 = rows[i0][1]
    win = rows[i0:i1]
    print("fit window %.2f ms" % ((rows[i1][1] - t0) / 1e6))
    qs = {}
    for r in win:
        qs.setdefault(r[3], []).append(r)
    for qid, lst in qs.items():
        print("  queue %d: %4d launches, busy %.2f ms" % (qid, len(lst), sum(r[2] - r[1] for r in lst) / 1e6))
    bulk = [r for r in win if short(r[0]) in ("G44L", "G44P", "TRAP")]
    print("bulk launches: total %.2f ms" % (sum(r[2] - r[1] for r in bulk) / 1e6))
    leaves = [r for r in win if "potrf" in r[0]]
    for p in range(0, len(leaves), 8):
        grp = leaves[p:p + 8]
        b = bulk[p // 8 - 1] if 0 < p // 8 <= len(bulk) else None
        print("  panel %2d chain %8.1f -> %8.1f (%7.1f us)  leaves %s   %s" % (
            p // 8, (grp[0][1] - t0) / 1e3, (grp[-1][2] - t0) / 1e3, (grp[-1][2] - grp[0][1]) / 1e3,
            " ".join("%4.0f" % ((g[2] - g[1]) / 1e3) for g in grp),
            ("bulk %8.1f -> %8.1f (%6.1f us, %d wgs)" % ((b[1] - t0) / 1e3, (b[2] - t0) / 1e3, (b[2] - b[1]) / 1e3, b[4])) if b else ""))


def detail(path, k):
    """every launch of the main queue inside the fit window, with the idle time in front of it"""
    rows = load(path)
    seeds = [i for i, r in enumerate(rows) if "ts_seed" in r[0]]
    i1 = seeds[k]
    i0 = max(i for i in range(i1) if "gram_kernel" in rows[i][0])
    for i in range(i0 - 1, max(i0 - 6, -1), -1):
        if "gram_kernel" in rows[i][0] and rows[i0][1] - rows[i][2] < 200000:
            i0 = i
    t0 = rows[i0][1]
    win = rows[i0:i1]
    mainq = max(set(r[3] for r in win), key=lambda q: sum(r[2] - r[1] for r in win if r[3] == q and short(r[0]) in ("G44L", "G44", "TRAP")))
    prev = None
    print("main queue %d: start(us) dur(us) idle-before(us) kernel grid" % mainq)
    for r in win:
        if r[3] != mainq:
            continue
        idle = (r[1] - prev) / 1e3 if prev is not None else 0.0
        print("  %9.1f %8.1f %8.1f  %-6s %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, idle, short(r[0]), r[4] * max(1, r[5])))
        prev = r[2]


if __name__ == "__main__":
    main()
    if len(sys.argv) > 3 and sys.argv[3] == "detail":
        detail(sys.argv[1], int(sys.argv[2]))
