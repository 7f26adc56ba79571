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
    if "chol_dataflow" in n:
        return "SQK"
    if "persistent" in n:
        return "G44P"
    if "trap_signal" in n:
        return "TRAP"
    m = re.search(r"gemm_nt_f64_kernelILi(\d)ELi(\d)ELb(\d)", n)
    if m:
        return "G%s%s%s" % (m.group(1), m.group(2), "L" if m.group(3) == "1" else "")
    return n[3:20]


def window(rows, k):
    """launches of the k-th fit of the trace: from its ts_pack (the fit packs its targets for the forward substitution before
    anything else) to the second ts_unpack behind it (y, alpha)"""
    packs = [i for i, r in enumerate(rows) if "ts_pack" in r[0]]
    i0 = packs[k]
    unp = [i for i in range(i0, len(rows)) if "ts_unpack" in rows[i][0]]
    i1 = unp[1] + 1
    return rows[i0:i1], rows[i0][1]


def main():
    rows = load(sys.argv[1])
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    win, t0 = window(rows, k)
    print("fit window %.2f ms" % ((win[-1][2] - t0) / 1e6))
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
    for r in win:
        if short(r[0]) == "SQK":
            print("  square kernel %8.1f -> %8.1f (%7.1f us, %d wgs)" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[4]))
    tail = [r for r in win if r[1] > leaves[-1][2]]
    print("after the last leaf: %.1f us, %d launches" % ((win[-1][2] - leaves[-1][2]) / 1e3, len(tail)))


def detail(path, k):
    """every launch of the main queue inside the fit window, with the idle time in front of it"""
    rows = load(path)
    win, t0 = window(rows, k)
    mainq = max(set(r[3] for r in win), key=lambda q: sum(r[2] - r[1] for r in win if r[3] == q and short(r[0]) in ("G44L", "G44", "TRAP")))
    prev = None
    print("main queue %d: start(us) dur(us) idle-before(us) kernel grid" % mainq)
    for r in win:
        if r[3] != mainq:
            continue
        idle = (r[1] - prev) / 1e3 if prev is not None else 0.0
        print("  %9.1f %8.1f %8.1f  %-6s %d" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, idle, short(r[0]), r[4] * max(1, r[5])))
        prev = r[2]


def everything(path, k, ta, tb):
    """every launch of every queue that overlaps [ta, tb] us of the fit window"""
    rows = load(path)
    win, t0 = window(rows, k)
    print("all queues, %.0f..%.0f us: queue start(us) dur(us) kernel grid" % (ta, tb))
    for r in win:
        a, b = (r[1] - t0) / 1e3, (r[2] - t0) / 1e3
        if b >= ta and a <= tb:
            print("  q%d %9.1f %8.1f  %-18s %d" % (r[3], a, b - a, short(r[0]), r[4] * max(1, r[5])))


if __name__ == "__main__":
    main()
    if len(sys.argv) > 3 and sys.argv[3] == "detail":
        detail(sys.argv[1], int(sys.argv[2]))
    if len(sys.argv) > 5 and sys.argv[3] == "all":
        everything(sys.argv[1], int(sys.argv[2]), float(sys.argv[4]), float(sys.argv[5]))
