"""Per-kernel summary of a rocprofv3 rocpd database (rocprofv3 --kernel-trace -d DIR -o NAME -> DIR/NAME_results.db)."""
import sqlite3
import sys


def summary(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from %s d join %s s "
         "on d.kernel_id=s.id group by s.kernel_name order by 4 desc" % (disp, sym))
    return list(c.execute(q))


if __name__ == "__main__":
    print("%-100s %7s %11s %11s" % ("kernel", "calls", "avg_us", "total_ms"))
    for name, n, avg, tot in summary(sys.argv[1]):
        print("%-100s %7d %11.1f %11.2f" % (name[:100], n, avg, tot))
