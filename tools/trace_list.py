"""Every launch of a rocprofv3 kernel trace (rocpd database) behind the k-th occurrence of a marker kernel, in start order:
start (us since the marker), duration, idle time in front of it on its queue, queue, workgroups, kernel.
usage: trace_list.py DB MARKER_SUBSTRING [k=-1] [max_rows=400]"""
import sys

from fit_timeline import load, short


def main():
    rows = load(sys.argv[1])
    marker = sys.argv[2]
    k = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    cap = int(sys.argv[4]) if len(sys.argv) > 4 else 400
    hits = [i for i, r in enumerate(rows) if marker in r[0]]
    i0 = hits[k]
    t0 = rows[i0][1]
    last = {}
    print("start_us   dur_us  idle_us queue  wgs    kernel")
    for r in rows[i0:i0 + cap]:
        idle = (r[1] - last[r[3]]) / 1e3 if r[3] in last else 0.0
        last[r[3]] = r[2]
        print("%9.1f %8.1f %8.1f %4d %6d x%-4d %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, idle, r[3], r[4], r[5], short(r[0]) + " " + r[0][:60]))


if __name__ == "__main__":
    main()
