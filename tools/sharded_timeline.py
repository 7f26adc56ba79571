"""Timeline of the LAST sharded fit in a rocprofv3 kernel trace (rocpd database) of a one-rank run of the sharded path
(GPX_BENCH_SHARDED=1 ... bench.py --gpus 1 --workload c3): per outer panel the chain (leaves or the square launch), the end of the
row slices that trail it (the next panel's square rows -- the message's head -- and the far rows -- its tail) and the start of the
NEXT panel's chain: with the message in two parts the next chain starts when the head rows are through, not when the far rows are.
usage: sharded_timeline.py DB [gram_marker_index=-1]"""
import sys

from fit_timeline import load, short


def main():
    rows = load(sys.argv[1])
    # a sharded fit starts with its Gram launches, one per owned panel back to back on the main stream (predict has a single one per
    # chunk): the last group of at least four gram kernels less than 2 ms apart
    grams = [i for i, r in enumerate(rows) if "gram" in r[0]]
    groups = []
    for i in grams:
        if groups and rows[i][1] - rows[groups[-1][-1]][1] < 2e6:
            groups[-1].append(i)
        else:
            groups.append([i])
    groups = [g for g in groups if len(g) >= 4]
    grp = groups[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
    i0 = grp[0]
    t0 = rows[i0][1]
    # ... and ends where the next Gram launch (predict's) begins
    later = [rows[i][1] for i in grams if i > grp[-1]]
    end = later[0] if later else rows[-1][2] + 1
    fit = [r for r in rows[i0:] if r[1] < end]
    chain_q = None
    chains = []          # [start, end, kind, steps]
    for r in fit:
        k = short(r[0])
        if k == "LEAF":
            chain_q = r[3]
            if chains and chains[-1][2] == "leaves" and chains[-1][3] < 8:
                chains[-1][1] = r[2]
                chains[-1][3] += 1
            else:
                chains.append([r[1], r[2], "leaves", 1])
        elif k == "SQK":
            chains.append([r[1], r[2], "square launch", 8])
    chains.sort()
    queues = sorted(set(r[3] for r in fit))
    print("fit window %.2f ms, %d launches on queues %s (chain on queue %s)" % ((fit[-1][2] - t0) / 1e6, len(fit), queues, chain_q))
    # every queue's busy time and launch count
    for q in queues:
        rs = [r for r in fit if r[3] == q]
        print("  queue %d: %4d launches, busy %.2f ms, first %.1f us, last end %.1f us" % (
            q, len(rs), sum(r[2] - r[1] for r in rs) / 1e6, (rs[0][1] - t0) / 1e3, (rs[-1][2] - t0) / 1e3))
    print("panel  chain start -> end (us)        kind           rows trailing it on other queues end at (us)      next chain starts")
    for p, c in enumerate(chains):
        nxt = chains[p + 1][0] if p + 1 < len(chains) else None
        # kernels of the other queues that start inside [chain start, next chain's start): the row slices of this panel (the slices of
        # the next panel begin with their update right behind the next square's)
        hi = chains[p + 1][0] if p + 1 < len(chains) else fit[-1][2]
        ends = {}
        for r in fit:
            if r[3] != chain_q and c[0] <= r[1] < hi and short(r[0]) not in ("LEAF", "SQK") and r[4] * max(1, r[5]) < 1200:
                ends.setdefault(r[3], []).append(r[2])
        tail = "  ".join("q%d: %.1f" % (q, (max(v) - t0) / 1e3) for q, v in sorted(ends.items()))
        print("%4d  %9.1f -> %9.1f (%6.1f)  %-13s  %-48s %s" % (
            p, (c[0] - t0) / 1e3, (c[1] - t0) / 1e3, (c[1] - c[0]) / 1e3, c[2], tail, "%.1f" % ((nxt - t0) / 1e3) if nxt else "-"))


if __name__ == "__main__":
    main()
