#!/usr/bin/env python3
"""Timeline of one dataflow factorisation from its GPX_DFLOW_TRACE file (csrc/dflow.hip, trace_event): per step the leaf's wait and
run time, the chain tasks in between, and a summary by task kind.  usage: dflow_trace.py FILE [detail_k0 detail_k1]"""
import sys

import numpy as np

KIND = {0: "SQ", 1: "COLc", 2: "DIAG", 3: "LEAF", 4: "COL", 5: "BULK"}   # DIAG: t1 first columns ready, t2 last column ready
CENSUS = 6


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.uint64)
    n = int(raw[0])
    ev = raw[8:8 + 8 * n].reshape(n, 8).astype(np.int64)
    kind = ev[:, 0] & 0xff
    wg = ev[:, 0] >> 8
    cen = ev[kind == CENSUS]
    if len(cen):
        from collections import Counter
        keys = Counter((int(c[2]) & 0xf, (int(c[1]) >> 8) & 0xff) for c in cen)
        print("census: %d workgroups reported, %d distinct (xcc, se/sh/cu) keys, per key: %s; mates that left %d; leaf key %#x" % (
            len(cen), len(keys), dict(Counter(keys.values())), int(raw[1]), int(raw[2])))
        print("  sample hw ids:", " ".join("%#x/%d" % (int(c[1]) & 0xffffffff, int(c[2]) & 0xf) for c in cen[:12]))
    for g in ev[kind == 7]:
        print("GAVE UP: wg %d waited for state word %d >= %d, saw %d" % (int(g[0]) >> 8, int(g[1]), int(g[2]), int(g[3])))
    for kk, names in ((8, ("look", "decode", "operands check + acquire")), (9, ("product+stores issued", "-", "drain+publish"))):
        m = kind == kk
        if m.any():
            Tx = ev[m][:, 4:8] / 100.0
            print("BULK phases (kind %d): " % kk + ", ".join("%s %.2f us" % (nm, (Tx[:, c + 1] - Tx[:, c]).mean()) for c, nm in enumerate(names) if nm != "-"))
    keep = (kind != CENSUS) & (kind != 7) & (kind != 8) & (kind != 9)
    ev, kind, wg = ev[keep], kind[keep], wg[keep]
    n = len(ev)
    t00 = ev[:, 4].min()
    T = (ev[:, 4:8] - t00) / 100.0       # us
    print("%d events, span %.1f us" % (n, T[:, 3].max()))
    for k in sorted(KIND):
        m = kind == k
        if m.any():
            wait = T[m, 1] - T[m, 0]
            run = T[m, 3] - T[m, 1]
            print("  %-5s n=%6d  wait mean %7.1f max %8.1f | run mean %7.1f max %8.1f  sum run %.1f ms  on %d WGs" % (
                KIND[k], m.sum(), wait.mean(), wait.max(), run.mean(), run.max(), run.sum() / 1e3, len(set(wg[m]))))
    leaf = ev[kind == 3]
    Tl = T[kind == 3]
    order = np.argsort(leaf[:, 1])
    print("leaf k: start-of-wait  wait  body  publish | step (end-to-end of consecutive leaves)")
    prev_end = None
    for o in order:
        k = leaf[o, 1]
        line = "  k=%3d  %9.1f  wait %7.1f  body %6.1f  pub %5.1f" % (k, Tl[o, 0], Tl[o, 1] - Tl[o, 0], Tl[o, 2] - Tl[o, 1], Tl[o, 3] - Tl[o, 2])
        if prev_end is not None:
            line += "  | step %7.1f" % (Tl[o, 3] - prev_end)
        prev_end = Tl[o, 3]
        print(line)
    # utilisation: share of the span's worker-time spent inside tasks, in bins
    span = T[:, 3].max()
    nbin = 60
    edges = np.linspace(0.0, span, nbin + 1)
    nw = len(set(wg))
    busy = np.zeros((nbin, 6))
    for k in (0, 1, 2, 4, 5):
        m = kind == k
        a, b = T[m, 1], T[m, 3]          # from ready to end
        for bi in range(nbin):
            lo, hi = edges[bi], edges[bi + 1]
            busy[bi, k] = np.clip(np.minimum(b, hi) - np.maximum(a, lo), 0, None).sum()
    print("utilisation by %.0f us bin (share of %d workgroups' time inside BULK | COL | SQ | chain tasks, from ready to end):" % (span / nbin, nw))
    for bi in range(nbin):
        tot = (edges[bi + 1] - edges[bi]) * nw
        print("  %8.0f us  bulk %5.1f%%  col %5.1f%%  sq %4.1f%%  chain %4.1f%%  | idle %5.1f%%" % (
            edges[bi], 100 * busy[bi, 5] / tot, 100 * busy[bi, 4] / tot, 100 * busy[bi, 0] / tot, 100 * (busy[bi, 1] + busy[bi, 2]) / tot,
            100 * (1 - busy[bi].sum() / tot)))
    # panel boundaries: what the first leaf of square q waited for
    nbp = 8
    leaf_end = {int(ev[i, 1]): T[i, 3] for i in range(n) if kind[i] == 3}
    leaf_start = {int(ev[i, 1]): T[i, 1] for i in range(n) if kind[i] == 3}
    nbr = max(leaf_end) + 1
    print("panel boundaries (us): leaf 8q-1 done | per column k of panel q-1: narrow tiles of column k all in / COL of the next square's rows done | SQ(q) first start .. last end | leaf 8q starts")
    for q in range(1, (nbr + nbp - 1) // nbp):
        k0 = nbp * (q - 1)
        line = "  q=%2d  leaf %3d done %8.1f |" % (q, nbp * q - 1, leaf_end.get(nbp * q - 1, -1))
        for k in range(k0, nbp * q):
            nar = [T[i, 3] for i in range(n) if kind[i] == 5 and ev[i, 2] == k]
            col = [T[i, 3] for i in range(n) if kind[i] in (1, 4) and ev[i, 2] == k and nbp * q <= ev[i, 1] < nbp * (q + 1)]
            line += " %6.0f/%6.0f" % (max(nar) if nar else -1, max(col) if col else -1)
        sq = [(T[i, 1], T[i, 3]) for i in range(n) if kind[i] == 0 and nbp * q <= ev[i, 1] < nbp * (q + 1)]
        if sq:
            line += " | SQ %8.1f .. %8.1f" % (min(a for a, b in sq), max(b for a, b in sq))
        line += " | leaf %3d starts %8.1f" % (nbp * q, leaf_start.get(nbp * q, -1))
        print(line)
    if len(sys.argv) > 3:
        k0, k1 = int(sys.argv[2]), int(sys.argv[3])
        sel = [i for i in range(n) if kind[i] != 5 and k0 <= (ev[i, 2] if kind[i] in (1, 4) else ev[i, 1]) <= k1 and (kind[i] != 4 or ev[i, 1] < k1 + 9)]
        sel.sort(key=lambda i: T[i, 0])
        for i in sel:
            print("  %-5s (%3d,%3d,%d) wg %3d  claim %9.1f ready %9.1f mid %9.1f end %9.1f" % (KIND[int(kind[i])], ev[i, 1], ev[i, 2], ev[i, 3], wg[i], T[i, 0], T[i, 1], T[i, 2], T[i, 3]))


if __name__ == "__main__":
    main()
