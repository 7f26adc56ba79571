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
    keep = (kind != CENSUS) & (kind != 7)
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
    if len(sys.argv) > 3:
        k0, k1 = int(sys.argv[2]), int(sys.argv[3])
        sel = [i for i in range(n) if kind[i] != 5 and k0 <= (ev[i, 2] if kind[i] in (1, 4) else ev[i, 1]) <= k1]
        sel.sort(key=lambda i: T[i, 0])
        for i in sel:
            print("  %-5s (%3d,%3d,%d) wg %3d  claim %9.1f ready %9.1f mid %9.1f end %9.1f" % (KIND[int(kind[i])], ev[i, 1], ev[i, 2], ev[i, 3], wg[i], T[i, 0], T[i, 1], T[i, 2], T[i, 3]))


if __name__ == "__main__":
    main()
