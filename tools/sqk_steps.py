#!/usr/bin/env python3
"""Critical path of a SQUARE launch of the dataflow kernel (csrc/dflow.hip) from its GPX_DFLOW_TRACE file, step by step:
leaf k published -> COL(k+1, k, s) sees it (hand-off 1) -> slab solved + published -> DIAG(k+1, s) sees the last column (hand-off 2) ->
last 128-deep product + publish -> leaf k+1 sees the four slabs (hand-off 3) -> leaf body.   usage: sqk_steps.py FILE"""
import sys

import numpy as np


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.uint64)
    n = int(raw[0])
    ev = raw[8:8 + 8 * n].reshape(n, 8).astype(np.int64)
    kind = ev[:, 0] & 0xff
    wg = ev[:, 0] >> 8
    t00 = ev[(kind == 3)][:, 4].min()
    T = (ev[:, 4:8] - t00) / 100.0
    leaf = {int(e[1]): T[i] for i, e in enumerate(ev) if kind[i] == 3}
    col = {}
    diag = {}
    for i, e in enumerate(ev):
        if kind[i] in (1, 4):
            col[(int(e[1]), int(e[2]), int(e[3]))] = (T[i], int(wg[i]), int(kind[i]))
        elif kind[i] == 2:
            diag[(int(e[1]), int(e[3]))] = (T[i], int(wg[i]))
    ks = sorted(leaf)
    print("launch span %.1f us, %d leaves, %d COL, %d DIAG events" % (T[:, 3].max(), len(ks), len(col), len(diag)))
    print("  k | leaf: wait  body   pub | h1 leaf pub -> COL(k+1,k,*) sees | solve+pub | h2 -> DIAG(k+1,*) sees | prod+pub | h3 -> leaf k+1 sees | step")
    rows = []
    for k in ks:
        L = leaf[k]
        line = " %2d | %6.1f %5.1f %5.1f |" % (k, L[1] - L[0], L[2] - L[1], L[3] - L[2])
        if k + 1 in leaf:
            cs = [col[(k + 1, k, s)][0] for s in range(4) if (k + 1, k, s) in col]
            ds = [diag[(k + 1, s)][0] for s in range(4) if (k + 1, s) in diag]
            if len(cs) == 4 and len(ds) == 4:
                h1 = max(c[2] for c in cs) - L[3]            # leafdone seen (t3 slot 2) by the last slab
                h1min = min(c[2] for c in cs) - L[3]
                sol = max(c[3] - c[2] for c in cs)
                cend = max(c[3] for c in cs)
                h2 = max(d[2] for d in ds) - cend
                prod = max(d[3] - d[2] for d in ds)
                dend = max(d[3] for d in ds)
                N = leaf[k + 1]
                h3 = N[1] - dend
                step = N[3] - L[3]
                line += "  %5.1f (first %5.1f)            |  %6.1f   |  %6.1f             |  %6.1f  |  %6.1f           | %6.1f" % (h1, h1min, sol, h2, prod, h3, step)
                rows.append((L[2] - L[1], L[3] - L[2], h1, sol, h2, prod, h3, step))
        print(line)
    if rows:
        r = np.array(rows)
        print("mean over %d steps: leaf body %.1f  pub %.1f | h1 %.1f  solve+pub %.1f | h2 %.1f  prod+pub %.1f | h3 %.1f | step %.1f us" % (
            (len(r),) + tuple(r.mean(axis=0))))
    # rows below the square (kind 4): how far behind the chain they trail
    below = [(k_, v) for k_, v in col.items() if v[2] == 4]
    if below:
        last = {}
        for (i, k, s), v in below:
            last[k] = max(last.get(k, 0.0), v[0][3])
        print("rows below the square: last COL of column k ends at (us after that column's leaf published): " +
              " ".join("%d:%.0f" % (k, last[k] - leaf[k][3]) for k in sorted(last) if k in leaf))


if __name__ == "__main__":
    main()
