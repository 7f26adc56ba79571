#!/usr/bin/env python3
"""Straw-man cross-check of the CPU baseline: time the oracle (oracle/oracle.py, what bench.py reports as
`cpu_baseline`, kind "port") against the GENUINE reference on the same machine, same inputs, same BLAS threads.  The oracle
must be within +-20 % of the reference, otherwise the reported baseline is not a fair stand-in.

Runs in the BUILD CONTAINER only: the reference (/root/reference) cannot travel to the GPU box, so this is a builder-side
tool whose output is committed under profiles/.  It imports the reference read-only (no shim is needed for
skgpuppy.Covariance / skgpuppy.GaussianProcess).

    python3 tools/check_baseline_speed.py [--n 4096] [--d 4] [--reps 2] [--out profiles/r02_baseline_speed_check.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("GPX_REFERENCE", "/root/reference")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--d", type=int, default=4)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if not os.path.isdir(os.path.join(REF, "skgpuppy")):
        raise SystemExit("reference tree not found at %s (this tool runs in the build container only)" % REF)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import bench
    from oracle import oracle as orc
    from skgpuppy.Covariance import GaussianCovariance
    from skgpuppy.GaussianProcess import GaussianProcess

    N = M = a.n
    x, t, xs, theta = bench.recipe(N, a.d, M)

    def run_ref():
        t0 = time.perf_counter()
        gp = GaussianProcess(x, t, GaussianCovariance(), theta.copy())
        t1 = time.perf_counter()
        mean, var = gp.estimate_many(xs)
        return t1 - t0, time.perf_counter() - t1, mean, var

    def run_orc():
        t0 = time.perf_counter()
        gp = orc.OracleGP(x, t, theta)
        t1 = time.perf_counter()
        mean, var = gp.estimate_many(xs)
        return t1 - t0, time.perf_counter() - t1, mean, var

    run_orc()   # warm BLAS threads / page in
    best_ref = best_orc = None
    for _ in range(a.reps):      # interleaved, best of reps
        r = run_ref()
        o = run_orc()
        if best_ref is None or r[0] + r[1] < best_ref[0] + best_ref[1]:
            best_ref = r
        if best_orc is None or o[0] + o[1] < best_orc[0] + best_orc[1]:
            best_orc = o
    tr, to = best_ref[0] + best_ref[1], best_orc[0] + best_orc[1]
    out = {
        "workload": "N=M=%d d=%d (bench.recipe), fit + estimate_many" % (N, a.d),
        "host": bench.host_info(),
        "reference": {"fit_s": best_ref[0], "estimate_many_s": best_ref[1], "pts_per_s": (N + M) / tr},
        "oracle": {"fit_s": best_orc[0], "estimate_many_s": best_orc[1], "pts_per_s": (N + M) / to},
        "oracle_over_reference_time": to / tr,
        "within_20_percent": bool(abs(to / tr - 1.0) <= 0.20),
        "outputs_agree": {"max_abs_dmean": float(np.abs(best_ref[2] - best_orc[2]).max()),
                          "max_abs_dvar": float(np.abs(best_ref[3] - best_orc[3]).max())},
    }
    line = json.dumps(out, indent=1)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
    if not out["within_20_percent"]:
        raise SystemExit("oracle is NOT within +-20 %% of the reference (ratio %.3f)" % (to / tr))


if __name__ == "__main__":
    main()
