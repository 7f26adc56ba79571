#!/usr/bin/env python3
"""BASELINE config 5 (Snelson SPGP, M = 2048 pseudo-inputs, N = 262144, d = 8) on one MI355X: fit, estimate_many and
Snelson's likelihood through the C-ABI, timed with host clocks around synchronous calls.  Not the headline bench
(bench.py is); the numbers go into DESIGN.md.

    python3 tools/bench_spgp.py [--n 262144] [--m 2048] [--d 8] [--queries 16384] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=262144)
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--queries", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    import torch  # noqa: F401  (one libamdhip64)
    import skgpuppy_amd as sk

    rng = np.random.RandomState(20240 + a.n + a.d)
    x = rng.uniform(0, 10, (a.n, a.d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(a.n)
    xs = rng.uniform(0, 10, (a.queries, a.d))
    th_gc = np.log(np.array([2.0, 0.01] + [0.04] * a.d))
    xb = x[rng.choice(a.n, a.m, replace=False)].copy()
    theta = np.concatenate([th_gc, xb.ravel()])
    cov = sk.SPGPCovariance(a.m)
    fit, pred, nll, grad, grad2, lbfgs = [], [], [], [], [], []
    for r in range(a.reps + 1):
        t0 = time.perf_counter()
        gp = sk.GaussianProcess(x, t, cov, theta)
        t1 = time.perf_counter()
        mu, var = gp.estimate_many(xs)
        t2 = time.perf_counter()
        val = gp._dev().nll()             # on its own (estimate_many overwrote whatever a likelihood and a gradient share)
        t3 = time.perf_counter()
        g = gp._dev().nll_grad()          # right behind the likelihood at the same theta: their common N m^2 part is there
        t4 = time.perf_counter()
        gp.estimate_many(xs[:16])         # (discards it again)
        t4a = time.perf_counter()
        g = gp._dev().nll_grad()          # on its own
        t4b = time.perf_counter()
        gp._dev().close()
        # one L-BFGS iteration as SPGPCovariance.ml_estimate drives it: likelihood + gradient at a new theta (host arrays in)
        th2 = theta + 1e-3 * (r + 1)
        t5 = time.perf_counter()
        f2 = cov._negativeloglikelihood(x, t - t.mean(), th2)
        g2 = cov._d_nll_d_theta(x, t - t.mean(), th2)
        t6 = time.perf_counter()
        if r:   # first round = warm-up (allocator, code objects)
            fit.append(t1 - t0); pred.append(t2 - t1); nll.append(t3 - t2); grad.append(t4b - t4a); grad2.append(t4 - t3); lbfgs.append(t6 - t5)
    N, M = a.n, a.m
    flops_fit = N * M * M + N * M * M + M ** 3 / 3 * 2     # TRSM + lower-only W^T W + two Cholesky
    out = {"workload": "SPGP fit + estimate_many, N=%d M=%d d=%d, %d queries" % (N, M, a.d, a.queries),
           "fit_ms": 1e3 * min(fit), "predict_ms": 1e3 * min(pred), "snelson_nll_ms": 1e3 * min(nll),
           "analytic_gradient_ms": 1e3 * min(grad), "gradient_after_nll_ms": 1e3 * min(grad2), "ml_fit_iteration_ms": 1e3 * min(lbfgs),
           "gradient_entries": int(g.size), "gradient_finite": bool(np.all(np.isfinite(g)) and np.all(np.isfinite(g2)) and np.isfinite(f2)),
           "fit_tflops_algorithmic": flops_fit / min(fit) / 1e12,
           "train_pts_per_s": N / min(fit), "query_pts_per_s": a.queries / min(pred),
           "nll": val, "mean_abs_residual": float(np.abs(mu - np.sin(0.3 * xs.sum(1))).mean()), "var_range": [float(var.min()), float(var.max())]}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
