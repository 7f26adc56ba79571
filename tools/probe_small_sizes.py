#!/usr/bin/env python3
"""End-to-end latency of the Python classes on SMALL problems (the reference's everyday sizes): GaussianProcess(...) fit,
estimate_many, propagate_GA (Approx), ms, best of 7 after 2 warm-ups."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import numpy as np
import torch  # noqa: F401
import skgpuppy_amd as sk

for N, d, M in ((100, 2, 50), (500, 3, 200), (1000, 4, 1000), (2000, 4, 2000), (4096, 4, 4096)):
    rng = np.random.RandomState(N)
    x = rng.uniform(0, 10, (N, d)); t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N); xs = rng.uniform(0, 10, (M, d))
    theta = np.log(np.array([2.0, 0.01] + [0.04] * d))
    cov = sk.GaussianCovariance()
    u, S = np.full(d, 5.0), 0.01 * np.eye(d)
    best = [1e9] * 4
    for rep in range(9):
        a = time.perf_counter()
        gp = sk.GaussianProcess(x, t, cov, theta.copy())
        b = time.perf_counter()
        gp.estimate_many(xs)
        c = time.perf_counter()
        up = sk.UncertaintyPropagationApprox(gp)
        up.propagate_GA(u, S)
        e = time.perf_counter()
        up.propagate_GA(u + 0.1, S)
        f = time.perf_counter()
        if rep >= 2:
            best = [min(best[0], b - a), min(best[1], c - b), min(best[2], e - c), min(best[3], f - e)]
        gp._dev().close()
    print("N=%5d d=%d M=%5d  fit %7.3f ms  estimate_many %7.3f ms  propagate_GA first %7.3f ms  next %7.3f ms" % (
        N, d, M, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3), flush=True)
