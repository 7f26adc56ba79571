#!/usr/bin/env python3
"""Repeatability of the fit: the factorisation has no atomics in its arithmetic and fixed reduction orders, so K^-1 t must come out
bit-identical from every run -- any difference between repetitions of the same problem is a race in the stream schedule.  Sizes are
interleaved (different panel counts, pool and stream-cache states), with propagation / K^-1 / prediction calls in between."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch  # noqa: F401,E402
import bench  # noqa: E402
import skgpuppy_amd as sk  # noqa: E402
from skgpuppy_amd import _gpx  # noqa: E402

sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16384, 8200, 16384, 4096, 12000, 16384, 2500, 16384, 12000, 8200, 16384]
first = {}
worst = {}
for r, N in enumerate(sizes):
    d = 8
    x, t, xs, theta = bench.recipe(N, d, 64)
    gp = sk.GaussianProcess(x, t, sk.GaussianCovariance(), theta.copy())
    if r % 3 == 1:
        sk.UncertaintyPropagationApprox(gp).propagate_GA(np.full(d, 5.0), 0.01 * np.eye(d))
    if r % 3 == 2:
        sk.UncertaintyPropagationExact(gp).propagate_GA(np.full(d, 5.0), 0.01 * np.eye(d))
    mean, var = gp.estimate_many(xs)
    beta = gp._get_beta()
    gp._dev().close()
    if r % 4 == 3:
        _gpx.lib.gpx_pool_trim()
    key = N
    if key not in first:
        first[key] = (beta, mean)
        print("run %2d N=%5d: first" % (r, N))
    else:
        db, dm = float(np.abs(beta - first[key][0]).max()), float(np.abs(mean - first[key][1]).max())
        worst[key] = max(worst.get(key, 0.0), db, dm)
        print("run %2d N=%5d: max |beta - beta_0| = %.3e   max |mean - mean_0| = %.3e" % (r, N, db, dm))
print("worst deviation per size:", worst)
