"""Cost of one hyper-parameter-fit iteration at C3: likelihood (fit + logdet), gradient (K^-1 build + fused gradient pass)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import numpy as np
import torch  # noqa: F401
import bench
import skgpuppy_amd as sk
N, d = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 8
x, t, xs, th = bench.recipe(N, d, 16)
tc = t - t.mean()
cov = sk.GaussianCovariance()
for rep in range(3):
    th2 = th + 0.01 * rep
    t0 = time.perf_counter(); v = cov._negativeloglikelihood(x, tc, th2)
    t1 = time.perf_counter(); g = cov._d_nll_d_theta(x, tc, th2)
    t2 = time.perf_counter()
    print("N=%d: nll %.2f ms, gradient (same theta, cached model) %.2f ms   nll=%.4f |g|=%.3e" % (N, (t1 - t0) * 1e3, (t2 - t1) * 1e3, v, np.abs(g).max()))
