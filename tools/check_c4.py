import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/scikit-gpuppy_amd")
import torch
import bench, skgpuppy_amd as sk
N, d = 65536, 16
x, t, xs, theta = bench.recipe(N, d, 2048)
cov = sk.GaussianCovariance()
t0 = time.time(); gp = sk.GaussianProcess(x, t, cov, theta.copy()); print("fit s", time.time() - t0)
vt = 0.01
beta = gp._get_beta()
rows = np.random.RandomState(1).choice(N, 512, replace=False)
Krows = cov.cov_matrix_ij(x[rows], x, theta); Krows[np.arange(512), rows] += vt
print("resid", np.abs(Krows.dot(beta) - gp.t[rows]).max(), "beta max", np.abs(beta).max())
mean, var = gp.estimate_many(x[rows])
print("interp", np.abs(mean - gp.meant - (gp.t[rows] - vt * beta[rows])).max(), var.min(), var.max())
m1, v1 = gp.estimate_many(xs); m2, v2 = gp.estimate_many(xs[::-1])
print("perm", np.array_equal(m1, m2[::-1]), np.array_equal(v1, v2[::-1]))
ma, va = sk.UncertaintyPropagationApprox(gp).propagate_GA(xs[0], 1e-14 * np.eye(d))
print("approx vs plain", abs(ma - m1[0]), abs(va - v1[0]))
