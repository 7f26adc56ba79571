"""One-off differential fuzz: GPU path vs the oracle on many random small configurations (sizes around tile and panel
boundaries, random dimensions and hyper-parameters, occasional duplicate rows)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch  # noqa: F401
import skgpuppy_amd as sk
from oracle import oracle as orc

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 150
sizes = [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 500, 640, 1023, 1024, 1025, 1100, 1500, 2047, 2050, 2300, 3073]
worst = dict(mean=0.0, var=0.0, approx=0.0, exact=0.0)
for c in range(ncase):
    N = int(rng.choice(sizes)) if rng.rand() < 0.7 else int(rng.randint(1, 1300))
    d = int(rng.randint(1, 11))
    M = int(rng.choice([1, 2, 7, 128, 129, 300]))
    x = rng.uniform(0, 10, (N, d))
    if N > 10 and rng.rand() < 0.2:
        x[N // 2] = x[N // 3]                      # an exact duplicate row (vt > 0 keeps K positive definite)
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
    xs = rng.uniform(0, 10, (M, d))
    if rng.rand() < 0.3 and N > 1:
        xs[0] = x[rng.randint(N)]                  # query equal to a training row (the +vt quirk)
    theta = np.concatenate([[rng.uniform(-1, 1), rng.uniform(-6, -2)], rng.uniform(-4.5, -1.0, d)])
    gp = sk.GaussianProcess(x, t, sk.GaussianCovariance(), theta.copy())
    og = orc.OracleGP(x, t, theta)
    m, v = gp.estimate_many(xs); om, ov = og.estimate_many(xs)
    sv = np.exp(theta[0])
    em, ev = np.abs(m - om).max(), np.abs(v - ov).max() / sv
    worst["mean"] = max(worst["mean"], em); worst["var"] = max(worst["var"], ev)
    ok = em < 1e-6 * max(1.0, np.abs(om).max()) and ev < 1e-6
    if N >= 2:
        u = xs[0]; S = np.diag(rng.uniform(0.001, 0.05, d))
        a = sk.UncertaintyPropagationApprox(gp).propagate_GA(u, S); oa = orc.approx_propagate(og, u, S)
        e = sk.UncertaintyPropagationExact(gp).propagate_GA(u, S); oe = orc.exact_propagate(og, u, S)
        ea = max(abs(a[0] - oa[0]), abs(a[1] - oa[1]) / sv); ee = max(abs(e[0] - oe[0]), abs(e[1] - oe[1]) / sv)
        worst["approx"] = max(worst["approx"], ea); worst["exact"] = max(worst["exact"], ee)
        ok = ok and ea < 1e-6 and ee < 1e-6
    # the few-vector solver and the sampler's L z on random right-hand sides (gpx_solve / gpx_chol_mul)
    nr = int(rng.choice([1, 3, 16, 17, 33]))
    B = rng.randn(nr, N)
    kb = gp._dev().solve(B)
    ko = og.Kinv.dot(B.T).T
    es = np.abs(kb - ko).max() / max(1e-300, np.abs(ko).max())
    Lz = gp._dev().chol_mul(B)
    with np.errstate(divide="ignore"):
        Lo = np.linalg.cholesky(orc.gram(x, theta))
    el = np.abs(Lz - B.dot(Lo.T)).max() / max(1e-300, np.abs(Lz).max())
    worst["solve"] = max(worst.get("solve", 0.0), es); worst["chol_mul"] = max(worst.get("chol_mul", 0.0), el)
    ok = ok and es < 1e-6 and el < 1e-9
    if not ok:
        print("MISMATCH case", c, "N", N, "d", d, "M", M, "theta", theta, em, ev, es, el)
        sys.exit(1)
    gp._dev().close()
print("fuzz ok: %d cases, worst abs deviations (variances relative to v): %s" % (ncase, worst))
