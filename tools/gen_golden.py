#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the genuine reference.

Runs ONLY in the build container, where the reference tree is mounted at
/root/reference (read-only).  Nothing of the reference travels: this script
writes explicit input AND output arrays (never RNG seeds) to tests/golden/*.npz.

    PYTHONDONTWRITEBYTECODE=1 python3 tools/gen_golden.py

Shims (harness side, the reference tree is never edited): skgpuppy/Utilities.py:10
imports scipy.integrate.romberg (removed in SciPy 1.15) and MLE.py / TaylorPropagation.py
import scipy.misc.derivative; dummies are installed before the import so that
skgpuppy.UncertaintyPropagation (pure-Python backend: weaving=False, cython=False)
can be imported.
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("GPX_REFERENCE", "/root/reference")
OUT = os.environ.get("GPX_GOLDEN_OUT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))


def _install_shims():
    import scipy.integrate
    if not hasattr(scipy.integrate, "romberg"):
        def _romberg(*a, **k):
            raise NotImplementedError("romberg removed from SciPy")
        scipy.integrate.romberg = _romberg
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    try:
        import scipy.misc as _m
    except Exception:  # pragma: no cover
        _m = types.ModuleType("scipy.misc")
        sys.modules["scipy.misc"] = _m
    if not hasattr(_m, "derivative"):
        def _derivative(*a, **k):
            raise NotImplementedError
        _m.derivative = _derivative


def _import_reference():
    _install_shims()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from skgpuppy.Covariance import GaussianCovariance
    from skgpuppy.GaussianProcess import GaussianProcess
    import skgpuppy.UncertaintyPropagation as UP
    assert not UP.weaving and not UP.cython, "expected the pure-Python backend"
    return GaussianCovariance, GaussianProcess, UP


def synth(N, d, M, seed_offset=0):
    """BASELINE.md §3 recipe."""
    rng = np.random.RandomState(20240 + N + d + seed_offset)
    x = rng.uniform(0, 10, (N, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
    xs = rng.uniform(0, 10, (M, d))
    theta = np.log(np.array([2.0, 0.01] + [0.04] * d))
    return x, t, xs, theta


def gram_cases(GC):
    cov = GC()
    out = {}
    rng = np.random.RandomState(7)
    # (name, xi, xj, theta)
    grid = np.array([[a, b] for a in range(10) for b in range(10)])  # int dtype, README.rst:100
    th2 = np.log(np.array([2.0, 0.01, 0.04, 0.04]))
    cases = [("grid_int", grid, grid, th2)]
    x257 = rng.uniform(-3, 3, (257, 5))
    th5 = np.array([0.3, -2.0, -1.0, 0.2, -0.5, 0.7, -2.2])
    cases.append(("n257_d5", x257, x257, th5))
    x64 = rng.randn(64, 16)
    th16 = np.concatenate([[0.1, -3.0], rng.uniform(-3, -1, 16)])
    cases.append(("n64_d16", x64, x64, th16))
    x33 = rng.uniform(-3, 3, (33, 5))
    cases.append(("rect_33x257_d5", x33, x257, th5))
    th_novt = th5.copy()
    th_novt[1] = -np.inf  # vt = 0 appears in the reference's tests
    cases.append(("n257_d5_vt0", x257, x257, th_novt))
    x1 = rng.uniform(0, 1, (130, 1))
    cases.append(("n130_d1", x1, x1, np.array([0.0, -4.0, 2.0])))
    for name, xi, xj, th in cases:
        out["%s__xi" % name] = xi
        out["%s__xj" % name] = xj
        out["%s__theta" % name] = th
        with np.errstate(divide="ignore"):
            out["%s__K_ij" % name] = cov.cov_matrix_ij(xi, xj, th)
            if xi is xj:
                out["%s__K" % name] = cov.cov_matrix(xi, th)
    # derivative Grams and log det of the operator interface (Covariance.py:485-512, :605-657, :189-195; the reference's
    # test_gaussian_cov_derivative, tests/tests.py:485-503, checks the scalar form against central differences)
    x130 = x257[:130]
    for name, xi, xj, th in (("n257_d5", x130, x130, th5), ("rect_33x257_d5", x33, x257, th5), ("n130_d1", x1, x1, np.array([0.0, -4.0, 2.0]))):
        for j in range(len(th)):
            out["%s__dKij_%d" % (name, j)] = cov._d_cov_matrix_d_theta_ij(xi, xj, th, j)
            if xi is xj and j in (0, 1, len(th) - 1):      # (the square form differs from the rectangular one only for j = 1)
                out["%s__dK_%d" % (name, j)] = cov._d_cov_matrix_d_theta(xi, th, j)
        if xi is xj:
            out["%s__logdet" % name] = np.float64(cov._log_det_cov_matrix(xi, th))     # n257_d5: its first 130 rows
    pairs = [(x33[0], x33[0]), (x33[0], x33[1]), (x33[5], x33[9])]
    out["scalar__pairs"] = np.array([[a, b] for a, b in pairs])
    out["scalar__theta"] = th5
    out["scalar__dcov"] = np.array([[cov._d_cov_d_theta(a, b, th5, j) for j in range(len(th5))] for a, b in pairs], dtype=float)
    return out


def gp_case(GC, GP, UP, x, t, theta, xs, us, Sigmas, v_out=0.02, keep_kinv=True,
            with_exact=True):
    """Fit + predict + propagate one configuration; returns a flat dict of arrays."""
    cov = GC()
    gp = GP(x, np.array(t, dtype=float), cov, np.array(theta, dtype=float))
    o = {"x": x, "t_raw": np.asarray(t), "theta": np.asarray(theta, dtype=float), "xs": xs}
    o["meant"] = np.float64(gp.meant)
    o["t_centered"] = gp.t
    if keep_kinv:
        o["Kinv"] = gp.Kinv
    o["beta"] = gp._get_beta()
    sign, logdet = np.linalg.slogdet(cov.cov_matrix(x, gp.theta_min))
    o["logdet"] = np.float64(logdet)
    mean, var = gp.estimate_many(xs)
    o["pred_mean"] = mean
    o["pred_var"] = var
    m1, v1 = gp.estimate(xs[0])
    o["est0"] = np.array([m1, v1])
    o["nu"] = np.int64(len(us))
    o["nS"] = np.int64(len(Sigmas))
    x = np.asarray(x)
    for iu, u in enumerate(us):
        u = np.array(u, dtype=float)
        o["u%d" % iu] = u
        for iS, S in enumerate(Sigmas):
            S = np.array(S, dtype=float)
            o["Sigma%d" % iS] = S
            upa = UP.UncertaintyPropagationApprox(gp)
            ma, va = upa.propagate_GA(u, S)
            o["approx_u%d_S%d" % (iu, iS)] = np.array([ma, va])
            o["approx_mean_only_u%d_S%d" % (iu, iS)] = np.float64(upa.propagate_mean(u, S))
            if iS == 0:
                o["C_ux_u%d" % iu] = upa.C_ux
                o["J_ux_u%d" % iu] = upa.J_ux
                o["H_ux_u%d" % iu] = upa.H_ux
                o["dvh_u%d" % iu] = np.array(
                    [upa._get_variance_dv_h(u, h) for h in range(x.shape[1])])
            o["factor_u%d_S%d" % (iu, iS)] = np.float64(upa._getFactor(u, S, v_out))
            if with_exact:
                upe = UP.UncertaintyPropagationExact(gp)
                me, ve = upe.propagate_GA(u, S)
                o["exact_u%d_S%d" % (iu, iS)] = np.array([me, ve])
                o["exact_mean_only_u%d_S%d" % (iu, iS)] = np.float64(upe.propagate_mean(u, S))
                # propagate_mean with a CALLER-SUPPLIED C_ux on the built-in operator (UncertaintyPropagation.py:269-290: it is used
                # instead of gp._covariance(u, x_i)); a vector that is not the GP's own
                C_alt = 0.5 * np.asarray(upa.C_ux, dtype=float).reshape(-1) * (1.0 + 0.1 * np.cos(np.arange(x.shape[0])))
                o["C_alt_u%d" % iu] = C_alt
                o["exact_mean_C_alt_u%d_S%d" % (iu, iS)] = np.float64(upe.propagate_mean(u, S, C_alt))
    o["v_out"] = np.float64(v_out)
    # "next" row f1: likelihood and gradient at this theta (and at a perturbed one), centred targets as the GP passes them
    for tag, th in (("", np.array(theta, dtype=float)), ("_p", np.array(theta, dtype=float) * 0.9 + 0.05)):
        o["nll" + tag] = np.float64(cov._negativeloglikelihood(x, gp.t, th))
        o["nll_grad" + tag] = np.array(cov._d_nll_d_theta(x, gp.t, th))
        o["theta" + tag + "_used"] = th
    return o


def spgp_cases():
    """"next" row f3: SPGPCovariance at fixed theta (cf. skgpuppy/tests/tests.py:506-530, :768-804)."""
    _install_shims()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from skgpuppy.Covariance import SPGPCovariance, Covariance
    from skgpuppy.GaussianProcess import GaussianProcess as GP
    out = {}

    def one(name, x, t, theta_gc, xm, xs, keep_dense=True, exact_u=None, exact_S=None):
        m, d = xm.shape
        cov = SPGPCovariance(m)
        theta = np.concatenate([theta_gc, xm.ravel()])
        c = {"x": x, "t_raw": t, "theta": theta, "m": np.int64(m), "xs": xs}
        if keep_dense:
            c["cov"] = cov.cov_matrix(x, theta)
            c["inv"] = cov.inv_cov_matrix(x, theta)
        c["cross"] = cov.cov_matrix_ij(xs[:16], x, theta)
        tc = t - np.mean(t)
        c["nll_snelson"] = np.float64(cov._negativeloglikelihood(x, tc, theta))
        c["nll_generic"] = np.float64(Covariance._negativeloglikelihood(cov, x, tc, theta))
        gp = GP(x, t, cov, theta.copy())
        c["pred_mean"], c["pred_var"] = gp.estimate_many(xs)
        c["est0"] = np.array(gp.estimate(xs[0]))
        c["scalar_01"] = np.float64(cov(x[0], x[1], theta))
        c["scalar_00"] = np.float64(cov(x[0], x[0], theta))
        if exact_u is not None:
            # UncertaintyPropagationExact on an SPGP model: the class reads the GP through _get_beta / _get_W_inv / _inv_cov_matrix /
            # _covariance only (UncertaintyPropagation.py:269-290, :323-379), so it runs on the dense SPGP inverse
            import skgpuppy.UncertaintyPropagation as UP
            upe = UP.UncertaintyPropagationExact(gp)
            c["exact_u"], c["exact_Sigma"] = exact_u, exact_S
            c["exact"] = np.array(upe.propagate_GA(exact_u, exact_S))
            c["exact_mean_only"] = np.float64(upe.propagate_mean(exact_u, exact_S))
        for k, v in c.items():
            out[name + "__" + k] = v

    rng = np.random.RandomState(4242)
    a, b = np.meshgrid(np.arange(10.0), np.arange(10.0), indexing="ij")
    xg = np.stack([a.ravel(), b.ravel()], 1)
    tg = np.sin(0.3 * (xg[:, 0] + xg[:, 1])) + 0.1 * rng.randn(100)
    thg = np.log(np.array([2.0, 0.01, 0.04, 0.04]))
    xm = xg[rng.choice(100, 10, replace=False)] + 0.25 * rng.randn(10, 2)
    xs = rng.uniform(0, 9, (23, 2))
    one("grid_m10", xg, tg, thg, xm, xs, exact_u=np.array([4.4, 5.3]), exact_S=np.diag([0.02, 0.01]))

    x, t, xs, th = synth(300, 3, 41, seed_offset=5)
    xm = x[rng.choice(300, 37, replace=False)] + 0.1 * rng.randn(37, 3)
    one("n300_d3_m37", x, t, np.array([0.4, -3.5, -1.2, -0.7, -1.9]), xm, xs)

    x, t, xs, th = synth(700, 4, 150, seed_offset=6)
    xm = rng.uniform(0, 10, (150, 4))
    one("n700_d4_m150", x, t, th, xm, xs, keep_dense=False)
    return out


def generic_operator_cases():
    """The operator interface with operators that are NOT the built-in kernel (plug-in seam #1): (A) a
    from-scratch subclass of the reference's Covariance that implements only __call__ / get_theta -- everything else is the
    reference's generic base class (Covariance.py:137-282) and GaussianProcess (GaussianProcess.py:19-111); (B) a subclass of the
    reference's GaussianCovariance with its own cov_matrix_ij.  Inputs AND outputs are stored; the operators are re-stated in the
    tests (they are test inputs, not reference code)."""
    _install_shims()
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    from skgpuppy.Covariance import Covariance, GaussianCovariance
    from skgpuppy.GaussianProcess import GaussianProcess
    import skgpuppy.UncertaintyPropagation as UP
    out = {}

    class RationalQuadratic(Covariance):
        """k = v (1 + r^2 / (2 a l^2))^-a + vt [xi == xj],  theta = log (v, vt, l, a)"""
        def __call__(self, xi, xj, theta):
            v, vt, ell, a = np.exp(theta)
            diff = np.asarray(xi, dtype=float) - np.asarray(xj, dtype=float)
            r2 = np.dot(diff, diff)
            return v * (1.0 + r2 / (2.0 * a * ell * ell)) ** (-a) + (vt if (np.asarray(xi) == np.asarray(xj)).all() else 0.0)

        def get_theta(self, x, t):
            return np.log(np.array([np.var(t), np.var(t) / 4, 1.0, 1.0]))

    rng = np.random.RandomState(4242)
    n, d, m = 60, 2, 15
    x = rng.uniform(0, 6, (n, d))
    t = np.sin(x[:, 0]) * np.cos(0.7 * x[:, 1]) + 0.05 * rng.randn(n)
    xs = rng.uniform(0, 6, (m, d))
    theta = np.log(np.array([1.3, 0.02, 1.7, 0.8]))
    cov = RationalQuadratic()
    gp = GaussianProcess(x, t, cov, theta.copy())
    out["rq_x"], out["rq_t"], out["rq_xs"], out["rq_theta"] = x, t, xs, theta
    out["rq_K"] = cov.cov_matrix(x, theta)
    out["rq_Kinv"] = np.array(gp.Kinv)
    out["rq_nll"] = np.float64(cov._negativeloglikelihood(x, gp.t, theta))
    out["rq_grad"] = np.array(cov._d_nll_d_theta(x, gp.t, theta))
    out["rq_logdet"] = np.float64(cov._log_det_cov_matrix(x, theta))
    out["rq_pred_mean"], out["rq_pred_var"] = gp.estimate_many(xs)
    out["rq_est0"] = np.array(gp.estimate(xs[0]))
    out["rq_beta"] = np.array(gp._get_beta())
    # UncertaintyPropagationExact talks to the GP through _get_beta / _get_W_inv / _inv_cov_matrix / _covariance only
    # (UncertaintyPropagation.py:269-290, :323-379): it RUNS for this operator -- _get_W_inv reads theta[2:2+d] (here log l, log a) as
    # the ARD weights, C_ux is the operator's own kernel -- and a drop-in must return the same numbers
    upe = UP.UncertaintyPropagationExact(gp)
    out["rq_u"], out["rq_Sigma"] = np.array([3.0, 2.5]), np.array([[0.02, 0.005], [0.005, 0.03]])
    out["rq_exact"] = np.array(upe.propagate_GA(out["rq_u"], out["rq_Sigma"]))
    out["rq_exact_mean_only"] = np.float64(upe.propagate_mean(out["rq_u"], out["rq_Sigma"]))

    class WarpedGaussian(GaussianCovariance):
        """GaussianCovariance whose cross-covariance is modulated by the (positive definite) factor 1 + 0.1 cos(xi_0 - xj_0)"""
        def cov_matrix_ij(self, xi, xj, theta):
            K = GaussianCovariance.cov_matrix_ij(self, xi, xj, theta)
            a = np.asarray(xi, dtype=float)[:, 0][:, None]
            b = np.asarray(xj, dtype=float)[:, 0][None, :]
            return K * (1.0 + 0.1 * np.cos(a - b))

    n, d, m = 120, 3, 20
    x = rng.uniform(0, 10, (n, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(n)
    xs = rng.uniform(0, 10, (m, d))
    theta = np.log(np.array([2.0, 0.01, 0.04, 0.05, 0.03]))
    cov = WarpedGaussian()
    gp = GaussianProcess(x, t, cov, theta.copy())
    out["wg_x"], out["wg_t"], out["wg_xs"], out["wg_theta"] = x, t, xs, theta
    out["wg_K"] = cov.cov_matrix(x, theta)
    out["wg_Kinv"] = np.array(gp.Kinv)
    out["wg_nll"] = np.float64(cov._negativeloglikelihood(x, gp.t, theta))
    out["wg_grad"] = np.array(cov._d_nll_d_theta(x, gp.t, theta))
    out["wg_pred_mean"], out["wg_pred_var"] = gp.estimate_many(xs)
    out["wg_est0"] = np.array(gp.estimate(xs[0]))
    u = np.array([5.0, 4.5, 5.5])
    S = np.array([[0.02, 0.004, 0.0], [0.004, 0.03, -0.002], [0.0, -0.002, 0.01]])
    up = UP.UncertaintyPropagationApprox(gp)
    out["wg_u"], out["wg_Sigma"] = u, S
    out["wg_approx"] = np.array(up.propagate_GA(u, S))
    out["wg_dvh"] = np.array([up._get_variance_dv_h(u, h) for h in range(d)])
    out["wg_factor"] = np.float64(up._getFactor(u, S, 0.02))
    # Exact propagation on the same operator: C_ux comes from the subclass's scalar kernel (the PARENT's __call__: the subclass
    # overrides cov_matrix_ij only), K^-1 from its own matrix; also with u on a training row (the +vt quirk of __call__) and through
    # propagate_mean with a caller-supplied C_ux (UncertaintyPropagation.py:269-276)
    upe = UP.UncertaintyPropagationExact(gp)
    out["wg_exact"] = np.array(upe.propagate_GA(u, S))
    out["wg_exact_mean_only"] = np.float64(upe.propagate_mean(u, S))
    out["wg_exact_on_row7"] = np.array(upe.propagate_GA(x[7].copy(), S))
    out["wg_C_half"] = 0.5 * np.array([gp._covariance(u, x[i]) for i in range(n)])
    out["wg_exact_mean_C_half"] = np.float64(upe.propagate_mean(u, S, out["wg_C_half"]))
    # the quadratic-form helpers with an EXPLICIT Kinv that is not the GP's own (UncertaintyPropagation.py:412-481)
    K2inv = np.linalg.inv(out["wg_K"] + 0.05 * np.eye(n))
    out["wg_K2inv"] = K2inv
    out["wg_sigma2_K2"] = np.float64(up._get_sigma2(u, K2inv, gp.x, up.C_ux, up.J_ux, up.H_ux))
    out["wg_rest_K2"] = np.float64(up._get_variance_rest(u, S, K2inv, gp.x, gp._get_beta(), up.C_ux, up.J_ux, up.H_ux))
    return out


def main():
    if "--generic" in sys.argv:
        os.makedirs(OUT, exist_ok=True)
        np.savez_compressed(os.path.join(OUT, "generic_ops.npz"), **generic_operator_cases())
        print("generic_ops.npz")
        return
    if "--spgp" in sys.argv:
        os.makedirs(OUT, exist_ok=True)
        np.savez_compressed(os.path.join(OUT, "spgp.npz"), **spgp_cases())
        print("spgp.npz")
        return
    GC, GP, UP = _import_reference()
    os.makedirs(OUT, exist_ok=True)

    np.savez_compressed(os.path.join(OUT, "gram.npz"), **gram_cases(GC))
    print("gram.npz")

    # KAT1: float README grid, deterministic target.
    a, b = np.meshgrid(np.arange(10.0), np.arange(10.0), indexing="ij")
    xg = np.stack([a.ravel(), b.ravel()], 1)
    tg = np.sin(0.3 * (xg[:, 0] + xg[:, 1])) + 0.05 * np.cos(1.7 * xg[:, 0] - 0.4 * xg[:, 1])
    thg = np.log(np.array([2.0, 0.01, 0.04, 0.04]))
    xs_g = np.array([[x1 / 2.0, x2 / 2.0] for x1 in range(20) for x2 in range(20)])  # README.rst:120
    xs_g[:4] = [[.5, .5], [4.5, 5], [9, 9], [2.25, 7.75]]
    full = np.array([[0.01, 0.004], [0.004, 0.02]])
    c = gp_case(GC, GP, UP, xg, tg, thg, xs_g,
                us=[[5.0, 5.0], [5.25, 4.75]],
                Sigmas=[np.diag([0.01, 0.01]), np.diag([1.0, 2.0]), full])
    # ML hyper-parameter fit of the README example (GaussianProcess(x, t, cov) without theta): the README draws the
    # targets from the GP itself (README.rst:108-109); the realisation is stored explicitly (never a seed).
    import io, contextlib
    np.random.seed(12345)
    t_real = GP.get_realisation(xg, GC(), thg)
    with contextlib.redirect_stdout(io.StringIO()):
        gp_ml = GP(xg, np.array(t_real), GC())
    c["ml_t_raw"] = np.array(t_real)
    c["ml_theta"] = np.array(gp_ml.theta_min)
    c["ml_theta_start"] = GC().get_theta(xg, np.array(t_real) - np.mean(t_real))
    c["ml_nll"] = np.float64(GC()._negativeloglikelihood(xg, gp_ml.t, gp_ml.theta_min))
    c["ml_nll_start"] = np.float64(GC()._negativeloglikelihood(xg, gp_ml.t, c["ml_theta_start"]))
    c["ml_pred_mean"], c["ml_pred_var"] = gp_ml.estimate_many(xs_g)
    # "next" row f2: inverse uncertainty propagation on the fixed-theta grid GP (cf. skgpuppy/tests/tests.py:361-400)
    from skgpuppy.InverseUncertaintyPropagation import (InverseUncertaintyPropagationApprox as IUPA,
                                                        InverseUncertaintyPropagationNumerical as IUPN)
    gp_fix = GP(xg, np.array(tg), GC(), thg.copy())
    cvec = np.array([4.0, 1.0])
    with contextlib.redirect_stdout(io.StringIO()):
        c["iup_approx"] = IUPA(0.02, gp_fix, np.array([5.25, 4.75]), cvec, 1 / cvec).get_best_solution()
        c["iup_numerical"] = IUPN(0.02, gp_fix, np.array([5.25, 4.75]), cvec, 1 / cvec,
                                  upga_class=UP.UncertaintyPropagationApprox).get_best_solution()
        c["iup_approx_coest"] = IUPA(0.02, gp_fix, np.array([5.25, 4.75]), cvec, np.array([0.25, 2.0]),
                                     coestimated=[[0, 1]]).get_best_solution()
    np.savez_compressed(os.path.join(OUT, "kat1_grid.npz"), **c)
    print("kat1_grid.npz")

    # integer-dtype grid exactly as the README builds it (x stored as given).
    xi = np.array([[x1, x2] for x1 in range(10) for x2 in range(10)])
    c = gp_case(GC, GP, UP, xi, tg, thg, xs_g[:7], us=[[5.0, 5.0]],
                Sigmas=[np.diag([0.01, 0.01])], with_exact=False)
    np.savez_compressed(os.path.join(OUT, "grid_int.npz"), **c)
    print("grid_int.npz")

    # generic d=3 ragged N (not a multiple of any tile), u generic and u == training row.
    x, t, xs, th = synth(203, 3, 37)
    th = np.array([0.4, -3.5, -1.2, -0.7, -1.9])
    Sd = np.diag([0.02, 0.05, 0.01])
    Sf = np.array([[0.02, 0.005, -0.002], [0.005, 0.05, 0.003], [-0.002, 0.003, 0.01]])
    c = gp_case(GC, GP, UP, x, t, th, xs, us=[[4.0, 5.5, 6.0], x[17].copy()], Sigmas=[Sd, Sf])
    np.savez_compressed(os.path.join(OUT, "n203_d3.npz"), **c)
    print("n203_d3.npz")

    # d=8 recipe at N=256 (a tile multiple), M=400; Kinv kept.
    x, t, xs, th = synth(256, 8, 400)
    c = gp_case(GC, GP, UP, x, t, th, xs, us=[[5.0] * 8], Sigmas=[0.01 * np.eye(8)])
    np.savez_compressed(os.path.join(OUT, "n256_d8.npz"), **c)
    print("n256_d8.npz")

    # medium: N=1000 d=4 recipe, M=64 — crosses several 128-blocks; Kinv dropped (size).
    x, t, xs, th = synth(1000, 4, 64)
    c = gp_case(GC, GP, UP, x, t, th, xs, us=[[5.0] * 4], Sigmas=[0.01 * np.eye(4)],
                keep_kinv=False)
    np.savez_compressed(os.path.join(OUT, "n1000_d4.npz"), **c)
    print("n1000_d4.npz")

    # d=16, N=300.
    x, t, xs, th = synth(300, 16, 50)
    c = gp_case(GC, GP, UP, x, t, th, xs, us=[[5.0] * 16], Sigmas=[0.01 * np.eye(16)],
                keep_kinv=False)
    np.savez_compressed(os.path.join(OUT, "n300_d16.npz"), **c)
    print("n300_d16.npz")

    # KAT2 / G6: METIS data held by the reference's own tests (tests/metis_data.py), fixed theta.
    from skgpuppy.tests.metis_data import x as xm, t as tm
    thm = np.array([-0.84329102, -10.77816567, -7.68527421, -5.47537322, 6.78674556])
    um = np.array([15.05, 5.0, 0.025])
    Sm = np.diag([2.0 ** 2, 1.0 ** 2, 0.005 ** 2])
    xs_m = np.vstack([um, xm[:20] * 1.001])
    c = gp_case(GC, GP, UP, np.array(xm, dtype=float), np.array(tm, dtype=float), thm, xs_m,
                us=[um], Sigmas=[Sm], keep_kinv=False, v_out=0.002)
    np.savez_compressed(os.path.join(OUT, "metis.npz"), **c)
    print("metis.npz")


if __name__ == "__main__":
    main()
