"""The operator interface (plug-in seam #1; section 8 a4): GaussianProcess and the likelihood machinery on
operators that are NOT the built-in kernel -- the generic device route (gpx_fit_matrix / gpx_predict_kv / gpx_nll_grad_matrix /
gpx_symv) against golden vectors produced by the genuine reference's generic base classes (tests/golden/generic_ops.npz,
tools/gen_golden.py --generic) and against the oracle's restatement of them (pinned in tests/test_oracle_golden.py).

Tolerances: the reference LU-inverts (scipy inv), the device route Cholesky-factors: SURVEY section 8a's calibrated bounds
(estimate_many rtol 1e-6 / atol 1e-9 v; propagated variances absolute 1e-8 v)."""
import ctypes
import pickle

import numpy as np
import pytest

from conftest import load_golden

import skgpuppy_amd as sk
from skgpuppy_amd import _gpx
from oracle import oracle as orc
from _operators import make_rational_quadratic, make_warped_gaussian

pytestmark = pytest.mark.gpu

# module-level names so that a GaussianProcess on them pickles (pickle finds classes by module + qualified name)
SkRationalQuadratic = make_rational_quadratic(sk.Covariance)
SkRationalQuadratic.__qualname__ = SkRationalQuadratic.__name__ = "SkRationalQuadratic"
SkRationalQuadratic.__module__ = __name__


@pytest.fixture(scope="module")
def g():
    return load_golden("generic_ops")


def test_from_scratch_operator_fit_predict_likelihood(g):
    """a Covariance subclass that implements only __call__ / get_theta: every matrix-sized step is the base class's, on the GPU"""
    cov = SkRationalQuadratic()
    x, t, xs, th = g["rq_x"], g["rq_t"], g["rq_xs"], g["rq_theta"]
    v = float(np.exp(th[0]))
    np.testing.assert_allclose(cov.cov_matrix(x, th), g["rq_K"], rtol=1e-13)
    gp = sk.GaussianProcess(x, t, cov, th.copy())
    assert gp._route() == "generic"
    np.testing.assert_allclose(gp.Kinv, g["rq_Kinv"], rtol=1e-6, atol=1e-7 * np.abs(g["rq_Kinv"]).max())
    np.testing.assert_allclose(cov.inv_cov_matrix(x, th), g["rq_Kinv"], rtol=1e-6, atol=1e-7 * np.abs(g["rq_Kinv"]).max())
    np.testing.assert_allclose(gp._get_beta(), g["rq_beta"], rtol=1e-6, atol=1e-8 * np.abs(g["rq_beta"]).max())
    m, var = gp.estimate_many(xs)
    np.testing.assert_allclose(m, g["rq_pred_mean"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(var, g["rq_pred_var"], rtol=1e-6, atol=1e-9 * v)
    m0, v0 = gp.estimate(xs[0])
    np.testing.assert_allclose([m0, v0], g["rq_est0"], rtol=1e-6, atol=1e-9 * v)
    np.testing.assert_allclose(gp(xs[0]), g["rq_est0"], rtol=1e-6, atol=1e-9 * v)
    # likelihood and its gradient from the operator's own derivative matrices (central differences of __call__, Covariance.py:219-265)
    np.testing.assert_allclose(cov._negativeloglikelihood(x, gp.t, th), g["rq_nll"], rtol=1e-9)
    np.testing.assert_allclose(cov._log_det_cov_matrix(x, th), g["rq_logdet"], rtol=1e-9)
    np.testing.assert_allclose(cov._d_nll_d_theta(x, gp.t, th), g["rq_grad"], rtol=1e-6, atol=1e-7)
    # ... and against the oracle's generic restatement on a different theta (beyond the golden point)
    th2 = th + np.array([0.2, -0.3, 0.1, 0.15])
    ocov = make_rational_quadratic(orc.OracleCovariance)()
    np.testing.assert_allclose(cov._negativeloglikelihood(x, gp.t, th2), ocov._negativeloglikelihood(x, gp.t, th2), rtol=1e-9)
    np.testing.assert_allclose(cov._d_nll_d_theta(x, gp.t, th2), ocov._d_nll_d_theta(x, gp.t, th2), rtol=1e-6, atol=1e-7)
    # Exact propagation RUNS for this operator in the reference (it reads theta[2:4] = log (l, a) as the ARD weights and the operator's
    # own kernel as C_ux: UncertaintyPropagation.py:269-290, :323-379) -- a drop-in returns the same numbers, nonsense variance included
    upe = sk.UncertaintyPropagationExact(gp)
    np.testing.assert_allclose(upe.propagate_GA(g["rq_u"], g["rq_Sigma"]), g["rq_exact"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(upe.propagate_mean(g["rq_u"], g["rq_Sigma"]), g["rq_exact_mean_only"], rtol=0, atol=1e-9)
    # pickling drops the device handle, the round trip predicts the same (reference pickle test, tests.py:626-659)
    gp2 = pickle.loads(pickle.dumps(gp))
    m2, var2 = gp2.estimate_many(xs)
    np.testing.assert_array_equal(m2, m)
    np.testing.assert_array_equal(var2, var)


def test_from_scratch_operator_ml_estimate_runs_on_the_generic_route(g):
    """GaussianProcess(x, t, cov) without theta: ml_estimate = L-BFGS-B on the generic NLL / gradient (Covariance.py:314-337)"""
    cov = make_rational_quadratic(sk.Covariance)()
    ocov = make_rational_quadratic(orc.OracleCovariance)()
    x, t = g["rq_x"][:40], g["rq_t"][:40]
    gp = sk.GaussianProcess(x, t, cov)
    th0 = cov.get_theta(x, gp.t)
    nll0, nll1 = ocov._negativeloglikelihood(x, gp.t, th0), ocov._negativeloglikelihood(x, gp.t, gp.theta_min)
    assert nll1 < nll0 - 1.0, (nll0, nll1)                        # the optimiser made real progress, judged by the ORACLE's likelihood
    assert np.abs(ocov._d_nll_d_theta(x, gp.t, gp.theta_min)).max() < 5e-2 * max(1.0, abs(nll1))


def test_gaussian_subclass_with_own_cross_covariance(g):
    """a GaussianCovariance subclass that overrides cov_matrix_ij must NOT be silently served by the fused kernel"""
    cov = make_warped_gaussian(sk.GaussianCovariance)()
    assert not cov._fused() and sk.GaussianCovariance()._fused()
    x, t, xs, th = g["wg_x"], g["wg_t"], g["wg_xs"], g["wg_theta"]
    v = float(np.exp(th[0]))
    np.testing.assert_allclose(cov.cov_matrix(x, th), g["wg_K"], rtol=1e-12)
    gp = sk.GaussianProcess(x, t, cov, th.copy())
    assert gp._route() == "generic"
    np.testing.assert_allclose(gp.Kinv, g["wg_Kinv"], rtol=1e-5, atol=1e-7 * np.abs(g["wg_Kinv"]).max())
    m, var = gp.estimate_many(xs)
    np.testing.assert_allclose(m, g["wg_pred_mean"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(var, g["wg_pred_var"], rtol=1e-6, atol=1e-9 * v)
    np.testing.assert_allclose(gp.estimate(xs[0]), g["wg_est0"], rtol=1e-6, atol=1e-9 * v)
    np.testing.assert_allclose(cov._negativeloglikelihood(x, gp.t, th), g["wg_nll"], rtol=1e-9)
    np.testing.assert_allclose(cov._d_nll_d_theta(x, gp.t, th), g["wg_grad"], rtol=1e-6, atol=1e-7)
    # the fused path on the same data gives DIFFERENT numbers: the override matters
    mf, _vf = sk.GaussianProcess(x, t, sk.GaussianCovariance(), th.copy()).estimate_many(xs)
    assert np.abs(mf - m).max() > 1e-4
    # Approx propagation on the generic route: C / J / H from the operator's own scalar methods, K^-1 products on the device
    u, S = g["wg_u"], g["wg_Sigma"]
    up = sk.UncertaintyPropagationApprox(gp)
    ma, va = up.propagate_GA(u, S)
    np.testing.assert_allclose(ma, g["wg_approx"][0], rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(va, g["wg_approx"][1], rtol=0, atol=1e-8 * v)
    np.testing.assert_allclose([up._get_variance_dv_h(u, h) for h in range(3)], g["wg_dvh"], rtol=0, atol=1e-8 * v)
    np.testing.assert_allclose(up._getFactor(u, S, 0.02), g["wg_factor"], rtol=1e-5)
    # Exact propagation on the generic route (round 4 refused it): the reference's class reads the GP through _get_beta / _get_W_inv /
    # _inv_cov_matrix / _covariance only (UncertaintyPropagation.py:269-290, :323-379) and returns these numbers for this operator --
    # C_ux from the subclass's scalar kernel on the host, the N and N^2 sums on the device (gpx_propagate_exact_matrix)
    upe = sk.UncertaintyPropagationExact(gp)
    me, ve = upe.propagate_GA(u, S)
    np.testing.assert_allclose(me, g["wg_exact"][0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(ve, g["wg_exact"][1], rtol=0, atol=1e-8 * v)
    np.testing.assert_allclose(upe.propagate_mean(u, S), g["wg_exact_mean_only"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(upe.propagate_mean(u, S, g["wg_C_half"]), g["wg_exact_mean_C_half"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(upe.propagate_GA(x[7].copy(), S), g["wg_exact_on_row7"], rtol=0, atol=1e-8 * v)   # u on a training row: the +vt quirk of __call__
    assert upe.LambdaInv.shape == (3, 3) and upe.normalize_C_corr2 > 0          # the attributes the reference's call leaves behind


@pytest.mark.parametrize("route", ["generic", "fused"])
def test_quadratic_form_helpers_honour_an_explicit_kinv(g, route):
    """_get_sigma2 / _get_variance_rest with a Kinv that is NOT the fitted model's own (UncertaintyPropagation.py:412-481): the
    reference loops over whatever matrix it is handed; round 3 ignored the argument"""
    x, t, th = g["wg_x"], g["wg_t"], g["wg_theta"]
    u, S = g["wg_u"], g["wg_Sigma"]
    v = float(np.exp(th[0]))
    if route == "generic":
        gp = sk.GaussianProcess(x, t, make_warped_gaussian(sk.GaussianCovariance)(), th.copy())
        K2inv, want_s2, want_rest = g["wg_K2inv"], g["wg_sigma2_K2"], g["wg_rest_K2"]
        up = sk.UncertaintyPropagationApprox(gp)
        up.propagate_GA(u, S)
        beta = gp._get_beta()
    else:
        gp = sk.GaussianProcess(x, t, sk.GaussianCovariance(), th.copy())
        og = orc.OracleGP(x, t, th)
        K2inv = np.linalg.inv(orc.gram(x, th) + 0.05 * np.eye(len(x)))
        up = sk.UncertaintyPropagationApprox(gp)
        up.propagate_GA(u, S)
        beta = gp._get_beta()
        # the oracle's loops on the explicit matrix
        class _G(object):
            pass
        o2 = _G()
        o2.Kinv, o2.n, o2.d, o2.theta_min, o2.x = K2inv, og.n, og.d, og.theta_min, og.x
        o2.beta = lambda: np.asarray(beta)
        _m, want_s2, want_rest = orc.approx_parts(o2, u, S)
    s2 = up._get_sigma2(u, K2inv, gp.x, up.C_ux, up.J_ux, up.H_ux)
    rest = up._get_variance_rest(u, S, K2inv, gp.x, beta, up.C_ux, up.J_ux, up.H_ux)
    np.testing.assert_allclose(s2, want_s2, rtol=0, atol=1e-8 * v)
    np.testing.assert_allclose(rest, want_rest, rtol=0, atol=1e-8 * v)
    # with the model's own Kinv (the way the reference itself calls them) the device cache answers -- same as propagate_GA's parts
    s2o, resto = up._get_sigma2_and_variance_rest(u, S, gp.Kinv, gp.x, None)
    _mean, var = up.propagate_GA(u, S)
    np.testing.assert_allclose(s2o + resto, var, rtol=0, atol=1e-12)
    assert abs(s2o - s2) > 1e-6                                   # ... and it is a different number from the explicit matrix's


def test_matrix_handle_abi_jitter_and_state_errors(g):
    """gpx_fit_matrix directly: the +1e-5 I retry of the base-class inv_cov_matrix (Covariance.py:180-185) on a singular matrix, and
    GPX_ERR_STATE from the entry points that would need the built-in kernel's inputs"""
    n = 200
    rng = np.random.RandomState(3)
    B = rng.randn(n, 5)
    K = B.dot(B.T)                                                # rank 5: not positive definite without the jitter
    t = rng.randn(n)
    h = ctypes.c_void_p()
    _gpx.check(_gpx.lib.gpx_fit_matrix(_gpx.ptr(K), _gpx.ptr(t), n, None, ctypes.byref(h)), "gpx_fit_matrix")
    try:
        jit = ctypes.c_double()
        _gpx.check(_gpx.lib.gpx_jitter_used(h, ctypes.byref(jit)), "gpx_jitter_used")
        assert jit.value == 1e-5
        alpha = np.empty(n)
        _gpx.check(_gpx.lib.gpx_alpha(h, _gpx.ptr(alpha)), "gpx_alpha")
        want = np.linalg.solve(K + 1e-5 * np.eye(n), t)
        np.testing.assert_allclose(alpha, want, rtol=1e-6, atol=1e-6 * np.abs(want).max())
        nn, dd = ctypes.c_int64(), ctypes.c_int()
        _gpx.check(_gpx.lib.gpx_n(h, ctypes.byref(nn), ctypes.byref(dd)), "gpx_n")
        assert (nn.value, dd.value) == (n, 0)
        xs = np.zeros((3, 2))
        mean, var = np.empty(3), np.empty(3)
        assert _gpx.lib.gpx_predict(h, _gpx.ptr(xs), 3, _gpx.ptr(mean), _gpx.ptr(var)) == _gpx.GPX_ERR_STATE
        assert "supplied matrix" in _gpx.last_error()
        out = [ctypes.c_double() for _ in range(4)]
        assert _gpx.lib.gpx_propagate_approx(h, _gpx.ptr(xs[0]), _gpx.ptr(np.eye(2)), *[ctypes.byref(o) for o in out]) == _gpx.GPX_ERR_STATE
        # predict_kv on ragged sizes: m = 1 and m = 131 (crosses a 128-row tile), against numpy on the jittered matrix
        Kj = K + 1e-5 * np.eye(n)
        for m in (1, 131):
            kv = rng.randn(m, n)
            kd = rng.rand(m) + 5.0
            mean, var = np.empty(m), np.empty(m)
            _gpx.check(_gpx.lib.gpx_predict_kv(h, _gpx.ptr(kv), m, _gpx.ptr(kd), _gpx.ptr(mean), _gpx.ptr(var)), "gpx_predict_kv")
            sol = np.linalg.solve(Kj, kv.T)
            np.testing.assert_allclose(mean, kv.dot(want), rtol=1e-6, atol=1e-6 * np.abs(kv.dot(want)).max())
            np.testing.assert_allclose(var, kd - np.einsum("ij,ji->i", kv, sol), rtol=1e-5, atol=1e-5 * np.abs(np.einsum("ij,ji->i", kv, sol)).max())
    finally:
        _gpx.lib.gpx_free(h)
    # an indefinite matrix stays an error after the retry: LinAlgError through the Python layer
    with pytest.raises(np.linalg.LinAlgError):
        sk.Covariance().inv_cov_matrix(None, None, cov_matrix=-np.eye(4))
    bad = np.eye(150)
    bad[7, 7] = -1.0
    hb = ctypes.c_void_p()
    assert _gpx.lib.gpx_fit_matrix(_gpx.ptr(bad), _gpx.ptr(np.zeros(150)), 150, None, ctypes.byref(hb)) > 0


def test_predict_kv_on_a_fused_handle_equals_predict():
    """gpx_predict_kv is valid on every handle: fed the built-in kernel's own cross-covariance it reproduces gpx_predict"""
    rng = np.random.RandomState(11)
    N, d, M = 1500, 4, 300
    x = rng.uniform(0, 10, (N, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
    xs = rng.uniform(0, 10, (M, d))
    th = np.log(np.array([2.0, 0.01] + [0.04] * d))
    cov = sk.GaussianCovariance()
    gp = sk.GaussianProcess(x, t, cov, th.copy())
    m1, v1 = gp.estimate_many(xs)
    kv = cov.cov_matrix_ij(xs, x, th)
    kd = np.full(M, np.exp(th[0]) + np.exp(th[1]))
    mean, var = np.empty(M), np.empty(M)
    _gpx.check(_gpx.lib.gpx_predict_kv(gp._dev().handle, _gpx.ptr(kv), M, _gpx.ptr(kd), _gpx.ptr(mean), _gpx.ptr(var)), "gpx_predict_kv")
    np.testing.assert_allclose(mean + gp.meant, m1, rtol=0, atol=1e-12)
    np.testing.assert_allclose(var, v1, rtol=0, atol=1e-12)


def test_exact_matrix_entry_rejects_bad_arguments(g):
    """gpx_propagate_exact_matrix: a handle of another size, missing arrays, a non-positive weight -> GPX_ERR_BAD_ARG (ValueError)"""
    import ctypes
    x, t, th = g["wg_x"], g["wg_t"], g["wg_theta"]
    gp = sk.GaussianProcess(x, t, make_warped_gaussian(sk.GaussianCovariance)(), th.copy())
    n, d = x.shape
    xx, w, C = _gpx.f64(x), np.exp(th[2:2 + d]), np.ones(n)
    u, S = _gpx.f64(g["wg_u"]), _gpx.f64(g["wg_Sigma"])
    m, v = ctypes.c_double(), ctypes.c_double()
    call = lambda h, nn, ww, KK=None, bb=None: _gpx.lib.gpx_propagate_exact_matrix(h, KK, bb, _gpx.ptr(xx), nn, d, _gpx.ptr(_gpx.f64(ww)), _gpx.ptr(C),
                                                                                   _gpx.ptr(u), _gpx.ptr(S), 1.0, ctypes.byref(m), ctypes.byref(v))
    assert call(gp._dev().handle, n, w) == 0
    assert call(gp._dev().handle, n - 1, w) == _gpx.GPX_ERR_BAD_ARG          # the handle holds another n
    assert call(None, n, w) == _gpx.GPX_ERR_BAD_ARG                          # no handle and no explicit Kinv / beta
    assert call(gp._dev().handle, n, np.concatenate([[0.0], w[1:]])) == _gpx.GPX_ERR_BAD_ARG
    # explicit K^-1 / beta give the handle's numbers
    K, b = _gpx.f64(gp.Kinv), _gpx.f64(gp._get_beta())
    m0, v0 = m.value, v.value
    assert call(gp._dev().handle, n, w) == 0
    m0, v0 = m.value, v.value
    assert call(None, n, w, _gpx.ptr(K), _gpx.ptr(b)) == 0
    assert m.value == pytest.approx(m0, abs=1e-10) and v.value == pytest.approx(v0, abs=1e-9)
