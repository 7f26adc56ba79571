"""Pin the oracle (CPU restatement) against golden vectors generated from the genuine reference
(tools/gen_golden.py) and the known answers KAT1 / KAT2 of SURVEY.md section 8c.  CPU only."""
import numpy as np
import pytest

from conftest import GP_CASES, load_golden
from oracle import oracle as orc

GRAM_CASES = ["grid_int", "n257_d5", "n64_d16", "rect_33x257_d5", "n257_d5_vt0", "n130_d1"]


@pytest.mark.parametrize("name", GRAM_CASES)
def test_gram_matches_reference(name):
    g = load_golden("gram")
    xi, xj, th = g[name + "__xi"], g[name + "__xj"], g[name + "__theta"]
    with np.errstate(divide="ignore"):
        K = orc.gram_ij(xi, xj, th)
        np.testing.assert_allclose(K, g[name + "__K_ij"], rtol=1e-14, atol=0)
        if name + "__K" in g:
            np.testing.assert_allclose(orc.gram(xi, th), g[name + "__K"], rtol=1e-14, atol=0)


def test_scalar_cov_quirk():
    th = np.log(np.array([2.0, 0.01, 0.04, 0.04]))
    a = np.array([5.0, 5.0])
    assert orc.scalar_cov(a, a.copy(), th) == pytest.approx(2.01, abs=1e-15)   # +vt on equality
    assert orc.scalar_cov(a, a + 1e-9, th) < 2.0 + 1e-12                      # no vt otherwise


@pytest.fixture(scope="module", params=GP_CASES)
def case(request):
    g = load_golden(request.param)
    gp = orc.OracleGP(g["x"], g["t_raw"], g["theta"])
    return request.param, g, gp


def test_fit(case):
    name, g, gp = case
    assert gp.meant == pytest.approx(float(g["meant"]), abs=1e-15)
    np.testing.assert_allclose(gp.t, g["t_centered"], rtol=0, atol=1e-15)
    scale = 1e-7 if name == "metis" else 1e-9      # LU inverse at cond 1.6e7 / <=1e6
    if "Kinv" in g:
        np.testing.assert_allclose(gp.Kinv, g["Kinv"], rtol=scale, atol=scale * np.abs(g["Kinv"]).max())
    np.testing.assert_allclose(gp.beta(), g["beta"], rtol=1e-7, atol=1e-7 * np.abs(g["beta"]).max())
    assert gp.logdet() == pytest.approx(float(g["logdet"]), rel=1e-10)


def test_estimate_many(case):
    name, g, gp = case
    v = np.exp(g["theta"][0])
    mean, var = gp.estimate_many(g["xs"])
    np.testing.assert_allclose(mean, g["pred_mean"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(var, g["pred_var"], rtol=1e-9, atol=1e-10 * v)
    m0, v0 = gp.estimate(g["xs"][0])
    assert m0 == pytest.approx(g["est0"][0], abs=1e-10)
    assert v0 == pytest.approx(g["est0"][1], abs=1e-10 * v)


def test_propagation(case):
    name, g, gp = case
    v = np.exp(g["theta"][0])
    tol = 1e-9 * v
    for iu in range(int(g["nu"])):
        u = g["u%d" % iu]
        C, J, H = orc.cjh(gp, u)
        np.testing.assert_allclose(C, g["C_ux_u%d" % iu], rtol=1e-14, atol=0)
        np.testing.assert_allclose(J, g["J_ux_u%d" % iu], rtol=1e-13, atol=1e-300)
        np.testing.assert_allclose(H, g["H_ux_u%d" % iu], rtol=1e-13, atol=1e-300)
        dv = np.array([orc.approx_dvh(gp, u, h, (C, J, H)) for h in range(gp.d)])
        np.testing.assert_allclose(dv, g["dvh_u%d" % iu], rtol=1e-7, atol=tol * 10)
        for iS in range(int(g["nS"])):
            S = g["Sigma%d" % iS]
            ma, va = orc.approx_propagate(gp, u, S, (C, J, H))
            ref = g["approx_u%d_S%d" % (iu, iS)]
            assert ma == pytest.approx(ref[0], abs=1e-10)
            assert va == pytest.approx(ref[1], abs=tol)
            f = orc.approx_factor(gp, u, S, float(g["v_out"]), (C, J, H))
            assert f == pytest.approx(float(g["factor_u%d_S%d" % (iu, iS)]), rel=1e-6)
            key = "exact_u%d_S%d" % (iu, iS)
            if key in g:
                me, ve = orc.exact_propagate(gp, u, S)
                assert me == pytest.approx(g[key][0], abs=1e-10)
                assert ve == pytest.approx(g[key][1], abs=tol)
                mo = orc.exact_mean(gp, u, S)
                assert mo == pytest.approx(float(g["exact_mean_only_u%d_S%d" % (iu, iS)]), abs=1e-10)
                # propagate_mean(u, Sigma, C_ux) with a caller-supplied C_ux (UncertaintyPropagation.py:269-290)
                mc = orc.exact_mean(gp, u, S, C=g["C_alt_u%d" % iu])
                assert mc == pytest.approx(float(g["exact_mean_C_alt_u%d_S%d" % (iu, iS)]), abs=1e-10)


def test_kat1_known_answers():
    """SURVEY.md 8c KAT1 literals (obtained from the reference during the survey)."""
    g = load_golden("kat1_grid")
    gp = orc.OracleGP(g["x"], g["t_raw"], g["theta"])
    assert gp.meant == pytest.approx(0.19262764562720766, abs=1e-15)
    mean, var = gp.estimate_many([[.5, .5], [4.5, 5], [9, 9], [2.25, 7.75]])
    np.testing.assert_allclose(mean, [0.3141169314449207, 0.28604463521448886, -0.7801159965052474,
                                      0.1491755086576004], atol=1e-11)
    np.testing.assert_allclose(var, [0.01187314685897833, 0.0108845405331639, 0.01498150754166194,
                                     0.01115813990257708], atol=1e-11)
    assert gp.Kinv[0, 0] == pytest.approx(50.18492458367548, rel=1e-10)
    assert gp.Kinv[0, 1] == pytest.approx(-25.607400325654574, rel=1e-10)
    assert gp.Kinv[55, 55] == pytest.approx(91.17198500797619, rel=1e-10)
    S = np.diag([0.01, 0.01])
    u = np.array([5.0, 5.0])                                  # equals a training row: quirk active
    assert orc.approx_propagate(gp, u, S) == pytest.approx((0.18983893618073702, 0.0018348329974112298), abs=1e-10)
    assert orc.exact_propagate(gp, u, S) == pytest.approx((0.18981740098493302, 0.0018392233503982257), abs=1e-10)
    dv = [orc.approx_dvh(gp, u, h) for h in range(2)]
    assert dv == pytest.approx([0.09267874425295342, 0.09080455550079503], abs=1e-9)
    assert orc.approx_factor(gp, u, np.diag([1.0, 2.0]), 0.02) == pytest.approx(0.07291609751209138, rel=1e-7)
    u = np.array([5.25, 4.75])
    assert orc.approx_propagate(gp, u, S) == pytest.approx((0.13524300017798102, 0.01271034188082463), abs=1e-10)
    assert orc.exact_propagate(gp, u, S) == pytest.approx((0.13524304792303363, 0.012706742317055097), abs=1e-10)


def test_kat2_metis_known_answers():
    """SURVEY.md 8c KAT2 (METIS data held by the reference's tests, fixed theta)."""
    g = load_golden("metis")
    gp = orc.OracleGP(g["x"], g["t_raw"], g["theta"])
    u = np.array([15.05, 5.0, 0.025])
    S = np.diag([4.0, 1.0, 2.5e-5])
    m, v = gp.estimate(u)
    assert m == pytest.approx(0.3037105446529179, abs=1e-8)
    assert v == pytest.approx(2.1113158520658093e-05, abs=1e-8)
    ma, va = orc.approx_propagate(gp, u, S)
    assert (ma, va) == pytest.approx((0.30309549405529035, 0.0016887385637408547), abs=1e-8)
    me, ve = orc.exact_propagate(gp, u, S)
    assert (me, ve) == pytest.approx((0.3030869237925962, 0.0017031624981441612), abs=1e-8)
    # the reference's own METIS assertions (tests.py:1381-1409): code uncertainty < 6e-4 and
    # sqrt(var - code_u) of both propagators inside the Monte-Carlo confidence interval
    code_u = v - np.exp(g["theta"][1])
    assert np.sqrt(code_u) < 0.0006
    for var in (va, ve):
        assert 0.0410788036621 < np.sqrt(var - code_u) < 0.0422334526251


def test_nll_and_gradient(case):
    """"next" row f1: likelihood and gradient of the oracle vs the reference (Covariance.py:197-216, :266-282)."""
    name, g, gp = case
    for tag in ("", "_p"):
        th = g["theta" + tag + "_used"]
        ref, refg = float(g["nll" + tag]), g["nll_grad" + tag]
        loose = 1e-5 if name == "metis" else 1e-8
        assert orc.nll(g["x"], gp.t, th) == pytest.approx(ref, rel=loose, abs=loose * 10)
        np.testing.assert_allclose(orc.nll_grad(g["x"], gp.t, th), refg, rtol=max(loose, 1e-7) * 10, atol=1e-6 * max(1.0, np.abs(refg).max()))


def test_theta_start():
    g = load_golden("kat1_grid")
    t = g["ml_t_raw"] - g["ml_t_raw"].mean()
    np.testing.assert_allclose(orc.theta_start(g["x"], t), g["ml_theta_start"], rtol=1e-14)


# ------------------------------------------------------------------------------------------------
# "next" row f3: SPGP restatement vs the reference (tests/golden/spgp.npz, tools/gen_golden.py --spgp)
# ------------------------------------------------------------------------------------------------
SPGP_CASES = ["grid_m10", "n300_d3_m37", "n700_d4_m150"]


def spgp_case(name):
    g = load_golden("spgp")
    pre = name + "__"
    return {k[len(pre):]: v for k, v in g.items() if k.startswith(pre)}


@pytest.mark.parametrize("name", SPGP_CASES)
def test_spgp_oracle_matches_reference(name):
    g = spgp_case(name)
    x, t, th, m, xs = g["x"], g["t_raw"], g["theta"], int(g["m"]), g["xs"]
    if "cov" in g:
        np.testing.assert_allclose(orc.spgp_cov_matrix(x, th, m), g["cov"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(orc.spgp_inv_cov_matrix(x, th, m), g["inv"], rtol=0, atol=1e-9 * np.abs(g["inv"]).max())
        # the reference's own check (tests.py:521-530): Woodbury inverse == inv(cov_matrix)
        assert np.abs(np.linalg.inv(g["cov"]) - g["inv"]).sum() <= 1e-5 * max(1.0, np.abs(g["inv"]).sum())
    np.testing.assert_allclose(orc.spgp_cov_matrix_ij(xs[:16], x, th, m), g["cross"], rtol=0, atol=1e-12)
    tc = t - t.mean()
    assert orc.spgp_nll(x, tc, th, m) == pytest.approx(float(g["nll_snelson"]), rel=1e-9)
    # the chunked form used by the full-size config-5 GPU test is the same number (chunk smaller than N on purpose)
    assert orc.spgp_nll_chunked(x, tc, th, m, chunk=97) == pytest.approx(float(g["nll_snelson"]), rel=1e-9)
    assert orc.spgp_generic_nll(x, tc, th, m) == pytest.approx(float(g["nll_generic"]), rel=1e-6)
    gp = orc.OracleSPGP(x, t, th, m)
    mu, var = gp.estimate_many(xs)
    np.testing.assert_allclose(mu, g["pred_mean"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(var, g["pred_var"], rtol=0, atol=2e-7)
    assert orc.spgp_scalar_cov(x[0], x[1], th, m) == pytest.approx(float(np.ravel(g["scalar_01"])[0]), abs=1e-11)
    assert orc.spgp_scalar_cov(x[0], x[0], th, m) == pytest.approx(float(np.ravel(g["scalar_00"])[0]), abs=1e-11)
    if "exact" in g:
        # UncertaintyPropagationExact on the SPGP model (the class reads the GP through beta / W_inv / Kinv / _covariance only)
        u, S = g["exact_u"], g["exact_Sigma"]
        kern = lambda a, b: orc.spgp_scalar_cov(a, b, th, m)
        np.testing.assert_allclose(orc.exact_propagate_operator(gp, kern, u, S), np.ravel(g["exact"]), rtol=0, atol=1e-8)
        assert orc.exact_propagate_operator(gp, kern, u, S, mean_only=True) == pytest.approx(float(g["exact_mean_only"]), abs=1e-8)


def test_spgp_reference_nll_agreement():
    """the reference's test_spgp_nll (tests.py:768-804): Snelson's likelihood within 2e-1 of the dense one."""
    for name in SPGP_CASES[:2]:
        g = spgp_case(name)
        assert abs(float(g["nll_snelson"]) - float(g["nll_generic"])) < 2e-1


def test_spgp_analytic_gradient_matches_central_differences():
    """oracle.spgp_nll_grad (the O(N M^2) analytic gradient the GPU path is checked against) vs central differences of
    oracle.spgp_nll, which the reference's golden likelihood values pin (Covariance.py:981-1019).  The reference's own
    gradient (:906-979) cannot run on Python 3: gradient parity unpinned, this is the substitute."""
    rng = np.random.RandomState(1)
    for (N, d, m) in ((60, 2, 5), (200, 3, 12)):
        x = rng.uniform(0, 10, (N, d))
        t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
        t = t - t.mean()
        xb = x[rng.choice(N, m, replace=False)] + 0.1 * rng.randn(m, d)
        theta = np.concatenate([np.log([1.7, 0.02]), np.log(rng.uniform(0.02, 0.08, d)), xb.ravel()])
        g = orc.spgp_nll_grad(x, t, theta, m)
        fd = np.empty_like(g)
        for j in range(len(theta)):
            e = np.zeros(len(theta))
            e[j] = 1e-6
            fd[j] = (orc.spgp_nll(x, t, theta + e, m) - orc.spgp_nll(x, t, theta - e, m)) / 2e-6
        np.testing.assert_allclose(g, fd, rtol=0, atol=5e-7 * np.abs(g).max())


def test_oracle_derivative_grams_against_reference_vectors():
    """oracle.d_gram_d_theta (the building block of the oracle's likelihood gradient) against the reference's
    _d_cov_matrix_d_theta outputs (Covariance.py:505-512, :605-657); log det against _log_det_cov_matrix (:189-195)."""
    g = load_golden("gram")
    for name in ("n257_d5", "n130_d1"):
        x, th = g[name + "__xi"], g[name + "__theta"]
        if name == "n257_d5":
            x = x[:130]
        for j in (0, 1, len(th) - 1):
            want = g["%s__dK_%d" % (name, j)]
            np.testing.assert_allclose(orc.d_gram_d_theta(x, th, j), want, rtol=1e-10, atol=1e-13 * np.abs(want).max() + 1e-300)
        og = orc.OracleGP(x, np.zeros(len(x)), th)
        assert abs(og.logdet() - float(g[name + "__logdet"])) < 1e-8 * max(1.0, abs(float(g[name + "__logdet"])))


# ------------------------------------------------------------------------------------------------
# the GENERIC operator interface of the oracle (OracleCovariance / OracleOperatorGP) against the genuine reference's base classes
# ------------------------------------------------------------------------------------------------
def test_generic_operator_oracle_against_reference_golden():
    from _operators import make_rational_quadratic, make_warped_gaussian
    g = load_golden("generic_ops")
    # (A) from-scratch operator: only __call__ / get_theta
    cov = make_rational_quadratic(orc.OracleCovariance)()
    x, t, xs, th = g["rq_x"], g["rq_t"], g["rq_xs"], g["rq_theta"]
    np.testing.assert_allclose(cov.cov_matrix(x, th), g["rq_K"], rtol=1e-13)
    gp = orc.OracleOperatorGP(x, t, cov, th)
    np.testing.assert_allclose(gp.Kinv, g["rq_Kinv"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(gp.beta(), g["rq_beta"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(cov._negativeloglikelihood(x, gp.t, th), g["rq_nll"], rtol=1e-12)
    np.testing.assert_allclose(cov._log_det_cov_matrix(x, th), g["rq_logdet"], rtol=1e-12)
    np.testing.assert_allclose(cov._d_nll_d_theta(x, gp.t, th), g["rq_grad"], rtol=1e-8, atol=1e-9)
    m, v = gp.estimate_many(xs)
    np.testing.assert_allclose(m, g["rq_pred_mean"], rtol=1e-10, atol=1e-11)
    np.testing.assert_allclose(v, g["rq_pred_var"], rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(gp.estimate(xs[0]), g["rq_est0"], rtol=1e-8, atol=1e-11)
    # Exact propagation runs for this operator in the reference (it reads theta[2:4] = log (l, a) as ARD weights): same numbers
    kern = lambda a, b: cov(a, b, th)
    np.testing.assert_allclose(orc.exact_propagate_operator(gp, kern, g["rq_u"], g["rq_Sigma"]), g["rq_exact"], rtol=0, atol=1e-9)
    assert orc.exact_propagate_operator(gp, kern, g["rq_u"], g["rq_Sigma"], mean_only=True) == pytest.approx(float(g["rq_exact_mean_only"]), abs=1e-10)
    # (B) GaussianCovariance subclass with its own cov_matrix_ij
    cov = make_warped_gaussian(orc.OracleGaussianCovariance)()
    x, t, xs, th = g["wg_x"], g["wg_t"], g["wg_xs"], g["wg_theta"]
    np.testing.assert_allclose(cov.cov_matrix(x, th), g["wg_K"], rtol=1e-12)
    gp = orc.OracleOperatorGP(x, t, cov, th)
    np.testing.assert_allclose(gp.Kinv, g["wg_Kinv"], rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(cov._negativeloglikelihood(x, gp.t, th), g["wg_nll"], rtol=1e-11)
    np.testing.assert_allclose(cov._d_nll_d_theta(x, gp.t, th), g["wg_grad"], rtol=1e-7, atol=1e-8)
    m, v = gp.estimate_many(xs)
    np.testing.assert_allclose(m, g["wg_pred_mean"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(v, g["wg_pred_var"], rtol=1e-7, atol=1e-10)
    u, S = g["wg_u"], g["wg_Sigma"]
    cache = gp.cjh(u)
    np.testing.assert_allclose(orc.approx_propagate(gp, u, S, cache), g["wg_approx"], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose([orc.approx_dvh(gp, u, h, cache) for h in range(3)], g["wg_dvh"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(orc.approx_factor(gp, u, S, 0.02, cache), g["wg_factor"], rtol=1e-8)
    # Exact propagation on the subclass: C_ux from its scalar kernel (the parent's), K^-1 from its own matrix
    kern = lambda a, b: cov(a, b, th)
    np.testing.assert_allclose(orc.exact_propagate_operator(gp, kern, u, S), g["wg_exact"], rtol=0, atol=1e-9)
    assert orc.exact_propagate_operator(gp, kern, u, S, mean_only=True) == pytest.approx(float(g["wg_exact_mean_only"]), abs=1e-10)
    np.testing.assert_allclose(orc.exact_propagate_operator(gp, kern, x[7].copy(), S), g["wg_exact_on_row7"], rtol=0, atol=1e-9)
    assert orc.exact_propagate_operator(gp, kern, u, S, C=g["wg_C_half"], mean_only=True) == pytest.approx(float(g["wg_exact_mean_C_half"]), abs=1e-10)
