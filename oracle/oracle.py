"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement (numpy / scipy + the serial C loops of oracle/loops.c) of the one hot path of
snphbaum/scikit-gpuppy that this repository accelerates: ARD squared-exponential Gram matrix ->
explicit inverse -> estimate_many -> Girard uncertainty propagation (Approx and Exact).

It is the *checker* for the HIP path and the timed CPU baseline of bench.py (`cpu_baseline.kind =
"port"`).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the
product package (scikit-gpuppy_amd/) never does.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against golden
vectors produced by importing the genuine reference in the build container (tools/gen_golden.py ->
tests/golden/*.npz) and against the known answers KAT1/KAT2 of SURVEY.md section 8c.  One exception, stated
at the function: spgp_nll_grad (analytic SPGP likelihood gradient) -- the reference's own gradient does not run on
Python 3, so it is pinned to central differences of the golden-pinned spgp_nll: gradient parity unpinned.

The algorithm class is deliberately the reference's (so that timing it is not a straw man):
GEMM-expansion Gram with full N1xN2 temporaries, LU `scipy.linalg.inv`, GEMM-based estimate_many
including the MxM products, explicit-Kinv serial double sums.  Citations are file:line into the
reference tree.
"""
import ctypes
import os
import subprocess

import numpy as np
from scipy.linalg import cholesky, inv, solve_triangular

_HERE = os.path.dirname(os.path.abspath(__file__))
_LOOPS = None


def build_loops(force=False):
    """Compile oracle/loops.c with gcc (no fast-math) into oracle/liboracle_loops.so."""
    so = os.path.join(_HERE, "liboracle_loops.so")
    src = os.path.join(_HERE, "loops.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return so


def _loops():
    global _LOOPS
    if _LOOPS is None:
        lib = ctypes.CDLL(build_loops())
        dp = ctypes.POINTER(ctypes.c_double)
        L = ctypes.c_long
        lib.orc_quad_form.restype = ctypes.c_double
        lib.orc_quad_form.argtypes = [dp, dp, L]
        lib.orc_var2.restype = ctypes.c_double
        lib.orc_var2.argtypes = [dp, dp, dp, dp, L, L]
        lib.orc_var3.restype = ctypes.c_double
        lib.orc_var3.argtypes = [dp, dp, dp, L]
        lib.orc_dvh2.restype = ctypes.c_double
        lib.orc_dvh2.argtypes = [dp, dp, dp, L, L, L]
        lib.orc_exact_sum.restype = ctypes.c_double
        lib.orc_exact_sum.argtypes = [dp, dp, dp, dp, dp, dp, ctypes.c_double, L, L]
        _LOOPS = lib
    return _LOOPS


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


# --------------------------------------------------------------------------------------------
# L1: covariance operator  (reference: skgpuppy/Covariance.py, class GaussianCovariance)
# --------------------------------------------------------------------------------------------

def unpack_theta(theta):
    """theta = (log v, log vt, log w_1..w_d)   (Covariance.py:442-445, :467-470)."""
    with np.errstate(divide="ignore"):
        theta = np.asarray(theta, dtype=float)
        return np.exp(theta[0]), np.exp(theta[1]), np.exp(theta[2:])


def scalar_cov(xi, xj, theta):
    """k(xi,xj) = v exp(-1/2 sum_k w_k (xi_k-xj_k)^2) + vt iff xi == xj elementwise
    (Covariance.py:440-451, the "slightly dirty hack" at :450-451)."""
    v, vt, w = unpack_theta(theta)
    xi = np.asarray(xi)
    xj = np.asarray(xj)
    diff = xi - xj
    return v * np.exp(-0.5 * np.dot(diff, w * diff)) + (vt if (xi == xj).all() else 0.0)


def gram_ij(xi, xj, theta):
    """N1 x N2 cross-covariance WITHOUT the noise term, via the ||a||^2+||b||^2-2ab expansion on
    sqrt(w)-scaled inputs (Covariance.py:466-483)."""
    v, _vt, w = unpack_theta(theta)
    sw = np.sqrt(w)
    a = np.array(xi, dtype=float) * sw
    b = np.array(xj, dtype=float) * sw
    n1, n2 = a.shape[0], b.shape[0]
    D = -2.0 * np.dot(a, b.T)
    D += np.tile((b * b).sum(1)[None, :], (n1, 1))
    D += np.tile((a * a).sum(1)[:, None], (1, n2))
    return v * np.exp(-0.5 * D)


def gram(x, theta):
    """K = gram_ij(x,x) + vt I  (Covariance.py:461-464)."""
    _v, vt, _w = unpack_theta(theta)
    return gram_ij(x, x, theta) + vt * np.eye(len(x))


def inv_cov_matrix(x, theta, cov_matrix=None):
    """LU inverse of K; jittered-Cholesky fallback on ValueError (Covariance.py:167-187)."""
    if cov_matrix is not None:
        return inv(cov_matrix)
    K = gram(x, theta)
    try:
        return inv(K)
    except ValueError:
        m = len(K)
        L = cholesky(K + np.eye(m) * 1e-5, lower=True)
        Linv = solve_triangular(L, np.eye(m), lower=True)
        return np.dot(Linv.T, Linv)


def hessian(u, xi, theta):
    """H[a,b] = (w_a d_a w_b d_b - w_a [a==b]) v exp(-1/2 d^T W d), d = xi - u  (Covariance.py:660-674)."""
    v, _vt, w = unpack_theta(theta)
    d = np.asarray(xi, dtype=float) - np.asarray(u, dtype=float)
    e = v * np.exp(-0.5 * np.dot(d, w * d))
    wd = w * d
    return (np.outer(wd, wd) - np.diag(w)) * e


def jacobian(u, xi, theta):
    """J = -(xi-u) * w * c as a (d,1) column  (Covariance.py:676-689)."""
    v, _vt, w = unpack_theta(theta)
    d = np.asarray(xi, dtype=float) - np.asarray(u, dtype=float)
    e = v * np.exp(-0.5 * np.dot(d, w * d))
    return np.atleast_2d(-d * w * e).T


def d_gram_d_theta(x, theta, j):
    """dK/dtheta_j  (GaussianCovariance._d_cov_matrix_d_theta, Covariance.py:505-512, and
    _d_cov_matrix_d_theta_ij :605-657): j=0 -> noise-free Gram, j=1 -> vt I, j>=2 -> -1/2 Kf w_k (dx_k)^2."""
    _v, vt, w = unpack_theta(theta)
    x = np.asarray(x, dtype=float)
    n = len(x)
    if j == 1:
        return np.eye(n) * vt
    Kf = gram_ij(x, x, theta)
    if j == 0:
        return Kf
    xk = x[:, j - 2][None, :]
    dsq = -2.0 * np.dot(xk.T, xk) + np.tile(xk * xk, (n, 1)) + np.tile((xk * xk).T, (1, n))
    return -0.5 * Kf * dsq * w[j - 2]


def nll(x, t, theta):
    """N/2 log 2pi + 1/2 logdet K + 1/2 t^T Kinv t  (Covariance._negativeloglikelihood, Covariance.py:197-216);
    t is used as passed (the reference hands the centred targets in, GaussianProcess.py:39)."""
    t = np.asarray(t, dtype=float)
    K = gram(x, theta)
    logdet = np.linalg.slogdet(K)[1]
    Kinv = inv(K)
    return len(x) / 2.0 * np.log(2 * np.pi) + 0.5 * logdet + 0.5 * np.dot(t, np.dot(Kinv, t))


def nll_grad(x, t, theta):
    """Covariance._d_nll_d_theta (Covariance.py:266-282): 1/2 tr(Kinv dK_j) - 1/2 t^T Kinv dK_j Kinv t."""
    t = np.asarray(t, dtype=float)
    Kinv = inv(gram(x, theta))
    a = np.dot(Kinv, t)
    g = []
    for j in range(len(theta)):
        dK = d_gram_d_theta(x, theta, j)
        g.append(0.5 * np.dot(np.ravel(Kinv.T), np.ravel(dK)) - 0.5 * np.dot(a, np.dot(dK, a)))
    return np.array(g)


def theta_start(x, t):
    """GaussianCovariance.get_theta (Covariance.py:453-459)."""
    n, d = np.shape(x)
    th = np.ones(2 + d)
    th[0] = np.log(np.var(t))
    th[1] = np.log(np.var(t) / 4)
    th[2:] = -2 * np.log((np.max(x, 0) - np.min(x, 0)) / 2.0)
    return th


# --------------------------------------------------------------------------------------------
# L2: GP object  (reference: skgpuppy/GaussianProcess.py)
# --------------------------------------------------------------------------------------------

class OracleGP(object):
    """fit = centre targets + explicit Kinv  (GaussianProcess.py:19-41)."""

    def __init__(self, x, t, theta):
        self.x = x
        self.n, self.d = np.shape(x)
        self.meant = np.mean(t)
        self.t = np.asarray(t, dtype=float) - self.meant
        self.theta_min = np.asarray(theta, dtype=float)
        self.Kinv = inv_cov_matrix(self.x, self.theta_min)

    def beta(self):
        """beta = Kinv t, recomputed on every call  (GaussianProcess.py:114-119)."""
        return np.dot(self.Kinv, self.t)

    def estimate_many(self, x_stars):
        """mean = kv (Kinv t) + meant; var = diag(k - kv Kinv kv^T), k = Gram(x*)+vt I
        (GaussianProcess.py:68-80) -- including the M x M products the reference forms."""
        xs = np.array(x_stars)
        k = gram(xs, self.theta_min)
        kv = gram_ij(xs, self.x, self.theta_min)
        mean = np.dot(kv, np.dot(self.Kinv, self.t))
        var = k - np.dot(kv, np.dot(self.Kinv, kv.T))
        return mean + self.meant, np.diag(var)

    def estimate(self, x_star):
        """single-point twin  (GaussianProcess.py:94-111)."""
        xs = np.array(x_star)
        k = scalar_cov(xs, xs, self.theta_min)
        kv = gram_ij(np.atleast_2d(xs), self.x, self.theta_min)
        mean = np.dot(kv, np.dot(self.Kinv, self.t))
        var = k - np.dot(kv, np.dot(self.Kinv, kv.T))
        return mean[0] + self.meant, var[0, 0]

    def logdet(self):
        """Covariance.py:189-195."""
        return np.linalg.slogdet(gram(self.x, self.theta_min))[1]


# --------------------------------------------------------------------------------------------
# L2b: the GENERIC operator interface  (reference: the base class skgpuppy/Covariance.py:111-337 and GaussianProcess talking to
# it only through that interface, skgpuppy/GaussianProcess.py:19-111).  Restated so that operators that are NOT the built-in
# kernel -- a from-scratch subclass implementing __call__, a GaussianCovariance subclass with its own cov_matrix_ij -- have a
# checker; pinned by tests/golden/generic_ops.npz (tools/gen_golden.py --generic runs the same operators on the genuine reference).
# --------------------------------------------------------------------------------------------
class OracleCovariance(object):
    """Covariance (Covariance.py:111-337): everything from the subclass's scalar __call__."""

    def __call__(self, xi, xj, theta):
        raise NotImplementedError

    def cov_matrix_ij(self, xi, xj, theta):
        # Covariance.py:137-152
        K = np.zeros((len(xi), len(xj)))
        for i in range(len(xi)):
            for j in range(len(xj)):
                K[i, j] = self(xi[i], xj[j], theta)
        return K

    def cov_matrix(self, x, theta):
        # Covariance.py:155-164
        return self.cov_matrix_ij(x, x, theta)

    def inv_cov_matrix(self, x, theta, cov_matrix=None):
        # Covariance.py:167-187
        if cov_matrix is not None:
            return inv(cov_matrix)
        K = np.array(self.cov_matrix(x, theta))
        try:
            return inv(K)
        except ValueError:
            m = len(K)
            L = cholesky(K + np.eye(m) * 1e-5, lower=True)
            Linv = solve_triangular(L, np.eye(m), lower=True)
            return np.dot(Linv.T, Linv)

    def _log_det_cov_matrix(self, x, theta):
        # Covariance.py:189-195
        return np.linalg.slogdet(self.cov_matrix(x, theta))[1]

    def _negativeloglikelihood(self, x, t, theta):
        # Covariance.py:197-216
        invK = self.inv_cov_matrix(x, theta)
        return len(x) / 2.0 * np.log(2 * np.pi) + 0.5 * self._log_det_cov_matrix(x, theta) + 0.5 * np.dot(t.T, np.dot(invK, t))

    def _d_cov_d_theta(self, xi, xj, theta, j):
        # Covariance.py:219-233
        eps = 1e-5
        d = np.zeros(len(theta))
        d[j] = eps
        return (self(xi, xj, theta + d) - self(xi, xj, theta - d)) / (2 * eps)

    def _d_cov_matrix_d_theta(self, x, theta, j):
        # Covariance.py:236-265
        K = np.zeros((len(x), len(x)))
        for i1 in range(len(x)):
            for i2 in range(len(x)):
                K[i1, i2] = self._d_cov_d_theta(x[i1], x[i2], theta, j)
        return K

    def _d_nll_d_theta(self, x, t, theta):
        # Covariance.py:266-282
        Kinv = self.inv_cov_matrix(x, theta)
        g = []
        for j in range(len(theta)):
            dK = self._d_cov_matrix_d_theta(x, theta, j)
            g.append(0.5 * np.dot(np.ravel(Kinv.T), np.ravel(dK)) - 0.5 * np.dot(t.T, np.dot(Kinv, np.dot(dK, np.dot(Kinv, t)))))
        return np.array(g)


class OracleGaussianCovariance(OracleCovariance):
    """GaussianCovariance (Covariance.py:435-689) as an operator object: vectorised cov_matrix_ij, cov_matrix = cov_matrix_ij + vt I,
    closed-form derivative Grams that recompute the built-in Gram inline (:605-657: NOT through self.cov_matrix_ij)."""

    def __call__(self, xi, xj, theta):
        return scalar_cov(xi, xj, theta)

    def cov_matrix_ij(self, xi, xj, theta):
        return gram_ij(xi, xj, theta)

    def cov_matrix(self, x, theta):
        _v, vt, _w = unpack_theta(theta)
        return self.cov_matrix_ij(x, x, theta) + vt * np.eye(len(x))

    def _d_cov_matrix_d_theta(self, x, theta, j):
        return d_gram_d_theta(x, theta, j)

    def get_Jacobian(self, u, xi, theta):
        return jacobian(u, xi, theta)

    def get_Hessian(self, u, xi, theta):
        return hessian(u, xi, theta)


class OracleOperatorGP(object):
    """GaussianProcess on ANY operator (GaussianProcess.py:19-111): Kinv = cov.inv_cov_matrix, predictions from cov.cov_matrix /
    cov.cov_matrix_ij / cov.__call__ only.  Carries the attributes approx_parts / approx_dvh read (beta(), Kinv, theta_min, n, d)."""

    def __init__(self, x, t, cov, theta):
        self.x = x
        self.n, self.d = np.shape(x)
        self.meant = np.mean(t)
        self.t = np.asarray(t, dtype=float) - self.meant
        self.cov = cov
        self.theta_min = np.asarray(theta, dtype=float)
        self.Kinv = cov.inv_cov_matrix(self.x, self.theta_min)

    def beta(self):
        return np.dot(self.Kinv, self.t)

    def estimate_many(self, x_stars):
        xs = np.array(x_stars)
        k = self.cov.cov_matrix(xs, self.theta_min)
        kv = self.cov.cov_matrix_ij(xs, self.x, self.theta_min)
        mean = np.dot(kv, np.dot(self.Kinv, self.t))
        var = k - np.dot(kv, np.dot(self.Kinv, kv.T))
        return mean + self.meant, np.diag(var)

    def estimate(self, x_star):
        xs = np.array(x_star)
        k = self.cov(xs, xs, self.theta_min)
        kv = self.cov.cov_matrix_ij(np.atleast_2d(xs), self.x, self.theta_min)
        mean = np.dot(kv, np.dot(self.Kinv, self.t))
        var = k - np.dot(kv, np.dot(self.Kinv, kv.T))
        return mean[0] + self.meant, var[0, 0]

    def cjh(self, u):
        """UncertaintyPropagation.py:504-510 through the operator's own scalar kernel / Jacobian / Hessian"""
        x = np.asarray(self.x)
        return (np.array([self.cov(u, x[i], self.theta_min) for i in range(self.n)]),
                np.array([self.cov.get_Jacobian(u, x[i], self.theta_min) for i in range(self.n)]),
                np.array([self.cov.get_Hessian(u, x[i], self.theta_min) for i in range(self.n)]))


# --------------------------------------------------------------------------------------------
# L3: uncertainty propagation  (reference: skgpuppy/UncertaintyPropagation.py, ...2.pyx)
# --------------------------------------------------------------------------------------------

def cjh(gp, u):
    """C_ux[N], J_ux[N,d,1], H_ux[N,d,d] by N point-wise evaluations (UncertaintyPropagation.py:504-510),
    C through the scalar kernel so the +vt-on-equality quirk applies."""
    x = np.asarray(gp.x)
    C = np.array([scalar_cov(u, x[i], gp.theta_min) for i in range(gp.n)])
    J = np.array([jacobian(u, x[i], gp.theta_min) for i in range(gp.n)])
    H = np.array([hessian(u, x[i], gp.theta_min) for i in range(gp.n)])
    return C, J, H


def approx_parts(gp, u, Sigma, cache=None):
    """(mean_without_meant, sigma2, variance_rest) of UncertaintyPropagationApprox
    (UncertaintyPropagation.py:397-408, :412-433, :435-481)."""
    C, J, H = cache if cache is not None else cjh(gp, u)
    Sigma = np.asarray(Sigma, dtype=float)
    beta = gp.beta()
    Kinv = _c(gp.Kinv)
    n, d = gp.n, gp.d
    tr = np.array([np.dot(np.ravel(H[i].T), np.ravel(Sigma)) for i in range(n)])  # tracedot, Covariance.py:101-109
    mean = float(np.dot(beta, C)) + 0.5 * float(np.dot(beta, tr))
    lib = _loops()
    Cc, Jc, trc, bc = _c(C), _c(J.reshape(n, d)), _c(tr), _c(beta)
    S = _c(np.diag(Sigma))
    v, vt, _w = unpack_theta(gp.theta_min)
    sigma2 = (v + vt) - lib.orc_quad_form(_p(Kinv), _p(Cc), n)           # C(u,u) = v + vt (a3 quirk)
    var2 = lib.orc_var2(_p(Kinv), _p(bc), _p(Jc), _p(S), n, d)
    var3 = lib.orc_var3(_p(Kinv), _p(Cc), _p(trc), n)
    return mean, sigma2, var2 + var3


def approx_propagate(gp, u, Sigma, cache=None):
    """UncertaintyPropagationApprox.propagate_GA  (UncertaintyPropagation.py:490-523)."""
    mean, sigma2, rest = approx_parts(gp, u, Sigma, cache)
    return mean + gp.meant, sigma2 + rest


def approx_factor(gp, u, Sigma, v_out, cache=None):
    """_getFactor = (v_out - sigma2) / rest  (UncertaintyPropagation.py:526-560)."""
    _m, sigma2, rest = approx_parts(gp, u, Sigma, cache)
    return (v_out - sigma2) / rest


def approx_dvh(gp, u, h, cache=None):
    """_get_variance_dv_h  (UncertaintyPropagation.py:564-630 / .pyx:340-380)."""
    C, J, H = cache if cache is not None else cjh(gp, u)
    beta = gp.beta()
    Kinv = _c(gp.Kinv)
    n, d = gp.n, gp.d
    lib = _loops()
    hh = _c(H[:, h, h])
    v2 = lib.orc_dvh2(_p(Kinv), _p(_c(beta)), _p(_c(J.reshape(n, d))), n, d, h)
    v3 = lib.orc_var3(_p(Kinv), _p(_c(C)), _p(hh), n)
    return v2 + v3


def exact_mean(gp, u, Sigma, C=None):
    """UncertaintyPropagationExact.propagate_mean  (UncertaintyPropagation.py:247-290):
    Delta^-1 = W^-1 - diag(w_k/(1+w_k S_kk)), nc1 = det(I + Winv*Sigma [elementwise])^-1/2."""
    x = np.asarray(gp.x, dtype=float)
    u = np.asarray(u, dtype=float)
    Sigma = np.asarray(Sigma, dtype=float)
    _v, _vt, w = unpack_theta(gp.theta_min)
    if C is None:
        C = np.array([scalar_cov(u, x[i], gp.theta_min) for i in range(gp.n)])
    Winv = np.diag(w)
    Dinv = Winv - np.diag(w / (1.0 + w * np.diag(Sigma)))
    nc1 = 1.0 / np.sqrt(np.linalg.det(np.eye(gp.d) + Winv * Sigma))
    beta = gp.beta()
    s = 0.0
    for i in range(gp.n):
        a = u - x[i]
        s += beta[i] * C[i] * nc1 * np.exp(0.5 * np.dot(a, np.dot(Dinv, a)))
    return s


def exact_propagate(gp, u, Sigma):
    """UncertaintyPropagationExact.propagate_GA  (UncertaintyPropagation.py:292-303, :323-379; K1 with the
    explicit d^2 loop of UncertaintyPropagation2.pyx:173-179)."""
    x = _c(gp.x)
    u = _c(u)
    Sigma = np.asarray(Sigma, dtype=float)
    v, vt, w = unpack_theta(gp.theta_min)
    Winv = np.diag(w)
    W = np.diag(1.0 / w)
    Linv = _c(2.0 * Winv - inv(0.5 * W + Sigma))
    nc2 = 1.0 / np.sqrt(np.linalg.det(2.0 * Winv * Sigma + np.eye(gp.d)))
    C = _c(np.array([scalar_cov(u, x[i], gp.theta_min) for i in range(gp.n)]))
    mu = exact_mean(gp, u, Sigma, C)
    beta = _c(gp.beta())
    Kinv = _c(gp.Kinv)
    s = _loops().orc_exact_sum(_p(Kinv), _p(beta), _p(C), _p(x), _p(u), _p(Linv), float(nc2), gp.n, gp.d)
    return mu + gp.meant, (v + vt) - s - mu ** 2


def exact_propagate_operator(gp, covariance, u, Sigma, C=None, mean_only=False):
    """UncertaintyPropagationExact on ANY GP object (UncertaintyPropagation.py:269-290, :323-379): the class reads the GP only through
    _get_beta (gp.beta()), _get_W_inv (exp(theta_min[2:2+d]) -- whatever those entries mean to the operator), _inv_cov_matrix (gp.Kinv),
    x, _get_mean_t and _covariance = covariance(u, x_i) (the operator's scalar kernel).  Returns (mean + meant, variance), or the mean
    WITHOUT meant for mean_only (propagate_mean, which also accepts a caller's C_ux)."""
    x = _c(np.asarray(gp.x, dtype=float))
    u = _c(u)
    Sigma = np.asarray(Sigma, dtype=float)
    n, d = x.shape
    w = np.exp(np.asarray(gp.theta_min, dtype=float)[2:2 + d])
    Winv = np.diag(w)
    if C is None:
        C = np.array([float(np.ravel(covariance(u, gp.x[i]))[0]) for i in range(n)])
    C = _c(C)
    Dinv = Winv - np.diag(w / (1.0 + w * np.diag(Sigma)))
    nc1 = 1.0 / np.sqrt(np.linalg.det(np.eye(d) + Winv * Sigma))
    beta = _c(gp.beta() if callable(getattr(gp, "beta", None)) else np.dot(gp.Kinv, gp.t))
    mu = 0.0
    for i in range(n):
        a = u - x[i]
        mu += beta[i] * C[i] * nc1 * np.exp(0.5 * np.dot(a, np.dot(Dinv, a)))
    if mean_only:
        return mu
    Linv = _c(2.0 * Winv - inv(0.5 * np.diag(1.0 / w) + Sigma))
    nc2 = 1.0 / np.sqrt(np.linalg.det(2.0 * Winv * Sigma + np.eye(d)))
    s_ = _loops().orc_exact_sum(_p(_c(gp.Kinv)), _p(beta), _p(C), _p(x), _p(u), _p(Linv), float(nc2), n, d)
    cuu = float(np.ravel(covariance(u, u))[0])
    return mu + gp.meant, cuu - s_ - mu ** 2


# --------------------------------------------------------------------------------------------
# "next" row f3: Snelson sparse pseudo-input covariance  (reference: Covariance.py:692-1019)
# theta = (log v, log vt, log w_1..d, pseudo-inputs flattened row-major)
# --------------------------------------------------------------------------------------------

def spgp_split(theta, d, m):
    """(theta_gc, x_m)  (Covariance.py:713-714, :738-739)."""
    theta = np.asarray(theta, dtype=float)
    return theta[:2 + d], np.reshape(theta[2 + d:], (m, d))


def _spgp_lm(theta_gc, xm):
    return cholesky(gram_ij(xm, xm, theta_gc) + 1e-5 * np.eye(len(xm)), lower=True)


def spgp_cov_matrix_ij(xi, xj, theta, m):
    """Q_ij = K_iM (K_M + 1e-5 I)^-1 K_Mj, no diagonal correction  (Covariance.py:734-757)."""
    d = np.shape(xi)[1]
    tg, xm = spgp_split(theta, d, m)
    Zi = solve_triangular(_spgp_lm(tg, xm), gram_ij(xm, xi, tg), lower=True)
    Zj = solve_triangular(_spgp_lm(tg, xm), gram_ij(xm, xj, tg), lower=True)
    return np.dot(Zi.T, Zj)


def spgp_scalar_cov(xi, xj, theta, m):
    """SPGPCovariance.__call__  (Covariance.py:709-728): the full Gaussian kernel (incl. its +vt quirk) when xi == xj elementwise,
    otherwise the subset-of-regressors value k_iM (K_M + 1e-5 I)^-1 k_Mj."""
    xi, xj = np.asarray(xi, dtype=float), np.asarray(xj, dtype=float)
    tg, _xm = spgp_split(theta, len(xi), m)
    if (xi == xj).all():
        return scalar_cov(xi, xj, tg)
    return float(spgp_cov_matrix_ij(xi[None, :], xj[None, :], theta, m)[0, 0])


def spgp_lambda(x, theta, m):
    """diag(K_N - Q_N) + vt  (Covariance.py:829-831)."""
    d = np.shape(x)[1]
    tg, _xm = spgp_split(theta, d, m)
    v, vt, _w = unpack_theta(tg)
    return v - np.diag(spgp_cov_matrix_ij(x, x, theta, m)) + vt


def spgp_cov_matrix(x, theta, m):
    """Q_N + diag(K_N - Q_N) + vt I  (Covariance.py:814-833)."""
    Q = spgp_cov_matrix_ij(x, x, theta, m)
    return Q + np.diag(spgp_lambda(x, theta, m))


def spgp_inv_cov_matrix(x, theta, m):
    """Woodbury: LIinv - LIinv K_NM (B + 1e-5 I)^-1 K_MN LIinv, B = K_M + K_MN LIinv K_NM  (Covariance.py:835-863)."""
    d = np.shape(x)[1]
    tg, xm = spgp_split(theta, d, m)
    li = 1.0 / spgp_lambda(x, theta, m)
    Knm = gram_ij(x, xm, tg)
    B = gram_ij(xm, xm, tg) + np.dot(Knm.T, li[:, None] * Knm)
    LB = cholesky(B + 1e-5 * np.eye(m), lower=True)
    Y = solve_triangular(LB, Knm.T, lower=True) * li[None, :]
    return np.diag(li) - np.dot(Y.T, Y)


def spgp_nll(x, t, theta, m):
    """Snelson's O(N M^2) likelihood with jitter 1e-6  (Covariance.py:981-1019)."""
    x = np.asarray(x, dtype=float)
    N, d = x.shape
    tg, xm = spgp_split(theta, d, m)
    v, vt, _w = unpack_theta(tg)
    y = np.asarray(t, dtype=float)
    L = cholesky(gram_ij(xm, xm, tg) + 1e-6 * np.eye(m), lower=True)
    V = solve_triangular(L, gram_ij(xm, x, tg), lower=True)
    ep = 1.0 + (v - (V ** 2).sum(0)) / vt
    V = V / np.sqrt(ep)[None, :]
    y = y / np.sqrt(ep)
    Lm = cholesky(vt * np.eye(m) + np.dot(V, V.T), lower=True)
    bet = solve_triangular(Lm, np.dot(V, y), lower=True)
    return (np.log(np.diag(Lm)).sum() + (N - m) / 2.0 * np.log(vt) + (np.dot(y, y) - np.dot(bet, bet)) / 2.0 / vt
            + np.log(ep).sum() / 2.0 + 0.5 * N * np.log(2 * np.pi))


def spgp_nll_grad(x, t, theta, m):
    """d spgp_nll / d theta in O(N M^2), theta = (log v, log vt, log w_1..d, pseudo-inputs row-major).

    The reference's own gradient (Covariance.py:906-979) differentiates the dense N x N likelihood with O(N^2 M) work per
    parameter and does not run on Python 3 (float index at :910, :970), so there is nothing to pin against: "gradient
    parity unpinned".  This is the analytic derivative of the likelihood the reference minimises (Covariance.py:981-1019):
    with Q = K_M + 1e-6 I, K = K_MN, Sigma = K^T Q^-1 K + diag(gamma), gamma_n = vt + v - (K^T Q^-1 K)_nn, alpha = Sigma^-1 y,
    G = (Sigma^-1 - alpha alpha^T) / 2, g = diag G, H = G - diag(g):
        d nll = 2 <P H, dK> - <P H P^T, dQ> + sum(g) (dv + dvt),   P = Q^-1 K,
    everything expressed through the M x M matrix B = Q + K diag(1/gamma) K^T (Woodbury); the kernel derivatives then need
    E = Kbar o K and F = Qbar o (Q - 1e-6 I) only through their row / column sums and E X, F Xb.  Checked against central
    differences of spgp_nll in tests/test_oracle_golden.py; the GPU path (gpx_spgp_nll_grad) is checked against this."""
    x = np.asarray(x, dtype=float)
    N, d = x.shape
    tg, xm = spgp_split(theta, d, m)
    v, vt, w = unpack_theta(tg)
    y = np.asarray(t, dtype=float)
    Qk = gram_ij(xm, xm, tg)
    K = gram_ij(xm, x, tg)                                      # [M, N]
    L = cholesky(Qk + 1e-6 * np.eye(m), lower=True)
    V = solve_triangular(L, K, lower=True)
    gamma = vt + v - (V ** 2).sum(0)
    A = vt * np.eye(m) + np.dot(V / gamma[None, :] * vt, V.T)   # = vt I + V D^-1 V^T, D = gamma / vt
    Ainv = np.linalg.inv(A)
    VD = V * (vt / gamma)[None, :]                              # V D^-1
    T1 = np.dot(Ainv, VD)                                       # A^-1 V D^-1   [M, N]
    betaA = np.dot(T1, y)
    alpha = (y - np.dot(V.T, betaA)) / gamma
    s_n = (T1 * V).sum(0)
    g = 0.5 * ((1.0 - s_n) / gamma - alpha ** 2)
    Vbar = T1 - np.outer(betaA, alpha) - 2.0 * V * g[None, :]
    Kbar = solve_triangular(L, Vbar, lower=True, trans="T")     # L^-T Vbar
    Qb = -0.5 * (np.eye(m) - vt * Ainv - np.outer(betaA, betaA)) + np.dot(V * g[None, :], V.T)
    Qbar = solve_triangular(L, solve_triangular(L, Qb, lower=True, trans="T").T, lower=True, trans="T")   # L^-T Qb L^-1
    Qbar = 0.5 * (Qbar + Qbar.T)
    E = Kbar * K
    F = Qbar * Qk
    E1, Ec, EX = E.sum(1), E.sum(0), np.dot(E, x)
    F1, FX = F.sum(1), np.dot(F, xm)
    grad = np.empty(2 + d + m * d)
    grad[0] = E.sum() + F.sum() + v * g.sum()
    grad[1] = vt * g.sum()
    for k in range(d):
        qe = np.dot(xm[:, k] ** 2, E1) - 2.0 * np.dot(xm[:, k], EX[:, k]) + np.dot(x[:, k] ** 2, Ec)
        qf = 2.0 * np.dot(xm[:, k] ** 2, F1) - 2.0 * np.dot(xm[:, k], FX[:, k])
        grad[2 + k] = -0.5 * w[k] * (qe + qf)
    dxb = -(xm * E1[:, None] - EX) * w[None, :] - 2.0 * (xm * F1[:, None] - FX) * w[None, :]
    grad[2 + d:] = dxb.ravel()
    return grad


def spgp_nll_chunked(x, t, theta, m, chunk=32768):
    """spgp_nll (Covariance.py:981-1019) with the N-long contractions accumulated over column chunks of K_MN, so that
    BASELINE config 5 (N = 262144, M = 2048) needs ~1.5 GB of host memory instead of ~30 GB.  Same formula, same jitter;
    pinned against spgp_nll in tests/test_oracle_golden.py."""
    x = np.asarray(x, dtype=float)
    N, d = x.shape
    tg, xm = spgp_split(theta, d, m)
    v, vt, _w = unpack_theta(tg)
    y = np.asarray(t, dtype=float)
    L = cholesky(gram_ij(xm, xm, tg) + 1e-6 * np.eye(m), lower=True)
    VVt = np.zeros((m, m))
    Vy = np.zeros(m)
    yy = 0.0
    logep = 0.0
    for c0 in range(0, N, chunk):
        V = solve_triangular(L, gram_ij(xm, x[c0:c0 + chunk], tg), lower=True)
        ep = 1.0 + (v - (V ** 2).sum(0)) / vt
        V = V / np.sqrt(ep)[None, :]
        yc = y[c0:c0 + chunk] / np.sqrt(ep)
        VVt += np.dot(V, V.T)
        Vy += np.dot(V, yc)
        yy += np.dot(yc, yc)
        logep += np.log(ep).sum()
    Lm = cholesky(vt * np.eye(m) + VVt, lower=True)
    bet = solve_triangular(Lm, Vy, lower=True)
    return (np.log(np.diag(Lm)).sum() + (N - m) / 2.0 * np.log(vt) + (yy - np.dot(bet, bet)) / 2.0 / vt
            + logep / 2.0 + 0.5 * N * np.log(2 * np.pi))


def spgp_generic_nll(x, t, theta, m):
    """Base-class likelihood on the dense SPGP covariance  (Covariance.py:197-216 with :814-833)."""
    K = spgp_cov_matrix(x, theta, m)
    t = np.asarray(t, dtype=float)
    return 0.5 * len(t) * np.log(2 * np.pi) + 0.5 * np.linalg.slogdet(K)[1] + 0.5 * np.dot(t, np.linalg.solve(K, t))


class OracleSPGP(object):
    """GaussianProcess(x, t, SPGPCovariance(m), theta): the generic dense formulas on Q and the Woodbury inverse
    (GaussianProcess.py:19-41, :68-80)."""

    def __init__(self, x, t, theta, m):
        self.x = np.asarray(x, dtype=float)
        self.m = m
        self.meant = np.mean(t)
        self.t = np.asarray(t, dtype=float) - self.meant
        self.theta_min = np.asarray(theta, dtype=float)
        self.Kinv = spgp_inv_cov_matrix(self.x, self.theta_min, m)

    def estimate_many(self, x_stars):
        xs = np.asarray(x_stars, dtype=float)
        k = spgp_cov_matrix(xs, self.theta_min, self.m)
        kv = spgp_cov_matrix_ij(xs, self.x, self.theta_min, self.m)
        mean = np.dot(kv, np.dot(self.Kinv, self.t))
        var = k - np.dot(kv, np.dot(self.Kinv, kv.T))
        return mean + self.meant, np.diag(var)
