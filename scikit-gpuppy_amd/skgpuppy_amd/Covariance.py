"""Covariance operator interface -- host-side mirror of skgpuppy/Covariance.py for the hot path.

`Covariance` is the reference's operator interface (skgpuppy/Covariance.py:111-359); only
`GaussianCovariance` (:435-689) is on the accelerated path.  Matrix-sized work (cov_matrix_ij,
cov_matrix, inv_cov_matrix) is done by libgpx on the GPU; the scalar accessors (__call__, get_theta,
get_Jacobian, get_Hessian: O(d^2) per call) are plain host arithmetic exactly as the reference defines
them -- their batched forms are gpx_cjh / the fused device kernels.
"""
import ctypes

import numpy as np

from . import _gpx


def tracedot(A, B):
    """trace(dot(A, B))  (skgpuppy/Covariance.py:101-109)."""
    return np.dot(np.ravel(np.asarray(A).T), np.ravel(B))


class _MatrixModel(object):
    """Owner of one gpx_handle built from a SUPPLIED covariance matrix (gpx_fit_matrix): the device route of the generic operator
    interface -- Cholesky factor of the operator's own cov_matrix(x, theta) (+1e-5 I retry), alpha, lazily K^-1."""

    def __init__(self, K, t_centered):
        K = _gpx.f64(K)
        if K.ndim != 2 or K.shape[0] != K.shape[1]:
            raise ValueError("cov_matrix must return a square matrix, got shape %r" % (K.shape,))
        self.n, self.d = K.shape[0], 0
        t = np.zeros(self.n) if t_centered is None else _gpx.f64(t_centered)
        if t.shape != (self.n,):
            raise ValueError("t must have %d entries" % self.n)
        self._h = ctypes.c_void_p()
        _gpx.check(_gpx.lib.gpx_fit_matrix(_gpx.ptr(K), _gpx.ptr(t), self.n, None, ctypes.byref(self._h)), "gpx_fit_matrix")

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("device model already released")
        return self._h

    def close(self, _free=_gpx.lib.gpx_free, _null=ctypes.c_void_p):
        if getattr(self, "_h", None):
            _free(self._h)
            self._h = _null()

    __del__ = close

    def predict_kv(self, kv, kdiag):
        kv = _gpx.f64(np.atleast_2d(kv))
        kdiag = _gpx.f64(np.atleast_1d(kdiag))
        m = kv.shape[0]
        if kv.shape[1] != self.n or kdiag.shape != (m,):
            raise ValueError("kv must be (m, %d) and kdiag (m,)" % self.n)
        mean = np.empty(m)
        var = np.empty(m)
        _gpx.check(_gpx.lib.gpx_predict_kv(self.handle, _gpx.ptr(kv), m, _gpx.ptr(kdiag), _gpx.ptr(mean), _gpx.ptr(var)), "gpx_predict_kv")
        return mean, var

    def alpha(self):
        out = np.empty(self.n)
        _gpx.check(_gpx.lib.gpx_alpha(self.handle, _gpx.ptr(out)), "gpx_alpha")
        return out

    def solve(self, B):
        B = _gpx.f64(np.atleast_2d(B))
        kb = np.empty_like(B)
        _gpx.check(_gpx.lib.gpx_solve(self.handle, _gpx.ptr(B), B.shape[0], None, _gpx.ptr(kb)), "gpx_solve")
        return kb

    def kinv(self):
        out = np.empty((self.n, self.n))
        _gpx.check(_gpx.lib.gpx_kinv(self.handle, _gpx.ptr(out)), "gpx_kinv")
        return out

    def logdet(self):
        v = ctypes.c_double()
        _gpx.check(_gpx.lib.gpx_logdet(self.handle, ctypes.byref(v)), "gpx_logdet")
        return v.value

    def nll(self):
        v = ctypes.c_double()
        _gpx.check(_gpx.lib.gpx_nll(self.handle, ctypes.byref(v)), "gpx_nll")
        return v.value

    def nll_grad_entry(self, dK):
        dK = _gpx.f64(dK)
        if dK.shape != (self.n, self.n):
            raise ValueError("_d_cov_matrix_d_theta must return an (%d, %d) matrix" % (self.n, self.n))
        v = ctypes.c_double()
        _gpx.check(_gpx.lib.gpx_nll_grad_matrix(self.handle, _gpx.ptr(dK), ctypes.byref(v)), "gpx_nll_grad_matrix")
        return v.value

    def jitter(self):
        v = ctypes.c_double()
        _gpx.check(_gpx.lib.gpx_jitter_used(self.handle, ctypes.byref(v)), "gpx_jitter_used")
        return v.value


class Covariance(object):
    """Superclass for all covariance functions (skgpuppy/Covariance.py:111-359): the interface GaussianProcess talks to.

    As in the reference the base class is GENERIC: a subclass that implements `__call__` (and `get_theta` for the ML fit)
    gets cov_matrix_ij / cov_matrix from it, and inv_cov_matrix, the likelihood and its gradient from those -- the matrix-sized
    algebra (factorisation with the +1e-5 I retry, inverse, log det, quadratic forms, trace terms) runs on the GPU on the
    operator's own matrices (gpx_fit_matrix and friends); only the entry-wise evaluation of a user's Python kernel is host work,
    as it is in the reference."""

    def __init__(self):
        pass

    def __call__(self, xi, xj, theta):
        raise NotImplementedError

    def get_theta(self, x, t):
        raise NotImplementedError

    def cov_matrix_ij(self, xi, xj, theta):
        """N1 x N2 matrix of the operator's own scalar kernel (Covariance.py:137-152)"""
        ni, nj = len(xi), len(xj)
        K = np.zeros((ni, nj))
        for i in range(ni):
            for j in range(nj):
                K[i, j] = self(xi[i], xj[j], theta)
        return K

    def cov_matrix(self, x, theta):
        # (Covariance.py:155-164)
        return self.cov_matrix_ij(x, x, theta)

    def inv_cov_matrix(self, x, theta, cov_matrix=None):
        """inverse of the operator's own cov_matrix(x, theta), or of `cov_matrix` when given (Covariance.py:167-187).  The reference
        LU-inverts and falls back to chol(K + 1e-5 I) when that raises; here K is Cholesky-factored on the GPU with the same
        single +1e-5 I retry and K^-1 = L^-T L^-1."""
        if cov_matrix is not None:
            K = _gpx.f64(cov_matrix)
            if K.ndim != 2 or K.shape[0] != K.shape[1]:
                raise ValueError("cov_matrix must be square")
            out = np.empty_like(K)
            _gpx.check(_gpx.lib.gpx_spd_inverse(_gpx.ptr(K), K.shape[0], _gpx.ptr(out), None), "gpx_spd_inverse")
            return out
        model = _MatrixModel(np.array(self.cov_matrix(x, theta)), None)
        try:
            return model.kinv()
        finally:
            model.close()

    def get_Hessian(self, u, xi, theta):
        raise NotImplementedError

    def get_Jacobian(self, u, xi, theta):
        raise NotImplementedError

    # ---- hyper-parameter maximum likelihood ("next" row f1; skgpuppy/Covariance.py:189-337) ----------------
    def _matrix_model_at(self, x, t, theta):
        """device model of the operator's own cov_matrix(x, theta) with targets t, kept for the next call at the same theta
        (L-BFGS-B asks for the value and the gradient one after the other)"""
        th = np.array(theta, dtype=float)
        ta = None if t is None else _gpx.f64(t)
        key = (id(x), np.shape(x), None if ta is None else ta.tobytes(), th.tobytes())
        cached = getattr(self, "_mm_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        if cached is not None:
            cached[1].close()
        self._mm_cache = None
        model = _MatrixModel(np.array(self.cov_matrix(x, theta)), ta)
        self._mm_cache = (key, model)
        return model

    def _log_det_cov_matrix(self, x, theta):
        """log det of cov_matrix(x, theta) (Covariance.py:189-195: numpy slogdet); here 2 sum log diag of its Cholesky factor,
        computed on the GPU from the operator's own matrix"""
        return self._matrix_model_at(x, None, theta).logdet()

    def _d_cov_d_theta(self, xi, xj, theta, j):
        """central difference of the scalar kernel in theta_j, eps = 1e-5 (Covariance.py:219-233)"""
        eps = 1e-5
        d = np.zeros(len(theta))
        d[j] = eps
        return (self(xi, xj, theta + d) - self(xi, xj, theta - d)) / (2 * eps)

    def _d_cov_matrix_d_theta_ij(self, xi, xj, theta, j):
        """entry-wise _d_cov_d_theta (Covariance.py:236-253); GaussianCovariance overrides it with the closed form"""
        ni, nj = len(xi), len(xj)
        K = np.zeros((ni, nj))
        for i1 in range(ni):
            for i2 in range(nj):
                K[i1, i2] = self._d_cov_d_theta(xi[i1], xj[i2], theta, j)
        return K

    def _d_cov_matrix_d_theta(self, x, theta, j):
        # (Covariance.py:256-265)
        return self._d_cov_matrix_d_theta_ij(x, x, theta, j)

    def _negativeloglikelihood(self, x, t, theta):
        """N/2 log 2pi + 1/2 log det K + 1/2 t^T K^-1 t for the operator's own K (Covariance.py:197-216): factorisation, log det and
        the quadratic form on the GPU; 1e20 when K cannot be factored, like the reference's except branch."""
        try:
            return self._matrix_model_at(x, t, theta).nll()
        except (np.linalg.LinAlgError, RuntimeWarning, ZeroDivisionError, ValueError):
            return 1.0e+20

    def _d_nll_d_theta(self, x, t, theta):
        """gradient of the NLL from the operator's own derivative matrices (Covariance.py:266-282): entry j is
        1/2 tr(K^-1 dK_j) - 1/2 alpha^T dK_j alpha, one fused device pass over K^-1 and dK_j each"""
        model = self._matrix_model_at(x, t, theta)
        return np.array([model.nll_grad_entry(self._d_cov_matrix_d_theta(x, theta, j)) for j in range(len(theta))])

    def _nll_function(self, x, t):
        # (Covariance.py:284-297)
        def nll(theta):
            return self._negativeloglikelihood(x, t, theta)
        return nll

    def _gradient_function(self, x, t):
        # (Covariance.py:299-312): retried at 0.999 theta when the factorisation fails
        def gradient(theta):
            try:
                gr = self._d_nll_d_theta(x, t, theta)
            except np.linalg.LinAlgError:
                gr = self._d_nll_d_theta(x, t, theta * 0.999)
            return gr
        return gradient

    def ml_estimate(self, x, t):
        """maximum-likelihood theta by L-BFGS-B with the analytic gradient (Covariance.py:314-337, which goes
        through Utilities.minimize(method=["l_bfgs_b"]) = scipy's fmin_l_bfgs_b with default settings)."""
        from scipy.optimize import fmin_l_bfgs_b
        theta_start = self.get_theta(x, t)
        func = self._nll_function(x, t)
        fprime = self._gradient_function(x, t)
        theta_min = fmin_l_bfgs_b(func, theta_start, bounds=None, approx_grad=False, fprime=fprime)
        return np.array(theta_min[0])

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_mm_cache", None)     # device handles never enter a pickle
        return state


def _theta(theta, d):
    th = _gpx.f64(theta)
    if th.ndim != 1 or th.shape[0] != d + 2:
        raise ValueError("theta must have 2 + d = %d entries, got shape %r" % (d + 2, th.shape))
    return th


class GaussianCovariance(Covariance):
    """ARD squared-exponential kernel, theta = (log v, log vt, log w_1..w_d)
    (skgpuppy/Covariance.py:435-689).

    The matrix methods are the fused HIP path (Gram kernel, Cholesky on the Gram tiles, one-pass likelihood gradient).  A user
    subclass that overrides one of the matrix builders keeps the reference's semantics -- cov_matrix is cov_matrix_ij + vt I,
    inv_cov_matrix / the likelihood / its gradient are built on whatever those return -- by falling back to the generic
    device route of `Covariance` (`_fused()` decides per instance)."""

    _MATRIX_METHODS = ("cov_matrix_ij", "cov_matrix", "inv_cov_matrix", "_d_cov_matrix_d_theta", "_d_cov_matrix_d_theta_ij",
                       "_log_det_cov_matrix")

    def _fused(self):
        """True when every matrix-defining method of this instance is GaussianCovariance's own"""
        cls = type(self)
        return cls is GaussianCovariance or all(getattr(cls, m) is getattr(GaussianCovariance, m) for m in self._MATRIX_METHODS)

    def __call__(self, xi, xj, theta):
        # scalar kernel incl. the "+vt iff xi == xj elementwise" hack (Covariance.py:440-451)
        xi = np.asarray(xi)
        xj = np.asarray(xj)
        with np.errstate(divide="ignore"):
            v = np.exp(theta[0])
            vt = np.exp(theta[1])
            w = np.exp(np.asarray(theta[2:], dtype=float))
        diff = xi - xj
        return v * np.exp(-0.5 * np.dot(diff, w * diff)) + (vt if (xi == xj).all() else 0)

    def get_theta(self, x, t):
        # initial guess for the hyper-parameter search (Covariance.py:453-459)
        n, d = np.shape(x)
        theta = np.ones(2 + d)
        theta[0] = np.log(np.var(t)) if t is not None else 1
        theta[1] = np.log(np.var(t) / 4) if t is not None else 1
        theta[2:] = -2 * np.log((np.max(x, 0) - np.min(x, 0)) / 2.0)
        return theta

    def _gram(self, xi, xj, theta, add_diag):
        a = _gpx.f64(xi)
        b = a if xj is xi else _gpx.f64(xj)
        if a.ndim != 2 or b.ndim != 2 or a.shape[1] != b.shape[1]:
            raise ValueError("inputs must be (n, d) arrays with equal d")
        th = _theta(theta, a.shape[1])
        K = np.empty((a.shape[0], b.shape[0]), dtype=np.float64)
        if K.size:
            st = _gpx.lib.gpx_gram(_gpx.ptr(a), a.shape[0], _gpx.ptr(b), b.shape[0], a.shape[1], _gpx.ptr(th),
                                   float(add_diag), _gpx.ptr(K))
            _gpx.check(st, "gpx_gram")
        return K

    def cov_matrix_ij(self, xi, xj, theta):
        """N1 x N2 cross-covariance without the noise term (Covariance.py:466-483) -- HIP Gram kernel."""
        return self._gram(xi, xj, theta, 0.0)

    def cov_matrix(self, x, theta):
        """cov_matrix_ij(x, x) + vt I (Covariance.py:461-464); the diagonal add is fused in the kernel."""
        with np.errstate(divide="ignore"):
            vt = float(np.exp(theta[1]))
        if type(self).cov_matrix_ij is not GaussianCovariance.cov_matrix_ij:     # a subclass's own cross-covariance
            return np.asarray(self.cov_matrix_ij(x, x, theta), dtype=float) + vt * np.eye(np.shape(x)[0])
        return self._gram(x, x, theta, vt)

    def inv_cov_matrix(self, x, theta, cov_matrix=None):
        """K^-1 (Covariance.py:167-187).  The reference LU-inverts; here K is Cholesky-factored on the GPU
        (with the reference's +1e-5 I retry on a non-PD pivot) and K^-1 = L^-T L^-1."""
        if cov_matrix is not None or not self._fused():
            return Covariance.inv_cov_matrix(self, x, theta, cov_matrix)
        from .GaussianProcess import _DeviceModel
        xa = _gpx.f64(x)
        model = _DeviceModel(xa, np.zeros(xa.shape[0]), _theta(theta, xa.shape[1]))
        try:
            return model.kinv()
        finally:
            model.close()

    def _log_det_cov_matrix(self, x, theta):
        """log det (K + vt I) (Covariance.py:189-195) from the Cholesky factor of a device fit"""
        if not self._fused():
            return Covariance._log_det_cov_matrix(self, x, theta)
        from .GaussianProcess import _DeviceModel
        xa = _gpx.f64(x)
        model = _DeviceModel(xa, np.zeros(xa.shape[0]), _theta(theta, xa.shape[1]))
        try:
            return model.logdet()
        finally:
            model.close()

    def _d_cov_d_theta(self, xi, xj, theta, j):
        """d k(xi, xj) / d theta_j in closed form (Covariance.py:485-503); theta is the LOG of (v, vt, w): j = 0 -> the
        noise-free kernel, j = 1 -> vt iff xi == xj elementwise, j >= 2 -> -1/2 (dx_k)^2 w_k k"""
        xi = np.asarray(xi, dtype=float)
        xj = np.asarray(xj, dtype=float)
        diff = xi - xj
        with np.errstate(divide="ignore"):
            v = np.exp(theta[0])
            vt = np.exp(theta[1])
            w = np.exp(np.asarray(theta[2:], dtype=float))
        kf = v * np.exp(-0.5 * np.dot(diff, w * diff))
        if j == 0:
            return kf
        if j == 1:
            return vt if (xi == xj).all() else 0
        return -0.5 * diff[j - 2] ** 2 * kf * w[j - 2]

    def _d_cov_matrix_d_theta(self, x, theta, j):
        # (Covariance.py:505-512)
        if j == 1:
            with np.errstate(divide="ignore"):
                return np.eye(np.shape(x)[0]) * np.exp(theta[1])
        return self._d_cov_matrix_d_theta_ij(x, x, theta, j)

    def _d_cov_matrix_d_theta_ij(self, xi, xj, theta, j, Cov=None):
        """derivative Gram d K_ij / d theta_j (Covariance.py:605-657): the noise-free Gram from the HIP kernel (or `Cov` when
        the caller has it), scaled entry-wise by -1/2 w_k (xi_k - xj_k)^2 for j = k + 2.  An accessor: the likelihood gradient
        itself never forms these matrices (nll_grad_kernel)."""
        a = _gpx.f64(xi)
        b = _gpx.f64(xj)
        if j == 1:
            return np.zeros((a.shape[0], b.shape[0]))
        # (the reference recomputes the built-in Gram inline here, :622-633 -- also for a subclass with its own cov_matrix_ij)
        K = np.asarray(Cov, dtype=float) if Cov is not None else GaussianCovariance.cov_matrix_ij(self, a, b, theta)
        if j == 0:
            return K
        w = np.exp(np.asarray(theta[2:], dtype=float))
        dk = a[:, j - 2][:, None] - b[:, j - 2][None, :]
        return -0.5 * K * (dk * dk) * w[j - 2]

    def _model_at(self, x, t, theta):
        """device model fitted at theta, cached on (x, t, theta) identity/bytes: L-BFGS-B asks for the value and the
        gradient at the same theta in two separate calls."""
        from .GaussianProcess import _DeviceModel
        xa = _gpx.f64(x)
        ta = _gpx.f64(t)
        th = _theta(theta, xa.shape[1]).copy()
        key = (xa.ctypes.data if xa is x else id(x), xa.shape, ta.tobytes(), th.tobytes())
        cached = getattr(self, "_ml_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        if cached is not None:
            cached[1].close()
        model = _DeviceModel(xa, ta, th)
        self._ml_cache = (key, model)
        return model

    def _negativeloglikelihood(self, x, t, theta):
        """N/2 log 2pi + 1/2 log det K + 1/2 t^T K^-1 t on the GPU (Covariance.py:197-216); 1e20 when K cannot be
        factored, like the reference's except branch."""
        if not self._fused():
            return Covariance._negativeloglikelihood(self, x, t, theta)
        try:
            model = self._model_at(x, t, theta)
            out = ctypes.c_double()
            _gpx.check(_gpx.lib.gpx_nll(model.handle, ctypes.byref(out)), "gpx_nll")
            return out.value
        except (np.linalg.LinAlgError, ValueError, ZeroDivisionError):
            return 1.0e+20

    def _d_nll_d_theta(self, x, t, theta):
        """gradient of the NLL (Covariance.py:266-282 with the derivative Grams of :505-512, :605-657): one fused
        pass over K^-1 on the GPU instead of 2+d derivative matrices."""
        if not self._fused():
            return Covariance._d_nll_d_theta(self, x, t, theta)
        model = self._model_at(x, t, theta)
        g = np.empty(len(theta))
        _gpx.check(_gpx.lib.gpx_nll_grad(model.handle, _gpx.ptr(g)), "gpx_nll_grad")
        return g

    def __getstate__(self):
        state = Covariance.__getstate__(self)
        state.pop("_ml_cache", None)     # device handles never enter a pickle
        return state

    def get_Hessian(self, u, xi, theta):
        # (Covariance.py:660-674)
        with np.errstate(divide="ignore"):
            v = np.exp(theta[0])
            w = np.exp(np.asarray(theta[2:], dtype=float))
        diff = np.asarray(xi, dtype=float) - np.asarray(u, dtype=float)
        e = v * np.exp(-0.5 * np.dot(diff, w * diff))
        wd = diff * w
        return (np.outer(wd, wd) - np.diag(w)) * e

    def get_Jacobian(self, u, xi, theta):
        # (Covariance.py:676-689): -(xi-u) w c as a (d,1) column
        with np.errstate(divide="ignore"):
            v = np.exp(theta[0])
            w = np.exp(np.asarray(theta[2:], dtype=float))
        diff = np.asarray(xi, dtype=float) - np.asarray(u, dtype=float)
        e = v * np.exp(-0.5 * np.dot(diff, w * diff))
        return np.atleast_2d(-diff * w * e).T


class _SPGPDeviceModel(object):
    """Owner of one gpx_spgp handle: K_NM, chol(K_M + 1e-5 I), Lambda and chol(B + 1e-5 I) resident in HBM."""

    def __init__(self, x, t_centered, theta_gc, xb):
        self.n, self.d = x.shape
        self.m = xb.shape[0]
        self._h = ctypes.c_void_p()
        st = _gpx.lib.gpx_spgp_fit(_gpx.ptr(x), _gpx.ptr(t_centered), self.n, self.d, _gpx.ptr(theta_gc), _gpx.ptr(xb),
                                   self.m, ctypes.byref(self._h))
        _gpx.check(st, "gpx_spgp_fit")

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("device model already released")
        return self._h

    def close(self, _free=_gpx.lib.gpx_spgp_free, _null=ctypes.c_void_p):
        if getattr(self, "_h", None):
            _free(self._h)
            self._h = _null()

    __del__ = close

    def predict(self, xs):
        k = xs.shape[0]
        mean = np.empty(k)
        var = np.empty(k)
        _gpx.check(_gpx.lib.gpx_spgp_predict(self.handle, _gpx.ptr(xs), k, _gpx.ptr(mean), _gpx.ptr(var)), "gpx_spgp_predict")
        return mean, var

    def nll(self):
        out = ctypes.c_double()
        _gpx.check(_gpx.lib.gpx_spgp_nll(self.handle, ctypes.byref(out)), "gpx_spgp_nll")
        return out.value

    def nll_grad(self):
        out = np.empty(2 + self.d + self.m * self.d)
        _gpx.check(_gpx.lib.gpx_spgp_nll_grad(self.handle, _gpx.ptr(out)), "gpx_spgp_nll_grad")
        return out

    def dense(self, which):
        out = np.empty((self.n, self.n))
        _gpx.check(_gpx.lib.gpx_spgp_dense(self.handle, which, _gpx.ptr(out)), "gpx_spgp_dense")
        return out

    def kinv(self):
        return self.dense(1)

    def cross(self, xi, xj):
        out = np.empty((xi.shape[0], xj.shape[0]))
        if out.size:
            _gpx.check(_gpx.lib.gpx_spgp_cross(self.handle, _gpx.ptr(xi), xi.shape[0], _gpx.ptr(xj), xj.shape[0], _gpx.ptr(out)),
                       "gpx_spgp_cross")
        return out


class SPGPCovariance(Covariance):
    """Snelson's sparse pseudo-input covariance ("next" row f3; skgpuppy/Covariance.py:692-1019).

    theta = (log v, log vt, log w_1..w_d, the m pseudo-inputs flattened row-major).  Like the reference's it offers no
    Jacobian/Hessian (no uncertainty propagation).  Fit, prediction and Snelson's likelihood are O(N m^2) on the GPU;
    cov_matrix / inv_cov_matrix materialise the reference's N x N matrices for small N only."""

    def __init__(self, m):
        self.m = m
        self.cov = GaussianCovariance()

    def _split(self, theta, d):
        th = _gpx.f64(theta)
        if th.ndim != 1 or th.shape[0] != 2 + d + self.m * d:
            raise ValueError("theta must have 2 + d + m*d = %d entries, got shape %r" % (2 + d + self.m * d, th.shape))
        return np.ascontiguousarray(th[:2 + d]), np.ascontiguousarray(np.reshape(th[2 + d:], (self.m, d)))

    def _model(self, x, t, theta):
        xa = _gpx.f64(x)
        if xa.ndim != 2:
            raise ValueError("x must be an (n, d) array")
        tg, xb = self._split(theta, xa.shape[1])
        ta = np.zeros(xa.shape[0]) if t is None else _gpx.f64(t)
        return _SPGPDeviceModel(xa, ta, tg, xb)

    def __call__(self, xi, xj, theta):
        # (Covariance.py:708-732): the full kernel on the diagonal, the low-rank one elsewhere
        xi = np.asarray(xi)
        xj = np.asarray(xj)
        d = np.shape(xi)[0]
        if (xi == xj).all():
            return self.cov(xi, xj, theta[0:2 + d])
        return self.cov_matrix_ij(np.atleast_2d(xi), np.atleast_2d(xj), theta)[0, 0]

    def get_theta(self, x, t):
        # (Covariance.py:734-741): GaussianCovariance start + m training rows drawn at random as pseudo-inputs
        n, d = np.shape(x)
        theta = np.ones(2 + d + self.m * d)
        theta[0:2 + d] = self.cov.get_theta(x, t)
        theta[2 + d:] = np.reshape(np.asarray(x)[np.random.randint(n, size=self.m), :], self.m * d)
        return theta

    def cov_matrix_ij(self, xi, xj, theta):
        """Q_ij = K_iM (K_M + 1e-5 I)^-1 K_Mj (Covariance.py:743-763).  Depends on theta only through the pseudo-inputs'
        factor: the device model behind it is fitted on the M pseudo-inputs themselves (O(M^3), not a fit of xi) and kept
        for the next call with the same theta."""
        a = _gpx.f64(xi)
        b = _gpx.f64(xj)
        th = _gpx.f64(theta)
        key = th.tobytes()
        cached = getattr(self, "_cross_model", None)
        if cached is None or cached[0] != key:
            if cached is not None:
                cached[1].close()
            _tg, xb = self._split(th, a.shape[1])
            cached = (key, self._model(xb, None, th))
            self._cross_model = cached
        return cached[1].cross(a, b)

    def cov_matrix(self, x, theta):
        """Q_N + diag(K_N - Q_N) + vt I (Covariance.py:814-833); shares the device model of (x, theta) with inv_cov_matrix."""
        return self._fit_model(x, None, theta).dense(0)

    def inv_cov_matrix(self, x, theta, cov_matrix=None):
        """the Woodbury inverse (Covariance.py:835-863); `cov_matrix` is ignored as in the reference."""
        return self._fit_model(x, None, theta).dense(1)

    @staticmethod
    def _signature(a):
        """content signature of an array: shape, dtype and a hash of ALL its bytes (xxh3 at several GB/s, crc32 otherwise: a few
        ms for C5's 16 MB of inputs next to a 40 ms fit).  A sampled hash (round 3) missed in-place edits of a few rows and
        handed back a stale device model."""
        if a is None:
            return None
        a = np.ascontiguousarray(a)
        buf = memoryview(a).cast("B")
        try:
            import xxhash
            digest = xxhash.xxh3_64_intdigest(buf)
        except ImportError:
            import zlib
            digest = zlib.crc32(buf)
        return (a.shape, a.dtype.str, digest)

    def _fit_model(self, x, t, theta):
        """the device model of (x, t, theta), kept until the next different request: L-BFGS asks for the likelihood and
        its gradient at the same theta, one after the other.  The key is the CONTENT of x, t and theta (hashed in full); `clear_cache()` releases the device buffers, and `ml_estimate` does so when it is done."""
        key = (self._signature(x), self._signature(t), _gpx.f64(theta).tobytes())
        cached = getattr(self, "_fit_cache", None)
        if cached is None or cached[0] != key:
            if cached is not None:
                cached[1].close()
            self._fit_cache = None
            cached = (key, self._model(x, t, theta))
            self._fit_cache = cached
        return cached[1]

    def clear_cache(self):
        """release the cached device models (K_NM, Z, W^T and the M x M buffers: about 3 N M doubles)"""
        for name in ("_fit_cache", "_cross_model"):
            cached = getattr(self, name, None)
            if cached is not None:
                cached[1].close()
            setattr(self, name, None)

    close = clear_cache

    def ml_estimate(self, x, t):
        try:
            return Covariance.ml_estimate(self, x, t)
        finally:
            self.clear_cache()

    def _negativeloglikelihood(self, x, t, theta):
        """Snelson's O(N m^2) likelihood (Covariance.py:981-1019); raises LinAlgError like the reference's Cholesky."""
        return self._fit_model(x, t, theta).nll()

    def _d_nll_d_theta(self, x, t, theta):
        """analytic gradient of Snelson's likelihood in O(N m^2) on the GPU (gpx_spgp_nll_grad).  The reference's own
        (Covariance.py:906-979) differentiates the dense N x N likelihood at O(N^2 m) per parameter and is itself only
        checked against central differences (skgpuppy/tests/tests.py:538-565)."""
        return self._fit_model(x, t, theta).nll_grad()

    def __getstate__(self):
        state = Covariance.__getstate__(self)
        state.pop("_cross_model", None)          # device handles never enter a pickle
        state.pop("_fit_cache", None)
        return state
