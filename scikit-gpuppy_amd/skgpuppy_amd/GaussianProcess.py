"""GaussianProcess -- host-side mirror of skgpuppy/GaussianProcess.py over the libgpx handle.

The fitted model (Cholesky factor of K, inverted diagonal blocks, alpha, lazily K^-1) lives in HBM
behind a gpx_handle; this object keeps the reference's public attributes (x, n, d, meant, t, cov,
theta_min, Kinv) and stays picklable (the handle is rebuilt lazily after unpickling, cf. the
reference's pickle test skgpuppy/tests/tests.py:626-659).
"""
import ctypes

import numpy as np

from . import _gpx
from .Covariance import GaussianCovariance, SPGPCovariance, _MatrixModel


class _DeviceModel(object):
    """Owner of one gpx_handle (single stream, single owner)."""

    def __init__(self, x, t_centered, theta, stream=None):
        self.n, self.d = x.shape
        self._h = ctypes.c_void_p()
        st = _gpx.lib.gpx_fit(_gpx.ptr(x), _gpx.ptr(t_centered), self.n, self.d, _gpx.ptr(theta),
                              ctypes.c_void_p(stream or 0), ctypes.byref(self._h))
        _gpx.check(st, "gpx_fit")

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("device model already released")
        return self._h

    def close(self, _free=_gpx.lib.gpx_free, _null=ctypes.c_void_p):
        # (defaults bound at definition time: the module globals may already be gone when __del__ runs at interpreter exit)
        if getattr(self, "_h", None):
            _free(self._h)
            self._h = _null()

    __del__ = close

    def predict(self, xs):
        m = xs.shape[0]
        mean = np.empty(m)
        var = np.empty(m)
        _gpx.check(_gpx.lib.gpx_predict(self.handle, _gpx.ptr(xs), m, _gpx.ptr(mean), _gpx.ptr(var)), "gpx_predict")
        return mean, var

    def alpha(self):
        out = np.empty(self.n)
        _gpx.check(_gpx.lib.gpx_alpha(self.handle, _gpx.ptr(out)), "gpx_alpha")
        return out

    def solve(self, B, want_linv=False):
        """K^-1 B^T for the rows of B (and L^-1 B^T): two sweeps over the factor, no K^-1."""
        B = _gpx.f64(np.atleast_2d(B))
        if B.shape[1] != self.n:
            raise ValueError("right-hand sides must be rows of length %d" % self.n)
        kb = np.empty_like(B)
        lb = np.empty_like(B) if want_linv else None
        _gpx.check(_gpx.lib.gpx_solve(self.handle, _gpx.ptr(B), B.shape[0], _gpx.ptr(lb) if want_linv else None, _gpx.ptr(kb)),
                   "gpx_solve")
        return (kb, lb) if want_linv else kb

    def chol_mul(self, Z):
        """L z for the rows of Z (a draw from N(0, K) per standard-normal row)."""
        Z = _gpx.f64(np.atleast_2d(Z))
        if Z.shape[1] != self.n:
            raise ValueError("vectors must be rows of length %d" % self.n)
        out = np.empty_like(Z)
        _gpx.check(_gpx.lib.gpx_chol_mul(self.handle, _gpx.ptr(Z), Z.shape[0], _gpx.ptr(out)), "gpx_chol_mul")
        return out

    def kinv(self):
        out = np.empty((self.n, self.n))
        _gpx.check(_gpx.lib.gpx_kinv(self.handle, _gpx.ptr(out)), "gpx_kinv")
        return out

    def kinv_rows(self, r0, r1):
        """rows [r0, r1) of K^-1 built alone (what a rank of the row-sharded propagation holds)"""
        out = np.empty((r1 - r0, self.n))
        _gpx.check(_gpx.lib.gpx_kinv_rows(self.handle, r0, r1, _gpx.ptr(out)), "gpx_kinv_rows")
        return out

    def chol(self):
        out = np.empty((self.n, self.n))
        _gpx.check(_gpx.lib.gpx_chol(self.handle, _gpx.ptr(out)), "gpx_chol")
        return out

    def chol_rows(self, r0, r1):
        out = np.empty((r1 - r0, self.n))
        _gpx.check(_gpx.lib.gpx_chol_rows(self.handle, r0, r1, _gpx.ptr(out)), "gpx_chol_rows")
        return out

    def logdet(self):
        v = ctypes.c_double()
        _gpx.check(_gpx.lib.gpx_logdet(self.handle, ctypes.byref(v)), "gpx_logdet")
        return v.value

    def jitter(self):
        v = ctypes.c_double()
        _gpx.check(_gpx.lib.gpx_jitter_used(self.handle, ctypes.byref(v)), "gpx_jitter_used")
        return v.value


class GaussianProcess(object):
    """GP regression object (skgpuppy/GaussianProcess.py:11-191)."""

    def __init__(self, x, t, cov, theta_min=None):
        # (GaussianProcess.py:19-41): x stored as given, targets centred, Kinv <- factorisation on the GPU
        self.x = x
        self.n, self.d = np.shape(x)
        self.meant = np.mean(t)
        self.t = t - self.meant
        self.cov = cov
        if theta_min is not None:
            self.theta_min = theta_min
        else:
            self.theta_min = self.cov.ml_estimate(self.x, self.t)
        self._model = None
        self._Kinv = None
        self._fit()

    # ---- device model management -----------------------------------------------------------
    def _route(self):
        """which device path serves this operator.  The reference's GaussianProcess talks to `cov` only through its interface
        (GaussianProcess.py:39-41, :75-78); here the two built-in kernels have fused paths and EVERY other operator -- a
        from-scratch Covariance, or a subclass of a built-in one that overrides a matrix builder -- takes the generic route:
        its own cov_matrix / cov_matrix_ij, factored and solved on the GPU (gpx_fit_matrix / gpx_predict_kv)."""
        cov = self.cov
        if type(cov) is SPGPCovariance:
            return "spgp"
        if isinstance(cov, GaussianCovariance) and cov._fused():
            return "gaussian"
        return "generic"

    def _fit(self):
        route = self._route()
        if route == "spgp":
            # low-rank fit: no N x N matrix (the reference's own "TODO Optimize for the SPGP covariance function")
            self._model = self.cov._model(self.x, self.t, self.theta_min)
        elif route == "gaussian":
            self._model = _DeviceModel(_gpx.f64(self.x), _gpx.f64(self.t), _gpx.f64(self.theta_min))
        else:
            self._model = _MatrixModel(np.array(self.cov.cov_matrix(self.x, self.theta_min)), self.t)

    def _dev(self):
        if self._model is None:
            self._fit()
        return self._model

    @property
    def Kinv(self):
        """K^-1 as a NumPy array, materialised on first access (the reference stores it eagerly,
        GaussianProcess.py:41)."""
        if self._Kinv is None:
            self._Kinv = self._dev().kinv()
        return self._Kinv

    @Kinv.setter
    def Kinv(self, value):
        self._Kinv = value

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_model"] = None          # raw device handles never enter the pickle
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._model = None

    # ---- reference API -------------------------------------------------------------------------
    @staticmethod
    def get_realisation(x, cov, theta, size=None):
        """(GaussianProcess.py:44-57): one draw t ~ N(0, cov_matrix(x, theta)).  The reference calls numpy's SVD-based
        multivariate_normal on the host; here K is assembled and factored on the GPU and t = L z with z from numpy's
        global generator (so np.random.seed governs it as in the reference -- the streams themselves differ, the
        distribution is the same).  size=k returns k draws as rows."""
        n, d = np.shape(x)
        if type(cov) is not GaussianCovariance:
            # any other operator (SPGP, a user subclass that overrides cov_matrix): the reference's own route on ITS matrix
            K = cov.cov_matrix(x, theta)
            return np.random.multivariate_normal(np.zeros(n), K, size)
        z = np.random.standard_normal((1 if size is None else int(size), n))
        try:
            model = _DeviceModel(_gpx.f64(x), np.zeros(n), _gpx.f64(theta))
        except np.linalg.LinAlgError:
            model = None
        try:
            if model is not None and model.jitter() == 0.0:
                draws = model.chol_mul(z)
                return draws[0] if size is None else draws
        finally:
            if model is not None:
                model.close()
        # K is numerically semi-definite (noise-free or tiny-vt kernels): the Cholesky route would sample from K + 1e-5 I (the
        # fit's documented retry) or fail.  numpy's SVD-based sampler accepts such K, as in the reference: the Gram matrix still
        # comes from the HIP kernel, the draw uses the z already taken from the global generator's stream position.
        K = cov.cov_matrix(x, theta)
        u_, s_, _vh = np.linalg.svd(K, hermitian=True)
        draws = z.dot((u_ * np.sqrt(np.maximum(s_, 0.0))).T)
        return draws[0] if size is None else draws

    def __call__(self, x_star):
        return self.estimate(x_star)

    def estimate_many(self, x_stars):
        """(GaussianProcess.py:68-80): mean + meant and diag(k - kv Kinv kv^T) for M query points --
        cross-covariance, multi-RHS triangular solve and row reductions on the GPU, no M x M matrix."""
        if self._route() == "generic":
            return self._estimate_many_generic(np.array(x_stars))
        xs = _gpx.f64(np.array(x_stars))
        if xs.ndim != 2 or xs.shape[1] != self.d:
            raise ValueError("x_stars must be (m, %d)" % self.d)
        mean, var = self._dev().predict(xs)
        return mean + self.meant, var

    def _estimate_many_generic(self, x_star, chunk=2048):
        """(GaussianProcess.py:68-80) for any operator: kv = cov.cov_matrix_ij(x_star, x) and the DIAGONAL of
        k = cov.cov_matrix(x_star) come from the operator itself (k in row chunks: its off-diagonal entries are never used),
        the solve against the factor and the row reductions run on the GPU (gpx_predict_kv)."""
        m = len(x_star)
        mean = np.empty(m)
        var = np.empty(m)
        for m0 in range(0, m, chunk):
            xc = x_star[m0:m0 + chunk]
            kdiag = np.diag(np.asarray(self.cov.cov_matrix(xc, self.theta_min), dtype=float)).copy()
            kv = self.cov.cov_matrix_ij(xc, self.x, self.theta_min)
            mean[m0:m0 + chunk], var[m0:m0 + chunk] = self._dev().predict_kv(kv, kdiag)
        return mean + self.meant, var

    def estimate(self, x_star):
        """(GaussianProcess.py:94-111): single-point twin of estimate_many."""
        if self._route() == "generic":
            # k = cov(x*, x*), kv = cov.cov_matrix_ij([x*], x)  (GaussianProcess.py:104-108)
            x_star = np.array(x_star)
            k = self.cov(x_star, x_star, self.theta_min)
            kv = self.cov.cov_matrix_ij(np.atleast_2d(x_star), self.x, self.theta_min)
            mean, var = self._dev().predict_kv(kv, [k])
            return mean[0] + self.meant, var[0]
        xs = _gpx.f64(np.atleast_2d(np.array(x_star)))
        mean, var = self._dev().predict(xs)
        return mean[0] + self.meant, var[0]

    def _get_beta(self):
        # beta = K^-1 t (GaussianProcess.py:114-119)
        if self._route() == "spgp":
            return np.dot(self.Kinv, self.t)
        return self._dev().alpha()

    def _get_W_inv(self):
        w = np.exp(self.theta_min[2:self.d + 2])
        return np.diag(w)

    def _get_v(self):
        return np.exp(self.theta_min[0])

    def _get_vt(self):
        return np.exp(self.theta_min[1])

    def _covariance(self, xi, xj, v=None, w=None):
        # (GaussianProcess.py:133-149) incl. the in-place theta mutation when v / w are passed
        theta = self.theta_min
        if v is not None:
            theta[0] = np.log(v)
        if w is not None:
            theta[2:] = np.log(w)
        return self.cov(xi, xj, theta)

    def _inv_cov_matrix(self):
        return self.Kinv

    def _get_mean_t(self):
        return self.meant

    def _get_Hessian(self, u, xi):
        return self.cov.get_Hessian(u, xi, self.theta_min)

    def _get_Jacobian(self, u, xi):
        return self.cov.get_Jacobian(u, xi, self.theta_min)
