"""Uncertainty propagation -- host-side mirror of skgpuppy/UncertaintyPropagation.py (Approx, Exact).

The reference swaps its pure-Python classes for the compiled Cython twins at import
(skgpuppy/UncertaintyPropagation.py:10-21,244); this module is the third backend: same class names and
methods, the O(N^2) loops K1..K8 run as HIP kernels on the handle of the GaussianProcess.
"""
import ctypes

import numpy as np

from . import _gpx

weaving = False     # the reference's module flags; neither native CPU backend exists here
cython = False
hip = True


class UncertaintyPropagationGA(object):
    """Superclass (UncertaintyPropagation.py:55-85)."""

    def __init__(self, gp):
        self.gp = gp

    def propagate_mean(self, u, Sigma_x):
        return 0

    def propagate_GA(self, u, Sigma_x):
        return 0, 0


def _u_sigma(gp, u, Sigma_x):
    uu = _gpx.f64(u)
    S = _gpx.f64(Sigma_x)
    if uu.shape != (gp.d,) or S.shape != (gp.d, gp.d):
        raise ValueError("u must be (%d,) and Sigma_x (%d, %d)" % (gp.d, gp.d, gp.d))
    return uu, S


class UncertaintyPropagationApprox(UncertaintyPropagationGA):
    """Girard's approximate Gaussian-approximation moments (UncertaintyPropagation.py:386-630 /
    UncertaintyPropagation2.pyx:189-380)."""

    def __init__(self, gp):
        UncertaintyPropagationGA.__init__(self, gp)
        self.v = self.gp._get_v()
        self.Winv = self.gp._get_W_inv()
        self.u = None
        self._cjh = None

    def _parts(self, u, Sigma_x):
        uu, S = _u_sigma(self.gp, u, Sigma_x)
        if self.u is None or (np.asarray(self.u) != uu).any():
            self.u = u               # stored by reference like UncertaintyPropagation.py:500
            self._cjh = None
        out = [ctypes.c_double() for _ in range(4)]
        st = _gpx.lib.gpx_propagate_approx(self.gp._dev().handle, _gpx.ptr(uu), _gpx.ptr(S),
                                           *[ctypes.byref(o) for o in out])
        _gpx.check(st, "gpx_propagate_approx")
        mean, var, sigma2, rest = [o.value for o in out]
        return mean, var, sigma2, rest

    def _fetch_cjh(self):
        if self._cjh is None:
            if self.u is None:
                raise AttributeError("C_ux/J_ux/H_ux exist only after a propagate call (as in the reference)")
            gp = self.gp
            uu = _gpx.f64(self.u)
            C = np.empty(gp.n)
            J = np.empty((gp.n, gp.d))
            H = np.empty((gp.n, gp.d, gp.d))
            st = _gpx.lib.gpx_cjh(gp._dev().handle, _gpx.ptr(uu), _gpx.ptr(C), _gpx.ptr(J), _gpx.ptr(H))
            _gpx.check(st, "gpx_cjh")
            self._cjh = (C, J.reshape(gp.n, gp.d, 1), H)
        return self._cjh

    # the pure-Python reference exposes these caches as attributes (UncertaintyPropagation.py:501-510)
    @property
    def C_ux(self):
        return self._fetch_cjh()[0]

    @property
    def J_ux(self):
        return self._fetch_cjh()[1]

    @property
    def H_ux(self):
        return self._fetch_cjh()[2]

    def propagate_mean(self, u, Sigma_x):
        # (UncertaintyPropagation.py:397-408): beta.C + 1/2 beta.tr(H Sigma)
        return np.float64(self._parts(u, Sigma_x)[0])

    def propagate_GA(self, u, Sigma_x):
        # (UncertaintyPropagation.py:490-523)
        mean, var, _s2, _rest = self._parts(u, Sigma_x)
        return np.float64(mean + self.gp._get_mean_t()), np.float64(var)

    # The three quadratic-form helpers of the reference (UncertaintyPropagation.py:412-488 / UncertaintyPropagation2.pyx:221-299).
    # There they loop over explicit Kinv / x / beta / C_ux / J_ux / H_ux arrays -- always the fitted GP's own and the caches of
    # the last propagation.  Here the sums come from the device cache of `u` (one pass K^-1 [C, J_1..J_d], then dot products);
    # the array arguments are accepted for signature parity and not read.
    def _get_sigma2(self, u, Kinv=None, x=None, C_ux=None, J_ux=None, H_ux=None):
        """C(u,u) - C^T K^-1 C  (UncertaintyPropagation.py:412-433); needs no Sigma_x"""
        return self._parts(u, np.zeros((self.gp.d, self.gp.d)))[2]

    def _get_variance_rest(self, u, Sigma_x, Kinv=None, x=None, beta=None, C_ux=None, J_ux=None, H_ux=None):
        """variance2 + variance3  (UncertaintyPropagation.py:435-481)"""
        return self._parts(u, Sigma_x)[3]

    def _get_sigma2_and_variance_rest(self, u, Sigma_x, Kinv=None, x=None, beta=None):
        """(UncertaintyPropagation.py:483-488)"""
        _m, _var, sigma2, rest = self._parts(u, Sigma_x)
        return sigma2, rest

    def _getFactor(self, u, Sigma_x, v):
        # (UncertaintyPropagation.py:526-560)
        _m, _var, sigma2, rest = self._parts(u, Sigma_x)
        return (v - sigma2) / rest

    def _get_variance_dv_h(self, u, h):
        # (UncertaintyPropagation.py:564-630); all d values come out of one device call
        uu = _gpx.f64(u)
        if self.u is None or (np.asarray(self.u) != uu).any():
            self.u = u
            self._cjh = None
        out = np.empty(self.gp.d)
        st = _gpx.lib.gpx_propagate_dvh(self.gp._dev().handle, _gpx.ptr(uu), _gpx.ptr(out))
        _gpx.check(st, "gpx_propagate_dvh")
        return out[h]


class UncertaintyPropagationExact(UncertaintyPropagationGA):
    """Girard's exact Gaussian-approximation moments (UncertaintyPropagation.py:246-379 /
    UncertaintyPropagation2.pyx:57-184)."""

    # scalar correction factors of Girard's exact moments -- O(d^2) host arithmetic exactly as the reference defines them
    # (UncertaintyPropagation.py:247-266, :292-321); their N and N^2-fold evaluation is the device kernels' job
    def _prepare_C_corr(self, D):
        I = np.eye(D)
        self.Deltainv = self.Winv - np.diag(np.array([self.Winv[i][i] / (1 + self.Winv[i][i] * self.Sigma_x[i][i]) for i in range(D)]))
        self.normalize_C_corr = 1 / np.sqrt(np.linalg.det(I + self.Winv * self.Sigma_x))

    def _get_C_corr(self, u, xi):
        diff = np.asarray(u, dtype=float) - np.asarray(xi, dtype=float)
        return self.normalize_C_corr * np.exp(0.5 * (np.dot(diff.T, np.dot(self.Deltainv, diff))))

    def _prepare_C_corr2(self, D):
        I = np.eye(D)
        W = np.diag(np.array([1 / self.Winv[i][i] for i in range(D)]))
        self.LambdaInv = 2 * self.Winv - np.linalg.inv(0.5 * W + self.Sigma_x)
        self.normalize_C_corr2 = 1 / np.sqrt(np.linalg.det(2 * self.Winv * self.Sigma_x + I))

    def _get_C_corr2(self, u, x):
        diff = np.asarray(u, dtype=float) - np.asarray(x, dtype=float)
        return self.normalize_C_corr2 * np.exp(0.5 * (np.dot(diff.T, np.dot(self.LambdaInv, diff))))

    def _set_constants(self, u, Sigma_x):
        # attributes the reference's propagate_* leave behind (UncertaintyPropagation.py:279-283, :326-334)
        self.Winv = self.gp._get_W_inv()
        self.Sigma_x = Sigma_x
        with np.errstate(all="ignore"):
            self._prepare_C_corr(len(u))
            self._prepare_C_corr2(len(u))

    def propagate_mean(self, u, Sigma_x, C_ux=None):
        # C_ux is accepted for signature parity (UncertaintyPropagation.py:269); it is rebuilt on device
        self._set_constants(u, np.asarray(Sigma_x, dtype=float))
        uu, S = _u_sigma(self.gp, u, Sigma_x)
        out = ctypes.c_double()
        st = _gpx.lib.gpx_exact_mean(self.gp._dev().handle, _gpx.ptr(uu), _gpx.ptr(S), ctypes.byref(out))
        _gpx.check(st, "gpx_exact_mean")
        return np.float64(out.value)

    def propagate_GA(self, u, Sigma_x):
        # (UncertaintyPropagation.py:323-379)
        uu, S = _u_sigma(self.gp, u, Sigma_x)
        self._set_constants(uu, S)
        mean, var = ctypes.c_double(), ctypes.c_double()
        st = _gpx.lib.gpx_propagate_exact(self.gp._dev().handle, _gpx.ptr(uu), _gpx.ptr(S), ctypes.byref(mean),
                                          ctypes.byref(var))
        _gpx.check(st, "gpx_propagate_exact")
        return np.float64(mean.value + self.gp._get_mean_t()), np.float64(var.value)
