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
    UncertaintyPropagation2.pyx:189-380).

    With the built-in GaussianCovariance everything runs fused on the device handle (C / J / tr built by a kernel, one pass
    K^-1 [C, J_1..J_d] or two sweeps over the factor, dot products).  With any other operator (the generic route of
    GaussianProcess) C_ux / J_ux / H_ux come from the operator's own __call__ / get_Jacobian / get_Hessian -- 3N host calls,
    as in the reference (:504-510) -- and the reference's N^2 double loops become two device sweeps over the factor
    (gpx_solve) plus O(N d) dot products."""

    def __init__(self, gp):
        UncertaintyPropagationGA.__init__(self, gp)
        self.v = self.gp._get_v()
        self.Winv = self.gp._get_W_inv()
        self.u = None
        self._cjh = None
        self._kv = None            # generic route: K^-1 [C, J_1..J_d] of the cached u

    def _generic(self):
        return self.gp._route() != "gaussian"

    def _new_u(self, u, uu):
        if self.u is None or (np.asarray(self.u) != uu).any():
            self.u = u               # stored by reference like UncertaintyPropagation.py:500
            self._cjh = None
            self._kv = None

    def _parts(self, u, Sigma_x):
        uu, S = _u_sigma(self.gp, u, Sigma_x)
        self._new_u(u, uu)
        if self._generic():
            return self._parts_generic(S)
        out = [ctypes.c_double() for _ in range(4)]
        st = _gpx.lib.gpx_propagate_approx(self.gp._dev().handle, _gpx.ptr(uu), _gpx.ptr(S),
                                           *[ctypes.byref(o) for o in out])
        _gpx.check(st, "gpx_propagate_approx")
        mean, var, sigma2, rest = [o.value for o in out]
        return mean, var, sigma2, rest

    # ---- generic operators ---------------------------------------------------------------------------------------
    def _kinv_rows(self):
        """K^-1 [C, J_1..J_d] for the cached u: two sweeps over the factor on the device"""
        if self._kv is None:
            C, J, _H = self._fetch_cjh()
            self._kv = self.gp._dev().solve(np.vstack([C[None, :], J[:, :, 0].T]))
        return self._kv

    @staticmethod
    def _quadratic_parts(cuu, beta, C, J, H, S, KC, KJ, Ktr=None):
        """(mean without meant, sigma2, variance2 + variance3) from the vectors and their products with K^-1
        (UncertaintyPropagation.py:397-408, :412-433, :435-481); KC = Kinv C, KJ[k] = Kinv J_k, Ktr = Kinv tr (None: Kinv is
        symmetric, C.Ktr = tr.KC)"""
        trace = np.einsum("iab,ba->i", H, S)                    # tracedot(H_i, Sigma_x)
        mean = np.dot(beta, C) + 0.5 * np.dot(beta, trace)
        sigma2 = cuu - np.dot(C, KC)
        sd = np.diag(S)
        var2 = -sum(sd[k] * (np.dot(J[:, k, 0], KJ[k]) - np.dot(beta, J[:, k, 0]) ** 2) for k in range(len(sd)))
        var3 = -np.dot(trace, KC) if Ktr is None else -0.5 * (np.dot(C, Ktr) + np.dot(trace, KC))
        return mean, sigma2, var2 + var3

    def _parts_generic(self, S):
        C, J, H = self._fetch_cjh()
        kv = self._kinv_rows()
        mean, sigma2, rest = self._quadratic_parts(self.gp._covariance(self.u, self.u), self.gp._get_beta(), C, J, H, S, kv[0], kv[1:])
        return mean, sigma2 + rest, sigma2, rest

    def _fetch_cjh(self):
        if self._cjh is None:
            if self.u is None:
                raise AttributeError("C_ux/J_ux/H_ux exist only after a propagate call (as in the reference)")
            gp = self.gp
            if self._generic():
                # (UncertaintyPropagation.py:504-510): the operator's own scalar kernel, Jacobian and Hessian per training point
                x, u = gp.x, self.u
                n = len(x)
                self._cjh = (np.array([gp._covariance(u, x[i]) for i in range(n)], dtype=float),
                             np.array([gp._get_Jacobian(u, x[i]) for i in range(n)], dtype=float).reshape(n, gp.d, 1),
                             np.array([gp._get_Hessian(u, x[i]) for i in range(n)], dtype=float))
                return self._cjh
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

    # The three quadratic-form helpers of the reference (UncertaintyPropagation.py:412-488 / UncertaintyPropagation2.pyx:221-299)
    # take explicit Kinv / x / beta / C_ux / J_ux / H_ux arrays.  Called the way the reference itself calls them -- with the fitted
    # GP's own Kinv and the caches of the last propagation (or with nothing) -- the sums come from the device cache of `u`.  Any
    # OTHER array is honoured: the vectors are taken from the arguments and their products with the explicit matrix run on the
    # device (gpx_symv), so a caller that passes a different Kinv gets that Kinv's numbers, as in the reference.
    def _is_own(self, Kinv=None, beta=None, C_ux=None, J_ux=None, H_ux=None):
        gp = self.gp
        cj = self._cjh if self._cjh is not None else (None, None, None)
        return ((Kinv is None or Kinv is gp._Kinv) and beta is None and (C_ux is None or C_ux is cj[0]) and
                (J_ux is None or J_ux is cj[1]) and (H_ux is None or H_ux is cj[2]))

    def _explicit_parts(self, u, Sigma_x, Kinv, beta, C_ux, J_ux, H_ux):
        gp = self.gp
        uu, S = _u_sigma(gp, u, Sigma_x)
        self._new_u(u, uu)
        C = np.asarray(C_ux if C_ux is not None else self.C_ux, dtype=float).reshape(gp.n)
        J = np.asarray(J_ux if J_ux is not None else self.J_ux, dtype=float).reshape(gp.n, gp.d, 1)
        H = np.asarray(H_ux if H_ux is not None else self.H_ux, dtype=float).reshape(gp.n, gp.d, gp.d)
        b = np.asarray(beta if beta is not None else gp._get_beta(), dtype=float).reshape(gp.n)
        trace = np.einsum("iab,ba->i", H, S)
        V = _gpx.f64(np.vstack([C[None, :], trace[None, :], J[:, :, 0].T]))
        if Kinv is None or Kinv is gp._Kinv:
            KV = gp._dev().solve(V)                              # the model's own K^-1: two sweeps over its factor
        else:
            M = _gpx.f64(Kinv)
            if M.shape != (gp.n, gp.n):
                raise ValueError("Kinv must be (%d, %d)" % (gp.n, gp.n))
            KV = np.empty_like(V)
            _gpx.check(_gpx.lib.gpx_symv(_gpx.ptr(M), gp.n, _gpx.ptr(V), V.shape[0], _gpx.ptr(KV)), "gpx_symv")
        return self._quadratic_parts(gp._covariance(u, u), b, C, J, H, S, KV[0], KV[2:], KV[1])

    def _get_sigma2(self, u, Kinv=None, x=None, C_ux=None, J_ux=None, H_ux=None):
        """C(u,u) - C^T K^-1 C  (UncertaintyPropagation.py:412-433); needs no Sigma_x"""
        Z = np.zeros((self.gp.d, self.gp.d))
        if self._is_own(Kinv, None, C_ux, J_ux, H_ux):
            return self._parts(u, Z)[2]
        return self._explicit_parts(u, Z, Kinv, None, C_ux, J_ux, H_ux)[1]

    def _get_variance_rest(self, u, Sigma_x, Kinv=None, x=None, beta=None, C_ux=None, J_ux=None, H_ux=None):
        """variance2 + variance3  (UncertaintyPropagation.py:435-481)"""
        if self._is_own(Kinv, beta, C_ux, J_ux, H_ux):
            return self._parts(u, Sigma_x)[3]
        return self._explicit_parts(u, Sigma_x, Kinv, beta, C_ux, J_ux, H_ux)[2]

    def _get_sigma2_and_variance_rest(self, u, Sigma_x, Kinv=None, x=None, beta=None):
        """(UncertaintyPropagation.py:483-488)"""
        if self._is_own(Kinv, beta):
            _m, _var, sigma2, rest = self._parts(u, Sigma_x)
            return sigma2, rest
        _m, sigma2, rest = self._explicit_parts(u, Sigma_x, Kinv, beta, None, None, None)
        return sigma2, rest

    def _getFactor(self, u, Sigma_x, v):
        # (UncertaintyPropagation.py:526-560)
        _m, _var, sigma2, rest = self._parts(u, Sigma_x)
        return (v - sigma2) / rest

    def _get_variance_dv_h(self, u, h):
        # (UncertaintyPropagation.py:564-630); all d values come out of one device call
        uu = _gpx.f64(u)
        self._new_u(u, uu)
        if self._generic():
            C, J, H = self._fetch_cjh()
            kv = self._kinv_rows()
            beta = self.gp._get_beta()
            v2 = -(np.dot(J[:, h, 0], kv[1 + h]) - np.dot(beta, J[:, h, 0]) ** 2)
            return v2 - np.dot(kv[0], H[:, h, h])
        out = np.empty(self.gp.d)
        st = _gpx.lib.gpx_propagate_dvh(self.gp._dev().handle, _gpx.ptr(uu), _gpx.ptr(out))
        _gpx.check(st, "gpx_propagate_dvh")
        return out[h]


class _ExplicitInverse(object):
    """Owner of one gpx_kinv_model: an explicit K^-1 / beta resident in HBM (gpx.h)."""

    def __init__(self, Kinv, beta, n):
        self._h = ctypes.c_void_p()
        _gpx.check(_gpx.lib.gpx_kinv_model_create(_gpx.ptr(Kinv), _gpx.ptr(beta), n, ctypes.byref(self._h)), "gpx_kinv_model_create")

    @property
    def handle(self):
        if not self._h:
            raise RuntimeError("explicit-inverse model already released")
        return self._h

    def close(self, _free=_gpx.lib.gpx_kinv_model_free, _null=ctypes.c_void_p):
        if getattr(self, "_h", None):
            _free(self._h)
            self._h = _null()

    __del__ = close


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

    def _generic_moments(self, u, Sigma_x, C_ux, want_var):
        """Any operator but the fused built-in kernel (a from-scratch Covariance, a subclass that overrides a matrix builder, SPGP):
        the reference's class talks to the GP only through _get_beta / _get_W_inv / _inv_cov_matrix / _covariance / x
        (UncertaintyPropagation.py:269-290, :323-379), so it returns numbers for such a GP -- Girard's correction factors built from
        theta[2:2+d], applied to the operator's own C(u, x_i).  C_ux comes from the operator's scalar kernel on the host (N calls, as
        in the reference), the N and N^2 sums run on the device over the model's K^-1 (gpx_propagate_exact_matrix)."""
        gp = self.gp
        uu, S = _u_sigma(gp, u, Sigma_x)
        x = _gpx.f64(gp.x)
        if C_ux is None:
            C_ux = np.array([gp._covariance(u, gp.x[i]) for i in range(gp.n)])
        C = _gpx.f64(C_ux).reshape(gp.n)
        w = _gpx.f64(np.diag(self.Winv))
        cuu = float(gp._covariance(u, u)) if want_var else 0.0
        mean, var = ctypes.c_double(), ctypes.c_double()
        pvar = ctypes.byref(var) if want_var else None      # mean only: the library neither builds nor reads K^-1
        if gp._route() in ("generic", "gaussian"):
            # (any fitted handle: a gpx_fit_matrix one, or -- a caller-supplied C_ux on the built-in route -- the gpx_fit one)
            st = _gpx.lib.gpx_propagate_exact_matrix(gp._dev().handle, None, None, _gpx.ptr(x), gp.n, gp.d, _gpx.ptr(w), _gpx.ptr(C),
                                                     _gpx.ptr(uu), _gpx.ptr(S), cuu, ctypes.byref(mean), pvar)
        else:
            # (SPGP: the model's dense K^-1 -- what the reference's class reads through _inv_cov_matrix -- and beta = Kinv t: kept on the
            # device across calls, uploaded once per Kinv array: gpx_kinv_model_*)
            st = _gpx.lib.gpx_propagate_exact_model(self._explicit_model(gp), _gpx.ptr(x), gp.d, _gpx.ptr(w), _gpx.ptr(C), _gpx.ptr(uu),
                                                    _gpx.ptr(S), cuu, ctypes.byref(mean), pvar)
        _gpx.check(st, "gpx_propagate_exact_matrix")
        return mean.value, var.value

    def _explicit_model(self, gp):
        """device copy of the GP's dense Kinv attribute and beta (padded, symmetrised), rebuilt only when the GP holds another Kinv array"""
        Kinv = gp._inv_cov_matrix()
        key = (id(Kinv), Kinv.__array_interface__["data"][0], Kinv.shape)
        cached = getattr(self, "_kinv_model", None)
        if cached is not None and cached[0] == key:
            return cached[1].handle
        self._release_explicit_model()
        self._kinv_model = (key, _ExplicitInverse(_gpx.f64(Kinv), _gpx.f64(gp._get_beta()), gp.n), Kinv)   # (Kinv kept alive: its id is the key)
        return self._kinv_model[1].handle

    def _release_explicit_model(self):
        cached = getattr(self, "_kinv_model", None)
        if cached is not None:
            cached[1].close()
            self._kinv_model = None

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_kinv_model", None)          # device handles never enter a pickle
        return state

    def propagate_mean(self, u, Sigma_x, C_ux=None):
        # (UncertaintyPropagation.py:269-290).  A caller-supplied C_ux is USED, as in the reference's class (sum_i beta_i C_ux_i corr_i):
        # on the built-in route too it goes through gpx_propagate_exact_matrix on the fitted handle; without one the device builds
        # C(u, x_i) itself (gpx_exact_mean).
        self._set_constants(u, np.asarray(Sigma_x, dtype=float))
        if self.gp._route() != "gaussian" or C_ux is not None:
            return np.float64(self._generic_moments(u, Sigma_x, C_ux, False)[0])
        uu, S = _u_sigma(self.gp, u, Sigma_x)
        out = ctypes.c_double()
        st = _gpx.lib.gpx_exact_mean(self.gp._dev().handle, _gpx.ptr(uu), _gpx.ptr(S), ctypes.byref(out))
        _gpx.check(st, "gpx_exact_mean")
        return np.float64(out.value)

    def propagate_GA(self, u, Sigma_x):
        # (UncertaintyPropagation.py:323-379)
        uu, S = _u_sigma(self.gp, u, Sigma_x)
        self._set_constants(uu, S)
        if self.gp._route() != "gaussian":
            mean, var = self._generic_moments(u, Sigma_x, None, True)
            return np.float64(mean + self.gp._get_mean_t()), np.float64(var)
        mean, var = ctypes.c_double(), ctypes.c_double()
        st = _gpx.lib.gpx_propagate_exact(self.gp._dev().handle, _gpx.ptr(uu), _gpx.ptr(S), ctypes.byref(mean),
                                          ctypes.byref(var))
        _gpx.check(st, "gpx_propagate_exact")
        return np.float64(mean.value + self.gp._get_mean_t()), np.float64(var.value)
