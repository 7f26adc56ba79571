"""User-defined covariance operators for the operator-interface tests: TEST INPUTS (the same two operators are defined inline in
tools/gen_golden.py --generic on top of the genuine reference's base classes).  Built on whichever base classes they are handed --
the product's (skgpuppy_amd.Covariance / GaussianCovariance) or the oracle's (OracleCovariance / OracleGaussianCovariance)."""
import numpy as np


def make_rational_quadratic(Base):
    class RationalQuadratic(Base):
        """k = v (1 + r^2 / (2 a l^2))^-a + vt [xi == xj],  theta = log (v, vt, l, a): only __call__ and get_theta"""

        def __call__(self, xi, xj, theta):
            v, vt, ell, a = np.exp(theta)
            diff = np.asarray(xi, dtype=float) - np.asarray(xj, dtype=float)
            r2 = np.dot(diff, diff)
            return v * (1.0 + r2 / (2.0 * a * ell * ell)) ** (-a) + (vt if (np.asarray(xi) == np.asarray(xj)).all() else 0.0)

        def get_theta(self, x, t):
            return np.log(np.array([np.var(t), np.var(t) / 4, 1.0, 1.0]))

    return RationalQuadratic


def make_warped_gaussian(GaussianBase):
    class WarpedGaussian(GaussianBase):
        """the built-in kernel with its cross-covariance modulated by the positive definite factor 1 + 0.1 cos(xi_0 - xj_0)"""

        def cov_matrix_ij(self, xi, xj, theta):
            K = GaussianBase.cov_matrix_ij(self, xi, xj, theta)
            a = np.asarray(xi, dtype=float)[:, 0][:, None]
            b = np.asarray(xj, dtype=float)[:, 0][None, :]
            return K * (1.0 + 0.1 * np.cos(a - b))

    return WarpedGaussian
