"""Inverse uncertainty propagation -- host-side mirror of skgpuppy/InverseUncertaintyPropagation.py ("next" row f2).

Pure callers of the propagation classes: which input variances v_i give a target output variance at minimal
sampling cost  sum_i c_i / (v_i I_ii).  The analytic solver needs d values of `_get_variance_dv_h` and one
`_getFactor` at the same u -- on this backend all of them come out of ONE pass over K^-1 kept in HBM (the u-cache
of libgpx); the numerical solver drives COBYLA with repeated `propagate_GA` calls, each O(d^2) host work plus one
tiny device launch once u is cached.
"""
import numpy as np

from .UncertaintyPropagation import UncertaintyPropagationApprox, UncertaintyPropagationExact


class InverseUncertaintyPropagation(object):
    """Base class (InverseUncertaintyPropagation.py:13-52): stores the problem."""

    def __init__(self, output_variance, gp, u, c, I, input_variances=None, upga_class=UncertaintyPropagationExact,
                 coestimated=[]):
        self.coestimated = coestimated
        self.gp = gp
        self.upga_class = upga_class
        self.u = u
        self.output_variance = output_variance
        self.c = c
        self.I = I

    def get_best_solution(self):
        pass


def _followers(coestimated):
    """indices tied to the first member of their co-estimation group (v_i = v_lead I_lead / I_i)."""
    return [(group[0], i) for group in coestimated for i in group if i != group[0]]


class InverseUncertaintyPropagationNumerical(InverseUncertaintyPropagation):
    """COBYLA on log-variances with the constraint output_variance - propagate_GA(u, diag(v))[1] >= 0
    (InverseUncertaintyPropagation.py:55-117)."""

    def get_best_solution(self, startvalue=None):
        from scipy.optimize import fmin_cobyla
        c_full = np.asarray(self.c, dtype=float)
        I_full = np.asarray(self.I, dtype=float)
        c, I = c_full, I_full
        free = np.array(list(range(len(c_full))))
        for _lead, i in _followers(self.coestimated):      # same successive np.delete semantics as the reference
            I = np.delete(I, i)
            c = np.delete(c, i)
            free = np.delete(free, i)
        upga = self.upga_class(self.gp)

        def expand_log(vx):
            full = np.zeros(len(c_full))
            for k, m in enumerate(free):
                full[m] = vx[k]
            for lead, i in _followers(self.coestimated):
                full[i] = np.log(np.exp(full[lead]) * I_full[lead] / I_full[i])
            return full

        def constraint(vx):
            return self.output_variance - upga.propagate_GA(self.u, np.diag(np.exp(expand_log(vx))))[1]

        cost = lambda vx: np.sum(c / np.exp(vx) / I)      # noqa: E731
        start = np.ones(len(c)) * -10 if startvalue is None else startvalue
        vx_min = np.exp(fmin_cobyla(cost, start, [constraint]))
        assert (vx_min > 0).all()
        out = np.zeros(len(c_full))
        for k, m in enumerate(free):
            out[m] = vx_min[k]
        for lead, i in _followers(self.coestimated):
            out[i] = out[lead] * I_full[lead] / I_full[i]
        return out


class InverseUncertaintyPropagationApprox(InverseUncertaintyPropagation):
    """Closed form under the approximate propagation (InverseUncertaintyPropagation.py:121-173): weights
    sqrt(c_i / (dv/dv_i I_ii)), scaled by `_getFactor` to hit the target variance."""

    def __init__(self, output_variance, gp, u, c, I, input_variances=None, coestimated=[]):
        InverseUncertaintyPropagation.__init__(self, output_variance, gp, u, c, I, input_variances=input_variances,
                                               coestimated=coestimated, upga_class=UncertaintyPropagationApprox)

    def get_best_solution(self):
        d = len(self.u)
        I = np.asarray(self.I, dtype=float)
        upga = self.upga_class(self.gp)
        dvdv = np.array([upga._get_variance_dv_h(self.u, h) for h in range(d)])
        for lead, i in _followers(self.coestimated):
            dvdv[lead] += dvdv[i] * I[lead] / I[i]
        weights = np.sqrt(np.asarray(self.c, dtype=float) / dvdv / I)
        for lead, i in _followers(self.coestimated):
            weights[i] = weights[lead] * I[lead] / I[i]
        assert (weights > 0).all()
        factor = upga._getFactor(self.u, np.diag(weights), self.output_variance)
        optimum = factor * weights
        assert (optimum > 0).all()
        return optimum
