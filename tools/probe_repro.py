"""Repeated fits of one problem: jitter used, stall refits and a hash of K^-1 t per fit -- the first thing to look at when
tests/test_gpu_parity.py::test_fit_is_bit_reproducible fails (a differing fit WITH jitter means a wrong pivot somewhere, a differing fit
without it a race that left the factor positive definite).

    python tools/probe_repro.py 65536 4 [16384 6 ...]
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scikit-gpuppy_amd"))
import torch  # noqa: F401,E402  (initialises the HIP runtime the way the tests do)
import skgpuppy_amd as sk  # noqa: E402


def recipe(N, d, M):
    rng = np.random.RandomState(20240 + N + d)
    x = rng.uniform(0, 10, (N, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
    xs = rng.uniform(0, 10, (M, d))
    theta = np.log(np.array([2.0, 0.01] + [0.04] * d))
    return x, t, xs, theta


def main():
    args = [int(a) for a in sys.argv[1:]] or [65536, 4]
    for N, reps in zip(args[0::2], args[1::2]):
        x, t, xs, theta = recipe(N, 8, 64)
        first = None
        for r in range(reps):
            gp = sk.GaussianProcess(x, t, sk.GaussianCovariance(), theta.copy())
            beta = gp._get_beta()
            jit = gp._dev().jitter()
            h = hashlib.sha1(np.ascontiguousarray(beta).tobytes()).hexdigest()[:12]
            if first is None:
                first = beta
            print("N=%d fit %d: jitter %g  sha1(beta) %s  max|beta - first| %.3e" % (N, r, jit, h, np.abs(beta - first).max()), flush=True)
            gp._dev().close()


if __name__ == "__main__":
    main()
