"""Gram assembly inside gpx_fit (lower-only, N x N) by input dimension: ms and store rate from the library's own event profiler.
usage: GPX_PROFILE=2 [GPX_GRAM_WIDE=0|1] python tools/probe_gram.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
os.environ.setdefault("GPX_PROFILE", "2")
import skgpuppy_amd as sk  # noqa: E402
from skgpuppy_amd import _gpx  # noqa: E402


def main():
    N = 16384
    for d in (2, 8, 15, 16):
        rng = np.random.RandomState(d)
        x = rng.uniform(0, 10, (N, d))
        t = np.sin(x.sum(1))
        theta = np.log(np.array([2.0, 0.01] + [0.04] * d))
        best = None
        for rep in range(3):
            gp = sk.GaussianProcess(x, t, sk.GaussianCovariance(), theta.copy())
            n, ms, work = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
            _gpx.check(_gpx.lib.gpx_profile_read(gp._dev().handle, _gpx.K_GRAM, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(work)), "read")
            if best is None or ms.value < best[1]:
                best = (n.value, ms.value, work.value)
            del gp
        print("d=%2d: %d launches, %.3f ms, %.2f GB stored -> %.2f TB/s" % (d, best[0], best[1], best[2] / 1e9, best[2] / best[1] / 1e9), flush=True)


if __name__ == "__main__":
    main()
