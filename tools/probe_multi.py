"""gpx_multi_fit / gpx_multi_predict (csrc/multi.hip: one process, several logical ranks) on the ONE GPU of the box: wall time per fit for
1, 2, 4 ranks sharing the chip, next to gpx_fit -- a functional rehearsal at full size (every rank holds its own factor copy; ranks on
one GPU share its CUs, so more ranks cannot be faster here).  usage: probe_multi.py N d ndev [ndev ...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import numpy as np
import torch  # noqa: F401
from skgpuppy_amd import _gpx
lib = _gpx.lib
N, d = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(20240 + N + d)
x = rng.uniform(0, 10, (N, d)); t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N); xs = rng.uniform(0, 10, (4096, d))
theta = np.ascontiguousarray(np.log(np.array([2.0, 0.01] + [0.04] * d)))
tc = np.ascontiguousarray(t - t.mean())
reps = 2 if N > 30000 else 5
ref = None
h = ctypes.c_void_p()
best = 1e9
for rep in range(reps):
    a = time.perf_counter()
    _gpx.check(lib.gpx_fit(_gpx.ptr(x), _gpx.ptr(tc), N, d, _gpx.ptr(theta), None, ctypes.byref(h)), "gpx_fit")
    best = min(best, time.perf_counter() - a)
    ref = np.empty(N); _gpx.check(lib.gpx_alpha(h, _gpx.ptr(ref)), "alpha")
    lib.gpx_free(h)
print("N=%d d=%d  gpx_fit (host arrays in)            %9.2f ms" % (N, d, best * 1e3), flush=True)
for ndev in [int(a) for a in sys.argv[3:]]:
    devs = (ctypes.c_int * ndev)(*([0] * ndev))
    best = bp = 1e9
    for rep in range(reps):
        m = ctypes.c_void_p()
        a = time.perf_counter()
        _gpx.check(lib.gpx_multi_fit(_gpx.ptr(x), _gpx.ptr(tc), N, d, _gpx.ptr(theta), devs, ndev, ctypes.byref(m)), "gpx_multi_fit")
        b = time.perf_counter()
        mean = np.empty(len(xs)); var = np.empty(len(xs))
        _gpx.check(lib.gpx_multi_predict(m, _gpx.ptr(xs), len(xs), _gpx.ptr(mean), _gpx.ptr(var)), "predict")
        c = time.perf_counter()
        beta = np.empty(N); _gpx.check(lib.gpx_multi_alpha(m, _gpx.ptr(beta)), "alpha")
        lib.gpx_multi_free(m)
        best, bp = min(best, b - a), min(bp, c - b)
    print("N=%d d=%d  gpx_multi_fit with %d rank(s) on one GPU %9.2f ms   predict(4096 queries) %8.2f ms   max |alpha - gpx_fit's| %.1e" % (
        N, d, ndev, best * 1e3, bp * 1e3, np.abs(beta - ref).max()), flush=True)
