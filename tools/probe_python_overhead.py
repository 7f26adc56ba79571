"""Where the Python-class figure of bench.py loses time against the C-ABI figure: gpx_fit on device pointers, on host
pointers, and through GaussianProcess(...)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import numpy as np
import torch
import bench
import skgpuppy_amd as sk
from skgpuppy_amd import _gpx
lib = _gpx.lib
N, d = 16384, 8
x, t, xs, th = bench.recipe(N, d, N)
tc = t - t.mean()
xd = torch.as_tensor(x).cuda(); td = torch.as_tensor(tc).cuda()
vp = lambda a: ctypes.c_void_p(a.data_ptr())
def timeit(f, reps=6):
    best = 1e9
    for r in range(reps):
        torch.cuda.synchronize(); a = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - a)
    return best * 1e3
def fit_dev():
    h = ctypes.c_void_p(); _gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "fit"); lib.gpx_free(h)
def fit_host():
    h = ctypes.c_void_p(); _gpx.check(lib.gpx_fit(_gpx.ptr(x), _gpx.ptr(tc), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "fit"); lib.gpx_free(h)
def fit_py():
    gp = sk.GaussianProcess(x, t, sk.GaussianCovariance(), th.copy()); gp._dev().close()
def fit_only_dev():
    h = ctypes.c_void_p(); _gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "fit"); return h
print("fit+free, device pointers: %.2f ms" % timeit(fit_dev))
print("fit+free, host pointers  : %.2f ms" % timeit(fit_host))
print("GaussianProcess(...)+close: %.2f ms" % timeit(fit_py))
hs = []
a = time.perf_counter(); h = fit_only_dev(); b = time.perf_counter(); lib.gpx_free(h); c = time.perf_counter()
print("fit %.2f ms, free %.3f ms" % ((b - a) * 1e3, (c - b) * 1e3))
