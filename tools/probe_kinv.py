"""Kinv build in isolation (for rocprofv3 traces): fit at C3, then two Exact propagations (the first builds K^-1)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import numpy as np
import torch
import bench
from skgpuppy_amd import _gpx
lib = _gpx.lib
N, d = 16384, 8
x, t, xs, th = bench.recipe(N, d, 16)
xd = torch.as_tensor(x).cuda(); td = torch.as_tensor(t - t.mean()).cuda()
vp = lambda a: ctypes.c_void_p(a.data_ptr())
for rep in range(2):
    h = ctypes.c_void_p()
    _gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "fit")
    u = np.full(d, 5.0); S = 0.01 * np.eye(d); m, v = ctypes.c_double(), ctypes.c_double()
    t0 = time.perf_counter()
    _gpx.check(lib.gpx_propagate_exact(h, _gpx.ptr(u), _gpx.ptr(S), ctypes.byref(m), ctypes.byref(v)), "exact")
    t1 = time.perf_counter()
    _gpx.check(lib.gpx_propagate_exact(h, _gpx.ptr(u + 0.1), _gpx.ptr(S), ctypes.byref(m), ctypes.byref(v)), "exact")
    t2 = time.perf_counter()
    print("exact first (builds Kinv) %.2f ms, second %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
    lib.gpx_free(h)
