"""Fixed per-handle overhead: gpx_fit + gpx_predict + gpx_free at a tiny size (the kernels take microseconds)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import numpy as np
import torch
import bench
from skgpuppy_amd import _gpx
lib = _gpx.lib
for N in (128, 2048):
    d = 8
    x, t, xs, th = bench.recipe(N, d, 128)
    xd = torch.as_tensor(x).cuda(); td = torch.as_tensor(t - t.mean()).cuda(); xq = torch.as_tensor(xs).cuda()
    m = torch.empty(128, dtype=torch.float64, device="cuda"); v = torch.empty_like(m)
    vp = lambda a: ctypes.c_void_p(a.data_ptr())
    tf = tp = tfree = 0.0
    reps = 50
    for r in range(reps + 5):
        h = ctypes.c_void_p()
        a = time.perf_counter()
        _gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "fit")
        b = time.perf_counter()
        _gpx.check(lib.gpx_predict(h, vp(xq), 128, vp(m), vp(v)), "predict")
        c = time.perf_counter()
        lib.gpx_free(h)
        e = time.perf_counter()
        if r >= 5:
            tf += b - a; tp += c - b; tfree += e - c
    print("N=%5d: fit %.3f ms  predict %.3f ms  free %.3f ms" % (N, tf / reps * 1e3, tp / reps * 1e3, tfree / reps * 1e3))
