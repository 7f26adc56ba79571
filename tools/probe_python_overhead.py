"""Where the Python-API fit's extra time over the device-resident C-ABI call goes: gpx_fit with device pointers, with pageable host
pointers, with pinned host pointers, and GaussianProcess(...) itself (C3 size)."""
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
xd, td, xsd = torch.as_tensor(x).cuda(), torch.as_tensor(tc).cuda(), torch.as_tensor(xs).cuda()
xp, tp, xsp = torch.as_tensor(x).pin_memory(), torch.as_tensor(tc).pin_memory(), torch.as_tensor(xs).pin_memory()
mean_d = torch.empty(N, dtype=torch.float64, device="cuda"); var_d = torch.empty_like(mean_d)
mean_h = np.empty(N); var_h = np.empty(N)
mean_p = torch.empty(N, dtype=torch.float64).pin_memory(); var_p = torch.empty(N, dtype=torch.float64).pin_memory()
vp = lambda a: ctypes.c_void_p(a.data_ptr())
def cyc(xa, ta, xsa, ma, va):
    best = (1e9, 0, 0)
    for r in range(8):
        h = ctypes.c_void_p()
        torch.cuda.synchronize(); a = time.perf_counter()
        _gpx.check(lib.gpx_fit(xa, ta, N, d, _gpx.ptr(th), None, ctypes.byref(h)), "fit")
        b = time.perf_counter()
        _gpx.check(lib.gpx_predict(h, xsa, N, ma, va), "predict")
        torch.cuda.synchronize(); c = time.perf_counter()
        lib.gpx_free(h)
        if r >= 2 and c - a < best[0]: best = (c - a, b - a, c - b)
    return best
for name, args in (("device pointers", (vp(xd), vp(td), vp(xsd), vp(mean_d), vp(var_d))),
                   ("pageable host pointers", (_gpx.ptr(x), _gpx.ptr(tc), _gpx.ptr(xs), _gpx.ptr(mean_h), _gpx.ptr(var_h))),
                   ("pinned host pointers", (vp(xp), vp(tp), vp(xsp), vp(mean_p), vp(var_p)))):
    tot, f, p = cyc(*args)
    print("%-26s fit %.3f ms  predict %.3f ms" % (name, f * 1e3, p * 1e3), flush=True)
best = (1e9, 0, 0)
for r in range(6):
    a = time.perf_counter()
    gp = sk.GaussianProcess(x, t, sk.GaussianCovariance(), th.copy())
    b = time.perf_counter()
    m, v = gp.estimate_many(xs)
    c = time.perf_counter()
    gp._dev().close()
    if r >= 2 and c - a < best[0]: best = (c - a, b - a, c - b)
print("%-26s fit %.3f ms  predict %.3f ms" % ("GaussianProcess classes", best[1] * 1e3, best[2] * 1e3))
