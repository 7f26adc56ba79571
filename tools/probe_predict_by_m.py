#!/usr/bin/env python3
"""gpx_predict at C3 size by query count (device pointers, best of 5): ms, and TFLOP/s of N^2 M."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch
from skgpuppy_amd import _gpx
lib = _gpx.lib
N, d = 16384, 8
rng = np.random.RandomState(3)
x = rng.uniform(0, 10, (N, d)); t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
theta = np.ascontiguousarray(np.log(np.array([2.0, 0.01] + [0.04] * d)))
dev = torch.device("cuda")
xd, td = torch.as_tensor(x).to(dev), torch.as_tensor(t - t.mean()).to(dev)
MM = 16384
xs = torch.as_tensor(rng.uniform(0, 10, (MM, d))).to(dev)
mean_d = torch.empty(MM, dtype=torch.float64, device=dev); var_d = torch.empty(MM, dtype=torch.float64, device=dev)
vp = lambda tt: ctypes.c_void_p(tt.data_ptr())
h = ctypes.c_void_p()
_gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(theta), None, ctypes.byref(h)), "fit")
for m in (64, 128, 256, 512, 1024, 2048, 3071, 3072, 4096, 8192, 16384):
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize(); a = time.perf_counter()
        _gpx.check(lib.gpx_predict(h, vp(xs), m, vp(mean_d), vp(var_d)), "predict")
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - a)
    print("M=%6d %8.3f ms  %6.2f TFLOP/s" % (m, best * 1e3, float(N) * N * m / best / 1e12), flush=True)
lib.gpx_free(h)
