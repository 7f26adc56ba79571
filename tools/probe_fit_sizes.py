"""gpx_fit (device-resident inputs, best of 5) by training-set size: how the schedule's thresholds (CU reservation, trapezoid launch),
tuned at N = 16384, behave at other sizes.   usage: [GPX_RESERVE_CUS=.. GPX_RESERVE_TILES=..] python tools/probe_fit_sizes.py [N ...]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import numpy as np
import torch
from skgpuppy_amd import _gpx
lib = _gpx.lib
vp = lambda a: ctypes.c_void_p(a.data_ptr())
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 12288, 16384, 24576, 32768]
d = 8
out = []
for N in sizes:
    rng = np.random.RandomState(N)
    x = rng.uniform(0, 10, (N, d)); t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N); t -= t.mean()
    th = np.log(np.array([2.0, 0.01] + [0.04] * d))
    xd, td = torch.as_tensor(x).cuda(), torch.as_tensor(t).cuda()
    best = 1e9
    for rep in range(6):
        torch.cuda.synchronize(); a = time.perf_counter()
        h = ctypes.c_void_p(); _gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "fit")
        torch.cuda.synchronize(); b = time.perf_counter()
        lib.gpx_free(h)
        if rep: best = min(best, b - a)
    out.append("N=%d %.2f ms (%.1f TFLOP/s)" % (N, best * 1e3, N ** 3 / 3.0 / best / 1e12))
print("  ".join(out), flush=True)
