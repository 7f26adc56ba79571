"""Row panel of K^-1 (gpx_kinv_rows) at C3 by panel height: time against the whole-matrix build."""
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
for m in (2048, 4096, 8192, 2048, 4096):
    h = ctypes.c_void_p()
    _gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "fit")
    out = torch.empty((m, N), dtype=torch.float64, device="cuda")
    r0 = 4096
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _gpx.check(lib.gpx_kinv_rows(h, r0, r0 + m, vp(out)), "kinv_rows")
    t1 = time.perf_counter()
    print("rows [%d, %d): %.2f ms  (2 m N^2 = %.2f TFLOP -> %.1f TFLOP/s)" % (r0, r0 + m, (t1 - t0) * 1e3, 2 * m * N * N / 1e12, 2 * m * N * N / (t1 - t0) / 1e12), flush=True)
    lib.gpx_free(h)
