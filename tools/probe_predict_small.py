#!/usr/bin/env python3
"""gpx_predict for a handful of queries at C3 size through a GIVEN libgpx.so (ctypes only): milliseconds per call by query count.
usage: probe_predict_small.py LIB [LIB ...]   (each library in a fresh process)"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np


def run(path):
    import torch
    lib = ctypes.CDLL(path)
    lib.gpx_fit.restype = ctypes.c_int
    lib.gpx_predict.restype = ctypes.c_int
    N, d = int(os.environ.get("PROBE_N", "16384")), 8
    rng = np.random.RandomState(3)
    x = rng.uniform(0, 10, (N, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
    theta = np.ascontiguousarray(np.log(np.array([2.0, 0.01] + [0.04] * d)))
    dev = torch.device("cuda")
    xd, td = torch.as_tensor(x).to(dev), torch.as_tensor(t - t.mean()).to(dev)
    xs = torch.as_tensor(rng.uniform(0, 10, (256, d))).to(dev)
    mean_d = torch.empty(256, dtype=torch.float64, device=dev)
    var_d = torch.empty(256, dtype=torch.float64, device=dev)
    vp = lambda tt: ctypes.c_void_p(tt.data_ptr())
    h = ctypes.c_void_p()
    assert lib.gpx_fit(vp(xd), vp(td), ctypes.c_int64(N), ctypes.c_int(d), ctypes.c_void_p(theta.ctypes.data), None, ctypes.byref(h)) == 0
    out = []
    for m in (1, 8, 32, 33, 128, 256):
        best = 1e9
        for rep in range(8):
            torch.cuda.synchronize()
            a = time.perf_counter()
            assert lib.gpx_predict(h, vp(xs), ctypes.c_int64(m), vp(mean_d), vp(var_d)) == 0
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - a)
        out.append("M=%d %.3f ms (mean[0] %.12f var[0] %.12f)" % (m, best * 1e3, float(mean_d[0]), float(var_d[0])))
    lib.gpx_free(h)
    print("RESULT " + " | ".join(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        for path in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, "run", path], capture_output=True, text=True, timeout=600)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
            print(path[-40:], line[-1] if line else "FAILED " + r.stderr[-300:], flush=True)
