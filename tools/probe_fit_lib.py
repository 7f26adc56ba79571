#!/usr/bin/env python3
"""gpx_fit wall time at C3 size through a GIVEN libgpx.so (ctypes only: any library version with gpx_fit / gpx_free):
A/B of two builds on the same box.  usage: probe_fit_lib.py LIB [LIB ...]"""
import ctypes
import sys
import time

import numpy as np


def run(path):
    import torch
    lib = ctypes.CDLL(path)
    lib.gpx_fit.restype = ctypes.c_int
    N, d = 16384, 8
    rng = np.random.RandomState(20240 + N + d)
    x = rng.uniform(0, 10, (N, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
    theta = np.ascontiguousarray(np.log(np.array([2.0, 0.01] + [0.04] * d)))
    dev = torch.device("cuda")
    xd = torch.as_tensor(x).to(dev)
    td = torch.as_tensor(t - t.mean()).to(dev)
    times = []
    for rep in range(10):
        h = ctypes.c_void_p()
        torch.cuda.synchronize()
        a = time.perf_counter()
        st = lib.gpx_fit(ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(td.data_ptr()), ctypes.c_int64(N), ctypes.c_int(d),
                         ctypes.c_void_p(theta.ctypes.data), None, ctypes.byref(h))
        times.append(time.perf_counter() - a)
        assert st == 0, st
        lib.gpx_free(h)
    print("%-40s fit ms: best %.2f median %.2f" % (path[-40:], min(times[2:]) * 1e3, sorted(times[2:])[4] * 1e3), flush=True)


if __name__ == "__main__":
    import subprocess
    if len(sys.argv) == 2:
        run(sys.argv[1])
    else:
        for p in sys.argv[1:]:
            subprocess.run([sys.executable, __file__, p], timeout=300)
