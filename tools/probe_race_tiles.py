#!/usr/bin/env python3
"""Where do two factorisations of the same matrix differ?  (diagnostic for a race in the stream schedule)  Fits the same problem
repeatedly, keeps the first factor on the device and lists the 1024 x 1024 blocks of L in which a later one deviates."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import skgpuppy_amd as sk  # noqa: E402
from skgpuppy_amd import _gpx  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
x, t, xs, theta = bench.recipe(N, 8, 16)
ref = None
CH = 4096
for r in range(reps):
    gp = sk.GaussianProcess(x, t, sk.GaussianCovariance(), theta.copy())
    h = gp._dev().handle
    cur = torch.empty((N, N), dtype=torch.float64, device="cuda")
    for r0 in range(0, N, CH):
        _gpx.check(_gpx.lib.gpx_chol_rows(h, r0, min(N, r0 + CH), ctypes.c_void_p(cur[r0].data_ptr())), "chol_rows")
    torch.cuda.synchronize()
    gp._dev().close()
    if ref is None:
        ref = cur
        print("run 0: reference factor kept")
        continue
    bad = []
    for bi in range(0, N, 1024):
        d = (cur[bi:bi + 1024] - ref[bi:bi + 1024]).abs()
        if float(d.max()) > 0:
            cols = torch.nonzero(d.reshape(d.shape[0], -1, 1024).amax(dim=(0, 2)) > 0).flatten().tolist()
            bad.append((bi // 1024, cols[:12], float(d.max())))
    print("run %d: %d block rows differ" % (r, len(bad)), bad[:10])
    del cur
