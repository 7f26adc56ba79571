#!/usr/bin/env python3
"""Leaf kernel (potrf128 + trtri128 in one launch, MFMA out of LDS): correctness against numpy and stand-alone latency."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
from skgpuppy_amd import _gpx  # noqa: E402


def main():
    rng = np.random.RandomState(1)
    B = rng.randn(128, 300)
    A = B @ B.T / 300 + 0.5 * np.eye(128)
    a = torch.as_tensor(A).cuda()
    dinv = torch.zeros(128, 128, dtype=torch.float64, device="cuda")
    diag = torch.zeros(128, dtype=torch.float64, device="cuda")
    info = torch.zeros(4, dtype=torch.int32, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda x: ctypes.c_void_p(x.data_ptr())
    w = a.clone()
    _gpx.check(_gpx.lib.gpx_dev_potrf_leaf(p(w), 128, p(dinv), p(diag), p(info), 0, st), "leaf")
    torch.cuda.synchronize()
    L = np.linalg.cholesky(A)
    print("L err", np.abs(w.cpu().numpy() - L).max(), "inv err",
          np.abs(dinv.cpu().numpy() - np.linalg.inv(L)).max() / np.abs(np.linalg.inv(L)).max(), "diag err",
          np.abs(diag.cpu().numpy() - np.diag(L)).max(), "info", int(info[0]))
    ws = [a.clone() for _ in range(50)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        ws = [a.clone() for _ in range(50)]
        torch.cuda.synchronize()
        e0.record()
        for x in ws:
            _gpx.lib.gpx_dev_potrf_leaf(p(x), 128, p(dinv), p(diag), p(info), 0, st)
        e1.record()
        torch.cuda.synchronize()
    print("  %.1f us per leaf (50 back-to-back launches)" % (e0.elapsed_time(e1) * 1e3 / 50))


if __name__ == "__main__":
    main()
