#!/usr/bin/env python3
"""Leaf kernel (potrf128 + trtri128 in one launch): correctness against numpy, stand-alone latency, and latency next to a
saturating bulk SYRK on another stream (the situation inside the look-ahead factorisation).
(Round 2's bpermute formulation, once selectable with GPX_LEAF=old, was removed in round 4.)"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
from skgpuppy_amd import _gpx  # noqa: E402

p = lambda x: ctypes.c_void_p(x.data_ptr())


def main():
    rng = np.random.RandomState(1)
    worst = [0.0, 0.0, 0.0]
    for trial in range(6):
        B = rng.randn(128, 130 + 60 * trial)
        A = B @ B.T / B.shape[1] + (0.5 if trial % 2 == 0 else 1e-3) * np.eye(128)
        a = torch.as_tensor(A).cuda()
        dinv = torch.zeros(128, 128, dtype=torch.float64, device="cuda")
        diag = torch.zeros(128, dtype=torch.float64, device="cuda")
        info = torch.zeros(4, dtype=torch.int32, device="cuda")
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        w = a.clone()
        _gpx.check(_gpx.lib.gpx_dev_potrf_leaf(p(w), 128, p(dinv), p(diag), p(info), 0, st), "leaf")
        torch.cuda.synchronize()
        L = np.linalg.cholesky(A)
        Li = np.linalg.inv(L)
        errs = (np.abs(w.cpu().numpy() - L).max() / np.abs(L).max(), np.abs(dinv.cpu().numpy() - Li).max() / np.abs(Li).max(),
                np.abs(diag.cpu().numpy() - np.diag(L)).max())
        worst = [max(a_, b_) for a_, b_ in zip(worst, errs)]
        assert int(info[0]) == 0, int(info[0])
    print("variant %s: worst rel err  L %.2e  inv %.2e  diag %.2e (6 matrices, cond up to ~1e3)" % (
        os.environ.get("GPX_LEAF", "new"), *worst))
    # a non-positive pivot: column 70 (1-based 71 + offset)
    Abad = A.copy()
    Abad[70, 70] = -1.0
    w = torch.as_tensor(Abad).cuda()
    info.zero_()
    _gpx.lib.gpx_dev_potrf_leaf(p(w), 128, p(dinv), p(diag), p(info), 1000, st)
    torch.cuda.synchronize()
    print("bad pivot info", int(info[0]), "(expect 1071)")

    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        ws = [a.clone() for _ in range(50)]
        torch.cuda.synchronize()
        e0.record()
        for x in ws:
            _gpx.lib.gpx_dev_potrf_leaf(p(x), 128, p(dinv), p(diag), p(info), 0, st)
        e1.record()
        torch.cuda.synchronize()
    print("  alone: %.1f us per leaf (50 back-to-back launches)" % (e0.elapsed_time(e1) * 1e3 / 50))

    # next to a bulk SYRK (lower-only, K = 1024, 8192 rows: 2080 tiles of 128 x 128 -- about 2 ms per launch)
    n, K = 8192, 1024
    P = torch.randn(n, K, dtype=torch.float64, device="cuda")
    C = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    main_s, side_s = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    nleaf = 40
    for rep in range(2):
        ws = [a.clone() for _ in range(nleaf)]
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nleaf + 1)]
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ms, ss = ctypes.c_void_p(main_s.cuda_stream), ctypes.c_void_p(side_s.cuda_stream)
        b0.record(main_s)
        for _ in range(4):
            _gpx.check(_gpx.lib.gpx_dev_gemm_nt(p(P), K, p(P), K, p(C), n, n, n, K, -1.0, 1.0, 1, ms), "syrk")
        b1.record(main_s)
        evs[0].record(side_s)
        for i, x in enumerate(ws):
            _gpx.lib.gpx_dev_potrf_leaf(p(x), 128, p(dinv), p(diag), p(info), 0, ss)
            evs[i + 1].record(side_s)
        torch.cuda.synchronize()
    per = np.array([evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(nleaf)])
    bulk_ms = b0.elapsed_time(b1)
    span = evs[0].elapsed_time(evs[-1])
    print("  next to the bulk SYRK (4 launches, %.2f ms; leaves span %.2f ms): per leaf median %.0f  mean %.0f  min %.0f  max %.0f us"
          % (bulk_ms, span, np.median(per), per.mean(), per.min(), per.max()))
    # bulk alone for reference
    torch.cuda.synchronize()
    b0.record(main_s)
    for _ in range(4):
        _gpx.lib.gpx_dev_gemm_nt(p(P), K, p(P), K, p(C), n, n, n, K, -1.0, 1.0, 1, ms)
    b1.record(main_s)
    torch.cuda.synchronize()
    print("  bulk alone: %.2f ms" % b0.elapsed_time(b1))


if __name__ == "__main__":
    main()
