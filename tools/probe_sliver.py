#!/usr/bin/env python3
"""Sliver GEMM (gpx_dev_gemm_nt_sliver) vs the regular small-tile launches of gpx_dev_gemm_nt on the shapes of one chain step
(TRSM: rows x 128, K = 128, in place; update: rows x rows, K = 128): correctness, time alone, time next to the bulk SYRK."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
from skgpuppy_amd import _gpx  # noqa: E402

p = lambda x: ctypes.c_void_p(x.data_ptr())
L = _gpx.lib


def check():
    torch.manual_seed(0)
    for (M, N, K) in [(128, 128, 128), (896, 128, 128), (896, 896, 128), (256, 192, 64), (64, 64, 4), (128, 128, 1024), (2048, 128, 384)]:
        A = torch.randn(M, K + 8, dtype=torch.float64, device="cuda")
        B = torch.randn(N, K + 24, dtype=torch.float64, device="cuda")
        C = torch.randn(M, N + 16, dtype=torch.float64, device="cuda")
        ref = 0.5 * C[:, :N] - 1.25 * A[:, :K] @ B[:, :K].T
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _gpx.check(L.gpx_dev_gemm_nt_sliver(p(A), K + 8, p(B), K + 24, p(C), N + 16, M, N, K, -1.25, 0.5, st), "sliver")
        torch.cuda.synchronize()
        err = (C[:, :N] - ref).abs().max().item() / ref.abs().max().item()
        print("  M=%4d N=%4d K=%4d  rel err %.2e" % (M, N, K, err))
        assert err < 1e-13
    # in place, N = 128: Z <- Z D^T
    Z = torch.randn(896, 128, dtype=torch.float64, device="cuda")
    D = torch.randn(128, 128, dtype=torch.float64, device="cuda")
    ref = Z @ D.T
    _gpx.check(L.gpx_dev_gemm_nt_sliver(p(Z), 128, p(D), 128, p(Z), 128, 896, 128, 128, 1.0, 0.0, st), "sliver in place")
    torch.cuda.synchronize()
    print("  in place rel err %.2e" % ((Z - ref).abs().max().item() / ref.abs().max().item()))
    assert (Z - ref).abs().max().item() / ref.abs().max().item() < 1e-13


def timed(fn, stream, n=40):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    evs[0].record(stream)
    for i in range(n):
        fn()
        evs[i + 1].record(stream)
    return evs


def main():
    check()
    ld = 16384
    Lm = torch.randn(1024, ld, dtype=torch.float64, device="cuda")      # a panel's diagonal square lives in rows of leading dimension ld
    D = torch.randn(128, 128, dtype=torch.float64, device="cuda")
    main_s, side_s = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    ss = ctypes.c_void_p(side_s.cuda_stream)
    n, K = 8192, 1024
    P = torch.randn(n, K, dtype=torch.float64, device="cuda")
    C = torch.zeros(n, n, dtype=torch.float64, device="cuda")
    for rows in (896, 512, 128):
        Z = Lm[128:128 + rows]
        zp = ctypes.c_void_p(Z.data_ptr())
        cp = ctypes.c_void_p(Z.data_ptr() + 128 * 8)
        cases = {
            "trsm  regular": lambda: L.gpx_dev_gemm_nt(zp, ld, p(D), 128, zp, ld, rows, 128, 128, 1.0, 0.0, 0, ss),
            "trsm  sliver ": lambda: L.gpx_dev_gemm_nt_sliver(zp, ld, p(D), 128, zp, ld, rows, 128, 128, 1.0, 0.0, ss),
            "update regular": lambda: L.gpx_dev_gemm_nt(zp, ld, zp, ld, cp, ld, rows, rows, 128, -1e-9, 1.0, 0, ss),
            "update sliver ": lambda: L.gpx_dev_gemm_nt_sliver(zp, ld, zp, ld, cp, ld, rows, rows, 128, -1e-9, 1.0, ss),
        }
        for name, fn in cases.items():
            out = []
            for with_bulk in (0, 1):
                for rep in range(2):
                    Lm.normal_()
                    torch.cuda.synchronize()
                    if with_bulk:
                        for _ in range(4):
                            L.gpx_dev_gemm_nt(p(P), K, p(P), K, p(C), n, n, n, K, -1.0, 1.0, 1, ctypes.c_void_p(main_s.cuda_stream))
                    evs = timed(fn, side_s)
                    torch.cuda.synchronize()
                per = np.array([evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(len(evs) - 1)])
                out.append("%s median %6.1f mean %6.1f max %6.1f us" % ("bulk" if with_bulk else "alone", np.median(per), per.mean(), per.max()))
            print("rows %4d %s: %s | %s" % (rows, name, out[0], out[1]))


if __name__ == "__main__":
    main()
