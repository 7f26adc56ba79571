#!/usr/bin/env python3
"""gpx_dev_syrk_trap (narrow update + bulk SYRK of a panel as one launch with a completion counter for the narrow tiles) against torch
on random data: every tile of the trapezoid, nothing outside it, the final count."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
from skgpuppy_amd import _gpx  # noqa: E402

p = lambda x: ctypes.c_void_p(x.data_ptr())


def main():
    torch.manual_seed(0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for (nt, off, K) in [(48, 8, 1024), (112, 8, 1024), (51, 8, 256), (64, 3, 128), (45, 8, 512)]:
        M, OC = nt * 128, off * 128
        ld = OC + M + 256
        P = torch.randn(OC + M, K + 16, dtype=torch.float64, device="cuda")          # rows of the panel: first OC "top" rows, then M rows
        C = torch.randn(M, ld, dtype=torch.float64, device="cuda")
        C0 = C.clone()
        cnt = torch.zeros(8, dtype=torch.int32, device="cuda")
        A = P[OC:]
        rc = _gpx.lib.gpx_dev_syrk_trap(p(A), K + 16, p(P), K + 16, p(C), ld, M, OC, K, -1.0, 1.0, p(cnt), st)
        torch.cuda.synchronize()
        if rc != 0:
            print("nt=%d off=%d: rc %d (%s)" % (nt, off, rc, _gpx.lib.gpx_last_error()))
            continue
        ref = C0[:, :OC + M] - A[:, :K] @ P[:, :K].T
        i = torch.arange(M, device="cuda")[:, None] // 128
        j = torch.arange(OC + M, device="cuda")[None, :] // 128
        inside = j <= i + off
        want = torch.where(inside, ref, C0[:, :OC + M])
        err = (C[:, :OC + M] - want).abs().max().item()
        outside_ok = torch.equal(C[:, OC + M:], C0[:, OC + M:])
        counts = cnt.tolist()
        print("nt=%3d off=%d K=%4d: max err %.2e  untouched outside: %s  counts %s (want %d per column)" % (nt, off, K, err, outside_ok, counts[:off], nt))
        assert err < 1e-10 and outside_ok and counts[:off] == [nt] * off and counts[off:] == [0] * (8 - off)


if __name__ == "__main__":
    main()
