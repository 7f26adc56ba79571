#!/usr/bin/env python3
"""Dataflow factorisation kernel (gpx_dev_chol_dataflow, csrc/dflow.hip): correctness against numpy on small shapes, then the fit's
time at C3 size for the multi-stream schedule ("off"), the whole-matrix dataflow launch ("whole": GPX_DFLOW_MAX_BLOCKS above the matrix)
or any variant given as environment assignments A=1,B=2 (switches are read once per process: one subprocess per setting).  Round 4's
hand-over panels (GPX_DFLOW_FROM) and the bulk-only diagnostic went with round 5's pruning; `bulk` below needs a round-4 library.
usage: probe_dflow.py [check|time [off|whole|A=1,B=2 ...]|fit LABEL]"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))


def check():
    import torch
    from skgpuppy_amd import _gpx
    L = _gpx.lib
    dev = torch.device("cuda")
    rng = np.random.RandomState(1)
    for nb in [int(v) for v in os.environ.get("PROBE_NB", "1,3,8,9,16,20,37").split(",")]:
        n = 128 * nb
        B = rng.randn(n, 64)
        A = B.dot(B.T) / 64.0 + np.diag(rng.uniform(1.0, 2.0, n))
        ref = np.linalg.cholesky(A)
        Ad = torch.as_tensor(A).to(dev).contiguous()
        dinv = torch.zeros(nb * 128 * 128, dtype=torch.float64, device=dev)
        diag = torch.zeros(n, dtype=torch.float64, device=dev)
        info = torch.zeros(4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = L.gpx_dev_chol_dataflow(ctypes.c_void_p(Ad.data_ptr()), n, nb, 0, ctypes.c_void_p(dinv.data_ptr()), ctypes.c_void_p(diag.data_ptr()),
                                     ctypes.c_void_p(info.data_ptr()), None)
        dt = time.perf_counter() - t0
        _gpx.check(st, "gpx_dev_chol_dataflow")
        got = np.tril(Ad.cpu().numpy())
        err = np.abs(got - ref).max() / np.abs(ref).max()
        dv = dinv.cpu().numpy().reshape(nb, 128, 128)
        derr = max(np.abs(dv[k].dot(ref[128 * k:128 * k + 128, 128 * k:128 * k + 128]) - np.eye(128)).max() for k in range(nb))
        print("nb=%3d n=%5d  info=%s  max|L - ref|/max|ref| = %.2e  max|Dinv L_kk - I| = %.2e  diag err %.2e  %.2f ms" % (
            nb, n, info.cpu().numpy()[:2], err, derr, np.abs(diag.cpu().numpy() - np.diag(ref)).max(), dt * 1e3), flush=True)
        assert info.cpu().numpy()[1] == 0 and err < 1e-11 and derr < 1e-9, "dataflow factor wrong"


def fit_once(label):
    import torch   # first: torch's bundled HIP runtime and libgpx must share one libamdhip64
    from skgpuppy_amd import _gpx
    N, d = int(os.environ.get("PROBE_N", "16384")), 8
    rng = np.random.RandomState(20240 + N + d)
    x = rng.uniform(0, 10, (N, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
    theta = np.log(np.array([2.0, 0.01] + [0.04] * d))
    dev = torch.device("cuda")
    xd = torch.as_tensor(x).to(dev)
    td = torch.as_tensor(t - t.mean()).to(dev)
    times = []
    beta = None
    for rep in range(8):
        h = ctypes.c_void_p()
        torch.cuda.synchronize()
        a = time.perf_counter()
        _gpx.check(_gpx.lib.gpx_fit(ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(td.data_ptr()), N, d, _gpx.ptr(theta), None, ctypes.byref(h)), "gpx_fit")
        times.append(time.perf_counter() - a)
        if rep == 7:
            beta = np.empty(N)
            _gpx.check(_gpx.lib.gpx_alpha(h, _gpx.ptr(beta)), "alpha")
        _gpx.lib.gpx_free(h)
    print("%-14s fit ms: best %.2f median %.2f  (all: %s)  |alpha| %.6e" % (label, min(times[2:]) * 1e3, sorted(times[2:])[3] * 1e3,
                                                                           " ".join("%.1f" % (v * 1e3) for v in times), np.abs(beta).sum()), flush=True)
    np.save(os.path.join(ROOT, "gpurun_out", "probe_dflow_beta_%s.npy" % label), beta)


def bulk():
    """the persistent loop's own tile rate (GPX_DFLOW_BULKONLY=1: panel 0's trailing tiles only, no dependencies) against the
    hardware-dispatched trapezoid launch on the same tiles"""
    import torch
    from skgpuppy_amd import _gpx
    L = _gpx.lib
    dev = torch.device("cuda")
    nb = int(os.environ.get("PROBE_NB", "120"))
    n = 128 * nb
    A = torch.rand((n, n), dtype=torch.float64, device=dev) * 1e-3
    dinv = torch.zeros(nb * 128 * 128, dtype=torch.float64, device=dev)
    diag = torch.zeros(n, dtype=torch.float64, device=dev)
    info = torch.zeros(4, dtype=torch.int32, device=dev)
    nt = nb - 16
    tiles = nt * 8 + nt * (nt + 1) // 2
    p = lambda t_, off=0: ctypes.c_void_p(t_.data_ptr() + 8 * off)
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        st = L.gpx_dev_chol_dataflow(p(A), n, nb, 0, p(dinv), p(diag), ctypes.c_void_p(info.data_ptr()), None)
        dt = time.perf_counter() - t0
        _gpx.check(st, "dataflow bulk only")
        print("dataflow bulk-only: %d tiles in %.3f ms = %.3f us/tile (host-timed, incl. launch + sync)" % (tiles, dt * 1e3, dt * 1e6 / tiles), flush=True)
    cnt = torch.zeros(16, dtype=torch.int32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(4):
        cnt.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        # C[rows >= 16 tiles, columns >= 8 tiles] -= A[rows, 0:1024] B[...]^T : the same tiles as panel 0's trailing update
        st = L.gpx_dev_syrk_trap(p(A, 16 * 128 * n), n, p(A, 8 * 128 * n), n, p(A, 16 * 128 * n + 8 * 128), n, nt * 128, 1024, 1024, -1.0, 1.0,
                                 ctypes.c_void_p(cnt.data_ptr()), None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        _gpx.check(st, "syrk_trap")
        print("trapezoid launch:   %d tiles in %.3f ms = %.3f us/tile" % (tiles, dt * 1e3, dt * 1e6 / tiles), flush=True)


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "bulk":
        return bulk()
    if mode == "check":
        check()
    elif mode == "fit":
        fit_once(sys.argv[2])
    else:
        for frm in sys.argv[2:] or ["off", "whole"]:
            if "=" in frm:      # a variant given as environment assignments: A=1,B=2
                env = dict(os.environ, GPX_DFLOW_MAX_BLOCKS="0")
                env.update(dict(kv.split("=", 1) for kv in frm.split(",")))
            else:
                env = dict(os.environ, GPX_DFLOW_MAX_BLOCKS="100000" if frm == "whole" else "0")
            subprocess.run([sys.executable, os.path.abspath(__file__), "fit", frm], env=env, timeout=300)
        base = np.load(os.path.join(ROOT, "gpurun_out", "probe_dflow_beta_off.npy"))
        for frm in sys.argv[2:] or ["whole"]:
            f = os.path.join(ROOT, "gpurun_out", "probe_dflow_beta_%s.npy" % frm)
            if os.path.exists(f) and frm != "off":
                print("alpha vs off (%s): max abs diff %.3e (scale %.3e)" % (frm, np.abs(np.load(f) - base).max(), np.abs(base).max()))


if __name__ == "__main__":
    main()
