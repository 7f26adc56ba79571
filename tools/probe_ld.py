"""Does a leading dimension that is a power of two hurt?  The dominant GEMM on K = 1024 column slices of [16384, ld] matrices (the
factorisation's operands: panels of L) and an HBM-bound row sweep, for ld = 16384 and padded values."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch
from skgpuppy_amd import _gpx
lib = _gpx.lib
p = lambda t: ctypes.c_void_p(t.data_ptr())
M = N = 16384
K = 1024
for ld in (16384, 16384 + 16, 16384 + 32, 16384 + 64, 16384 + 512):
    big = torch.randn(M, ld, dtype=torch.float64, device="cuda")
    c = torch.zeros(M, ld, dtype=torch.float64, device="cuda")
    a = big[:, 2048:2048 + K]                      # a panel: K columns of every row, rows ld apart
    def run():
        _gpx.check(lib.gpx_dev_gemm_nt(p(a), ld, p(a), ld, p(c), ld, M, N, K, -1.0, 1.0, 1, None), "gemm")
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 6
    tiles = (M // 128) * (M // 128 + 1) / 2
    # HBM-bound: row sums of a 16-row-per-instruction sweep stand-in = torch's sum over dim 1 of the strided view
    e0.record()
    for _ in range(6):
        s_ = big[:, :16384].sum(dim=1)
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / 6
    print("ld=%6d  lower SYRK K=1024: %7.3f ms  %6.2f TFLOP/s   row sums of 2.1 GB: %6.3f ms = %5.2f TB/s" % (ld, ms, tiles * 2 * 128 * 128 * K / ms / 1e9, ms2, M * 16384 * 8 / ms2 / 1e9), flush=True)
