"""Diagnostic build (-DGPX_GEMM_STAMP): shares of a GEMM stage spent in each segment."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GPX_LIB"] = os.path.join(ROOT, "scikit-gpuppy_amd", "skgpuppy_amd", "libgpx_stamp.so")
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch
from skgpuppy_amd import _gpx
lib = _gpx.lib
lib.gpx_stamp_read.argtypes = [ctypes.c_void_p]
M = N = K = 8192
p = lambda t: ctypes.c_void_p(t.data_ptr())
a = torch.randn(M, K, dtype=torch.float64, device="cuda"); b = torch.randn(N, K, dtype=torch.float64, device="cuda")
c = torch.zeros(M, N, dtype=torch.float64, device="cuda")
out = (ctypes.c_ulonglong * 5)()
for big in (0,):
    _gpx.check(lib.gpx_dev_gemm_nt(p(a), K, p(b), K, p(c), N, M, N, K, -1.0, 1.0, 0, None), "gemm"); torch.cuda.synchronize()
    lib.gpx_stamp_read(out)
    _gpx.check(lib.gpx_dev_gemm_nt(p(a), K, p(b), K, p(c), N, M, N, K, -1.0, 1.0, 0, None), "gemm"); torch.cuda.synchronize()
    lib.gpx_stamp_read(out)
    v = [out[i] for i in range(5)]
    tot = float(sum(v))
    names = ["barrier-end..stage top (phase3 MFMAs + loads issue)", "phases 0,1 (32 MFMA)", "LDS write of next stage", "phase 2 (16 MFMA)", "barrier wait"]
    waves = (M // 128) * (N // 128) * 4; stages = K // 16
    print("cycles per stage per wave: %.0f (ideal 64 MFMA x 64 = 4096 x 2 waves/SIMD = 8192)" % (tot / waves / stages))
    for n_, x in zip(names, v):
        print("  %-55s %5.1f%%  %.0f cyc" % (n_, 100 * x / tot, x / waves / stages))
