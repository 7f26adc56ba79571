"""libgpx GEMM on the shapes of the Cholesky trailing update (K = 1024): full vs lower-only, operands strided like the panel
inside L (lda = N), plus the same square GEMMs at K = 8192 for reference."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch
from skgpuppy_amd import _gpx
lib = _gpx.lib
p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 8 * off)
N = 16384
Lm = torch.randn(N, N, dtype=torch.float64, device="cuda")
for T in (14336, 8192, 4096, 2048):
    for K in (1024, 2048):
        for lower in (0, 1):
            r0 = N - T                      # trailing rows, panel = columns [r0-K, r0)
            A = p(Lm, r0 * N + (r0 - K))
            C = p(Lm, r0 * N + r0)
            def run():
                _gpx.check(lib.gpx_dev_gemm_nt(A, N, A, N, C, N, T, T, K, -1.0e-9, 1.0, lower, None), "gemm")
            run(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record()
            for _ in range(reps):
                run()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            tiles = (T // 128) * (T // 128 + 1) // 2 if lower else (T // 128) ** 2
            print("T=%5d K=%4d lower=%d  %8.3f ms  %6.2f TFLOP/s (algorithmic tiles %d)" % (T, K, lower, ms, tiles * 2.0 * 128 * 128 * K / ms / 1e9, tiles))
