"""Time libgpx's fp64 MFMA GEMM on large shapes next to the vendor DGEMM (same box, same data)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch
from skgpuppy_amd import _gpx
lib = _gpx.lib
p = lambda t: ctypes.c_void_p(t.data_ptr())
for (M, N, K) in [(8192, 8192, 8192), (16384, 16384, 2048), (16384, 128, 128), (16384, 2048, 2048), (8192, 8192, 128), (4096, 4096, 4096)]:
    a = torch.randn(M, K, dtype=torch.float64, device="cuda")
    b = torch.randn(N, K, dtype=torch.float64, device="cuda")
    c = torch.zeros(M, N, dtype=torch.float64, device="cuda")
    for name in ("gpx", "torch"):
        def run():
            if name == "gpx":
                _gpx.check(lib.gpx_dev_gemm_nt(p(a), K, p(b), K, p(c), N, M, N, K, -1.0, 1.0, 0, None), "gemm")
            else:
                torch.addmm(c, a, b.T, beta=1.0, alpha=-1.0, out=c)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-5s M=%5d N=%5d K=%5d  %8.3f ms  %6.2f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
