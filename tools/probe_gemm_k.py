"""libgpx fp64 GEMM (128 x 128 tiles) by contraction length, beta = 1 and beta = 0, next to the vendor DGEMM: what a tile pays per K."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch
from skgpuppy_amd import _gpx
lib = _gpx.lib
p = lambda t: ctypes.c_void_p(t.data_ptr())
M = N = 16384
for K in (256, 512, 1024, 2048, 4096, 8192):
    a = torch.randn(M, K, dtype=torch.float64, device="cuda")
    b = torch.randn(N, K, dtype=torch.float64, device="cuda")
    c = torch.zeros(M, N, dtype=torch.float64, device="cuda")
    for name, beta in (("gpx beta=1", 1.0), ("gpx beta=0", 0.0), ("vendor beta=1", 1.0)):
        def run():
            if name.startswith("gpx"):
                _gpx.check(lib.gpx_dev_gemm_nt(p(a), K, p(b), K, p(c), N, M, N, K, -1.0, beta, 0, None), "gemm")
            else:
                torch.addmm(c, a, b.T, beta=1.0, alpha=-1.0, out=c)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3 if K >= 4096 else 6
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        rounds = (M // 128) * (N // 128) / 512.0
        print("%-14s K=%5d  %8.3f ms  %6.2f TFLOP/s  %7.1f us per round of 512 tiles" % (name, K, ms, 2.0 * M * N * K / ms / 1e9, ms * 1e3 / rounds), flush=True)
