"""The factorisation chain's small products alone on an idle chip, per launch (dependent launches on one stream, same C):
rank-128 update of a square (64 x 64 tiles, K = 128, beta = 1), in-place solve leaf (32 x 128 slabs, K = 128, beta = 0),
square update before a chain (32 x 32 lower tiles, K = 1024).  usage: probe_small_gemm.py [LIB]  -- A/B of builds on one box"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
lib = ctypes.CDLL(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "scikit-gpuppy_amd", "skgpuppy_amd", "libgpx.so"))
lib.gpx_dev_gemm_nt.restype = ctypes.c_int
lib.gpx_dev_gemm_nt.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
p = lambda t: t.data_ptr()
ld = 16384
L = torch.randn(4096, ld, dtype=torch.float64, device="cuda") * 1e-3
D = torch.randn(128, 128, dtype=torch.float64, device="cuda") * 1e-3
def bench(name, fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-64s %7.2f us per launch" % (name, e0.elapsed_time(e1) * 1e3 / reps), flush=True)
for rb in (7, 4, 2):   # block rows below the diagonal block inside the square
    Z = p(L) + 8 * (128 * ld)            # rows 128.., column block 0
    C = p(L) + 8 * (128 * ld + 128)
    bench("rank-128 update of %d x %d blocks (64x64 tiles, K=128, beta=1)" % (rb, rb),
          lambda: lib.gpx_dev_gemm_nt(Z, ld, Z, ld, C, ld, rb * 128, rb * 128, 128, -1.0, 1.0, 0, None))
    bench("in-place solve of %d blocks (32x128 slabs, K=128, beta=0)" % rb,
          lambda: lib.gpx_dev_gemm_nt(Z, ld, p(D), 128, Z, ld, rb * 128, 128, 128, 1.0, 0.0, 0, None))
P = p(L) + 8 * (1024 * ld)
bench("square update 1024 x 1024 lower, K=1024 (32x32 tiles)", lambda: lib.gpx_dev_gemm_nt(P, ld, P, ld, p(L) + 8 * (1024 * ld + 1024), ld, 1024, 1024, 1024, -1.0, 1.0, 1, None))
bench("column solve update 6144 rows x 128, K=512 (32x128 tiles)", lambda: lib.gpx_dev_gemm_nt(p(L), ld, P, ld, p(L) + 8 * 2048, ld, 3072, 128, 512, -1.0, 1.0, 0, None))
