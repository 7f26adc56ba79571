"""Does the fp64 MFMA GEMM rate hold under sustained load?  Same SYRK launch repeated for ~2 s, rate per 50-launch window."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch
from skgpuppy_amd import _gpx
lib = _gpx.lib
p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 8 * off)
N = 16384
Lm = torch.randn(N, N, dtype=torch.float64, device="cuda")
T, K = 14336, 1024
r0 = N - T
A = p(Lm, r0 * N + (r0 - K)); C = p(Lm, r0 * N + r0)
tiles = (T // 128) * (T // 128 + 1) // 2
evs = [torch.cuda.Event(enable_timing=True) for _ in range(14)]
evs[0].record()
for w in range(13):
    for _ in range(50):
        lib.gpx_dev_gemm_nt(A, N, A, N, C, N, T, T, K, -1.0e-9, 1.0, 1, None)
    evs[w + 1].record()
torch.cuda.synchronize()
for w in range(13):
    ms = evs[w].elapsed_time(evs[w + 1]) / 50
    print("window %2d: %.3f ms per SYRK  %.1f TFLOP/s" % (w, ms, tiles * 2.0 * 128 * 128 * K / ms / 1e9))
