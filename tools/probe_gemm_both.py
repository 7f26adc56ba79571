"""libgpx GEMM and vendor DGEMM back to back on one shape (for rocprofv3 --pmc comparisons)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch
from skgpuppy_amd import _gpx
M = N = K = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = lambda t: ctypes.c_void_p(t.data_ptr())
a = torch.randn(M, K, dtype=torch.float64, device="cuda")
b = torch.randn(N, K, dtype=torch.float64, device="cuda")
c = torch.zeros(M, N, dtype=torch.float64, device="cuda")
for _ in range(3):
    _gpx.check(_gpx.lib.gpx_dev_gemm_nt(p(a), K, p(b), K, p(c), N, M, N, K, -1.0, 1.0, 0, None), "gemm")
    torch.addmm(c, a, b.T, beta=1.0, alpha=-1.0, out=c)
torch.cuda.synchronize()
