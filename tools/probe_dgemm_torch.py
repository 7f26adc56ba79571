"""Known-good reference on the same hardware: what does the vendor DGEMM (rocBLAS via torch) reach?"""
import time, torch
for n in (4096, 8192, 16384):
    a = torch.randn(n, n, dtype=torch.float64, device="cuda")
    b = torch.randn(n, n, dtype=torch.float64, device="cuda")
    c = a @ b.T
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        c = a @ b.T
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("torch fp64 NT gemm n=%d: %.2f ms  %.2f TFLOP/s" % (n, ms, 2.0 * n ** 3 / ms / 1e9))
    L = torch.linalg.cholesky(a @ a.T + n * torch.eye(n, dtype=torch.float64, device="cuda"))
    torch.cuda.synchronize()
    spd = a @ a.T + n * torch.eye(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    L = torch.linalg.cholesky(spd)
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("torch fp64 cholesky n=%d: %.2f ms  %.2f TFLOP/s" % (n, t * 1e3, n ** 3 / 3 / t / 1e12))
