"""GEMM throughput on CU-masked streams (hipExtStreamCreateWithCUMask): does excluding a few CUs cost more than its share?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
import torch
from skgpuppy_amd import _gpx
lib = _gpx.lib
hip = ctypes.CDLL("libamdhip64.so")
p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 8 * off)
N = 16384
Lm = torch.randn(N, N, dtype=torch.float64, device="cuda")
T, K = 14336, 1024
r0 = N - T
A = p(Lm, r0 * N + (r0 - K)); C = p(Lm, r0 * N + r0)
tiles = (T // 128) * (T // 128 + 1) // 2


def masked_stream(excl_bits):
    words = (ctypes.c_uint32 * 8)(*([0xFFFFFFFF] * 8))
    for b in excl_bits:
        words[b >> 5] &= ~(1 << (b & 31)) & 0xFFFFFFFF
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return st


def bench(st, label):
    for _ in range(3):
        lib.gpx_dev_gemm_nt(A, N, A, N, C, N, T, T, K, -1.0e-9, 1.0, 1, st)
    torch.cuda.synchronize(); hip.hipStreamSynchronize(st)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    hip.hipEventCreate(ctypes.byref(e0)); hip.hipEventCreate(ctypes.byref(e1))
    hip.hipEventRecord(e0, st)
    for _ in range(10):
        lib.gpx_dev_gemm_nt(A, N, A, N, C, N, T, T, K, -1.0e-9, 1.0, 1, st)
    hip.hipEventRecord(e1, st); hip.hipStreamSynchronize(st)
    ms = ctypes.c_float(); hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1)
    ms = ms.value / 10
    print("%-34s %.3f ms  %.1f TFLOP/s" % (label, ms, tiles * 2.0 * 128 * 128 * K / ms / 1e9))


plain = ctypes.c_void_p()
hip.hipStreamCreateWithFlags(ctypes.byref(plain), 1)
bench(plain, "unmasked stream")
bench(masked_stream([]), "mask = all 256 CUs")
bench(masked_stream(range(8)), "mask excludes bits 0-7")
bench(masked_stream(range(16)), "mask excludes bits 0-15")
bench(masked_stream(range(248, 256)), "mask excludes bits 248-255")
bench(masked_stream([0, 32, 64, 96, 128, 160, 192, 224]), "mask excludes bit 0 of each word")
bench(masked_stream(range(0, 32)), "mask excludes bits 0-31")
