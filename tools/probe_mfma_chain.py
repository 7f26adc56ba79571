"""MFMA f64 issue interval vs dependent-accumulator latency: vary independent accumulators and waves/SIMD."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
from skgpuppy_amd import _gpx
lib = _gpx.lib
for blocks in (256, 512, 1024):
    for k in range(6):
        tf, cy, ck = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _gpx.check(lib.gpx_bench_fp64_pipes(blocks, 4000, 10 + k, ctypes.byref(tf), ctypes.byref(cy), ctypes.byref(ck)), "pipes")
        print("blocks=%5d (%d waves/SIMD) nacc=%2d  %7.2f TFLOP/s  %7.1f cyc/MFMA per wave  clock %.3f GHz" % (blocks, blocks // 256, 1 << k, tf.value, cy.value, ck.value))
