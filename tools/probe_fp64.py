"""Re-verify the fp64 roofline denominators on the box: MFMA / VALU issue rates, held clock, HBM."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scikit-gpuppy_amd"))
from skgpuppy_amd import _gpx
lib = _gpx.lib
for blocks in (256, 512, 1024, 2048):
    for mode, name in ((0, "mfma"), (1, "valu"), (2, "mixed")):
        tf, cy, ck = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _gpx.check(lib.gpx_bench_fp64_pipes(blocks, 20000, mode, ctypes.byref(tf), ctypes.byref(cy), ctypes.byref(ck)), "pipes")
        print("blocks=%5d %-5s %7.2f TFLOP/s  %6.1f cyc/inst(per wave)  clock %.3f GHz" % (blocks, name, tf.value, cy.value, ck.value))
w, c = ctypes.c_double(), ctypes.c_double()
_gpx.check(lib.gpx_bench_hbm(4 << 30, 5, ctypes.byref(w), ctypes.byref(c)), "hbm")
print("HBM fill %.0f GB/s, copy (r+w) %.0f GB/s" % (w.value, c.value))
