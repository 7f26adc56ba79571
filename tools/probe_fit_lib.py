#!/usr/bin/env python3
"""gpx_fit / gpx_predict wall time at C3 size (or N given by PROBE_N) through a GIVEN libgpx.so (ctypes only: any library version with
gpx_fit / gpx_predict / gpx_free): A/B of builds and of environment switches on the same box, alternating.
usage: probe_fit_lib.py [ENV=V,ENV=V@]LIB [...]        (each variant runs in a fresh process, ROUNDS times in turn)"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np


def run(path):
    import torch
    lib = ctypes.CDLL(path)
    lib.gpx_fit.restype = ctypes.c_int
    lib.gpx_predict.restype = ctypes.c_int
    N, d = int(os.environ.get("PROBE_N", "16384")), int(os.environ.get("PROBE_D", "8"))
    rng = np.random.RandomState(20240 + N + d)
    x = rng.uniform(0, 10, (N, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
    xs = rng.uniform(0, 10, (N, d))
    theta = np.ascontiguousarray(np.log(np.array([2.0, 0.01] + [0.04] * d)))
    dev = torch.device("cuda")
    xd, td, xsd = torch.as_tensor(x).to(dev), torch.as_tensor(t - t.mean()).to(dev), torch.as_tensor(xs).to(dev)
    mean_d = torch.empty(N, dtype=torch.float64, device=dev)
    var_d = torch.empty(N, dtype=torch.float64, device=dev)
    vp = lambda tt: ctypes.c_void_p(tt.data_ptr())
    fits, preds = [], []
    for rep in range(int(os.environ.get("PROBE_REPS", "12"))):
        h = ctypes.c_void_p()
        torch.cuda.synchronize()
        a = time.perf_counter()
        st = lib.gpx_fit(vp(xd), vp(td), ctypes.c_int64(N), ctypes.c_int(d), ctypes.c_void_p(theta.ctypes.data), None, ctypes.byref(h))
        b = time.perf_counter()
        assert st == 0, st
        st = lib.gpx_predict(h, vp(xsd), ctypes.c_int64(N), vp(mean_d), vp(var_d))
        torch.cuda.synchronize()
        c = time.perf_counter()
        assert st == 0, st
        lib.gpx_free(h)
        fits.append(b - a)
        preds.append(c - b)
    f, p = sorted(fits[3:]), sorted(preds[3:])
    print("RESULT %.3f %.3f %.3f %.3f %.6e" % (f[0] * 1e3, f[len(f) // 2] * 1e3, p[0] * 1e3, p[len(p) // 2] * 1e3, float(mean_d.sum().item())), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 2 and "@" not in sys.argv[1]:
        run(sys.argv[1])
    else:
        variants = sys.argv[1:]
        res = {v: [] for v in variants}
        for rnd in range(int(os.environ.get("ROUNDS", "2"))):
            for v in variants:
                envs, path = v.split("@") if "@" in v else ("", v)
                env = dict(os.environ)
                for kv in filter(None, envs.split(",")):
                    k, val = kv.split("=", 1)
                    env[k] = val
                r = subprocess.run([sys.executable, __file__, path], env=env, capture_output=True, text=True, timeout=600)
                line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
                if r.returncode or not line:
                    print("%-70s FAILED %s" % (v[-70:], r.stderr[-300:]), flush=True)
                    continue
                res[v].append([float(z) for z in line[-1].split()[1:]])
        for v in variants:
            if res[v]:
                a = np.array(res[v])
                print("%-72s fit best %.2f med %.2f | predict best %.2f med %.2f | checksum %.6e" % (
                    v[-72:], a[:, 0].min(), np.median(a[:, 1]), a[:, 2].min(), np.median(a[:, 3]), a[-1, 4]), flush=True)
