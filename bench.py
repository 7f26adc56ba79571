#!/usr/bin/env python3
"""bench.py -- headline benchmark: GP fit + estimate_many points/second on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4]

`--gpus N` with N > 1 may be started either under torch.distributed.run (one rank per GPU; RANK / WORLD_SIZE in the
environment) or as a plain script: it then launches torch.distributed.run itself as a child process and forwards rank 0's line.
`--gpus 1 --workload c4` is the one-GPU point of the C4 scaling curve (same workload as the N > 1 lines).

One "step" = one pass of the hot path over one batch of synthetic input:
fit (Gram + vt I -> blocked fp64 Cholesky -> alpha) followed by estimate_many on M = N queries.
Inputs (x, t, xs) are resident in HBM before the timed region; outputs stay on the device.
value = (N + M) * steps / time   [pts/s], fp64 throughout.

N=1 : BASELINE.json config C3 (N=16384, d=8, M=16384), the configuration the metric is quoted on.
N>1 : one rank per GPU (torch.distributed over RCCL), config C4 (N=65536, d=16) with the K panels
      sharded block-cyclically across ranks (strong scaling) -- see DESIGN.md "multi-GPU".

The JSON line also carries
  roofline     : the dominant kernel (fp64 MFMA GEMM) timed live with HIP events on the handle's stream
  cpu_baseline : the oracle (CPU restatement of the reference algorithm, "port") timed on the host
                 cores on the FULL workload when that fits --cpu-budget seconds (rank 0, N=1 only), and
  parity_vs_oracle : the GPU outputs of the timed workload compared with the oracle's.
  python_api   : the same workload through the user-facing classes, host arrays in and out.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "scikit-gpuppy_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def requested_gpus(argv):
    """--gpus N from a raw argument list (no other parsing: runs before any heavy import)"""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    return n


def self_launch(argv, ngpus, target=None, port=None):
    """`bench.py --gpus N` (N > 1) started as a plain script: run one rank per GPU as a FRESH child process
    (python -m torch.distributed.run ... bench.py <the same arguments>), pass rank 0's JSON line on to stdout and
    return the child's exit code.  This process never touches the GPU and nothing is exec'ed over a process that did."""
    import subprocess
    target = target or os.environ.get("GPX_BENCH_LAUNCH_TARGET") or os.path.abspath(__file__)
    port = port or os.environ.get("GPX_BENCH_MASTER_PORT", "29511")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), target] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    last = None
    for line in proc.stdout:
        line = line.rstrip("\n")
        if line.startswith("{") and line.endswith("}"):
            last = line                      # the contract: ONE JSON line on stdout (rank 0's)
        elif line:
            sys.stderr.write(line + "\n")   # anything else a rank printed
    rc = proc.wait()
    if last is not None:
        print(last)
        sys.stdout.flush()
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank child run exited with code %d\n" % (ngpus, rc))
    return rc if rc != 0 or last is not None else 1


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and not os.environ.get("GPX_BENCH_SHARDED"):
    _n = requested_gpus(sys.argv[1:])
    if _n > 1:
        raise SystemExit(self_launch(sys.argv[1:], _n))

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before libgpx: one shared HIP runtime)

FP64_MFMA_PEAK_TFLOPS = 78.6     # MI355X datasheet fp64 matrix peak; re-measured below
HBM_PEAK_GBS = 8000.0

WORKLOADS = {
    "c2": dict(N=4096, d=4, M=4096),
    "c3": dict(N=16384, d=8, M=16384),
    "c4": dict(N=65536, d=16, M=65536),
}


def recipe(N, d, M):
    """BASELINE.md section 3 synthetic inputs."""
    rng = np.random.RandomState(20240 + N + d)
    x = rng.uniform(0, 10, (N, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
    xs = rng.uniform(0, 10, (M, d))
    theta = np.log(np.array([2.0, 0.01] + [0.04] * d))
    return x, t, xs, theta


def host_info():
    """CPU model, visible cores and the BLAS thread count the oracle's numpy/scipy calls use."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count()
    blas = []
    try:
        from threadpoolctl import threadpool_info
        blas = [{"lib": i.get("internal_api"), "threads": i.get("num_threads")} for i in threadpool_info()]
    except Exception:
        pass
    threads = max([b["threads"] for b in blas if b["lib"] in ("openblas", "mkl", "blis")] or [affinity])
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "sched_affinity": affinity, "blas": blas, "blas_threads": threads}


def cpu_baseline(N, d, M, budget_s=300.0):
    """Time the oracle (same algorithm class as the reference: tile/GEMM Gram, LU inverse, GEMM-based estimate_many with
    the M x M products) on the host cores.  The FULL workload is run when two probes (2048 and 4096 points, power-law fit)
    predict that it fits `budget_s`; otherwise the largest multiple of 1024 that does.  Returns sizes, stage times, and the oracle's outputs."""
    from oracle import oracle as orc
    xw, tw, xsw, thw = recipe(512, d, 512)
    orc.OracleGP(xw, tw, thw).estimate_many(xsw)          # imports, BLAS thread start-up
    # two probes fix the exponent of t(N) = a N^p on this host (BLAS efficiency and the memory-bound tile passes make it
    # 2.0 - 2.6 between 2048 and 16384, far from the cubic flop count: a cubic guess from one small probe overshoots 8x)
    probes = {}
    for n_ in (2048, 4096):
        xp, tp_, xsp, thp = recipe(n_, d, n_)
        t0 = time.perf_counter()
        orc.OracleGP(xp, tp_, thp).estimate_many(xsp)
        probes[n_] = time.perf_counter() - t0
    probe = probes[2048]
    p_exp = min(3.0, max(2.0, np.log2(probes[4096] / probes[2048])))
    predict_s = lambda n_: 1.25 * probes[4096] * (n_ / 4096.0) ** p_exp     # 25 % margin
    Ns = N
    while Ns > 4096 and predict_s(Ns) > budget_s:
        Ns -= 1024
    Ms = M if Ns == N else Ns
    x, t, xs, theta = recipe(Ns, d, Ms) if Ns != N else recipe(N, d, M)
    t0 = time.perf_counter()
    gp = orc.OracleGP(x, t, theta)
    t1 = time.perf_counter()
    mean, var = gp.estimate_many(xs)
    t2 = time.perf_counter()
    return {"N": Ns, "M": Ms, "fit_s": t1 - t0, "predict_s": t2 - t1, "probe_2048_s": probe, "probe_4096_s": probes[4096],
            "probe_exponent": float(p_exp), "mean": mean, "var": var,
            "inputs": (x, t, xs, theta), "oracle_gp": gp}


def python_api_section(N, d, M, reps=2):
    """Second figure (predict outputs copied to the host): the user-facing classes with NumPy inputs and NumPy
    outputs -- GaussianProcess(x, t, GaussianCovariance(), theta) + estimate_many(xs) -- PCIe copies included."""
    import skgpuppy_amd as sk
    x, t, xs, theta = recipe(N, d, M)
    best = None
    for _ in range(reps + 1):     # first repetition warms the allocator
        a = time.perf_counter()
        gp = sk.GaussianProcess(x, t, sk.GaussianCovariance(), theta.copy())
        b = time.perf_counter()
        mean, var = gp.estimate_many(xs)
        c = time.perf_counter()
        gp._dev().close()
        if best is None or c - a < best[0]:
            best = (c - a, b - a, c - b)
    return {"value": (N + M) / best[0], "unit": "pts/s", "fit_ms": best[1] * 1e3, "estimate_many_ms": best[2] * 1e3,
            "note": "GaussianProcess(...) + estimate_many(...) through the Python classes, host arrays in, host arrays out"}, mean, var


def propagate_section(lib, _gpx, vp, xd, td, th, N, d):
    """Outside the timed region: one propagate_GA (Approx and Exact) on a fitted handle (config C3 of BASELINE.json:
    u = 5*1_d, Sigma = 0.01 I), with the HBM-bound kernels timed by HIP events on the handle's stream."""
    u = np.full(d, 5.0)
    S = 0.01 * np.eye(d)
    o = [ctypes.c_double() for _ in range(4)]
    res = {}
    m, v = ctypes.c_double(), ctypes.c_double()
    # one untimed fit -> propagate -> free cycle first, like the warm-up steps of the headline figure: the library's caching
    # allocator then serves the propagation's buffers (a first-ever hipMalloc of them costs more than the propagation)
    h = ctypes.c_void_p()
    _gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "gpx_fit")
    _gpx.check(lib.gpx_propagate_approx(h, _gpx.ptr(u), _gpx.ptr(S), *[ctypes.byref(x_) for x_ in o]), "approx")
    lib.gpx_free(h)
    h = ctypes.c_void_p()
    _gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "gpx_fit")
    lib.gpx_profile_enable(h, 1)
    lib.gpx_profile_reset(h)
    call = lambda uu, SS: _gpx.check(lib.gpx_propagate_approx(h, _gpx.ptr(uu), _gpx.ptr(SS), *[ctypes.byref(x_) for x_ in o]), "approx")
    t0 = time.perf_counter()
    call(u, S)                       # right after the fit: two triangular solves on the right-hand-side block, no K^-1
    t1 = time.perf_counter()
    _gpx.check(lib.gpx_propagate_exact(h, _gpx.ptr(u), _gpx.ptr(S), ctypes.byref(m), ctypes.byref(v)), "exact")   # builds K^-1 (once per fit)
    t2 = time.perf_counter()
    call(u + 0.25, S)                # K^-1 resident: one pass over it per new u
    t3 = time.perf_counter()
    call(u + 0.25, 2 * S)            # same u, new Sigma: dot products only
    t4 = time.perf_counter()
    _gpx.check(lib.gpx_propagate_exact(h, _gpx.ptr(u + 0.25), _gpx.ptr(S), ctypes.byref(m), ctypes.byref(v)), "exact")
    t5 = time.perf_counter()
    res["approx_first_call_after_fit_ms"] = (t1 - t0) * 1e3
    res["exact_first_call_incl_Kinv_build_ms"] = (t2 - t1) * 1e3
    res["approx_new_u_ms"] = (t3 - t2) * 1e3
    res["approx_same_u_new_Sigma_ms"] = (t4 - t3) * 1e3
    res["exact_ms"] = (t5 - t4) * 1e3
    res["approx_calls_per_s"] = 1e3 / res["approx_new_u_ms"]
    res["exact_calls_per_s"] = 1e3 / res["exact_ms"]
    for cls, name in ((_gpx.K_QUAD, "approx_kinv_pass"), (_gpx.K_EXACT, "exact_sum")):
        n_, ms_, w_ = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
        lib.gpx_profile_read(h, cls, ctypes.byref(n_), ctypes.byref(ms_), ctypes.byref(w_))
        if n_.value:
            per = ms_.value / n_.value
            nbytes = 8.0 * N * N if cls == _gpx.K_QUAD else 4.0 * N * N      # exact_sum visits the j <= i half only
            res[name] = {"launches": n_.value, "avg_ms": per, "kinv_bytes": nbytes,
                         "hbm_GBs": nbytes / (per * 1e-3) / 1e9, "frac_of_8TBs": nbytes / (per * 1e-3) / 1e9 / HBM_PEAK_GBS}
    lib.gpx_free(h)
    return res


def likelihood_section(N, d):
    """"next" row f1 at the benchmark size, outside the timed region: one evaluation of the hyper-parameter likelihood and of its
    gradient at a new theta through the Python classes (what one L-BFGS-B iteration of GaussianCovariance.ml_estimate costs):
    fit + log det, then K^-1 and the fused gradient pass on the same device model."""
    import skgpuppy_amd as sk
    x, t, _xs, theta = recipe(N, d, 16)
    tc = t - t.mean()
    cov = sk.GaussianCovariance()
    best = None
    for rep in range(3):
        th = theta + 0.01 * (rep + 1)
        a = time.perf_counter()
        val = cov._negativeloglikelihood(x, tc, th)
        b = time.perf_counter()
        g = cov._d_nll_d_theta(x, tc, th)
        c = time.perf_counter()
        if rep and (best is None or c - a < best[0]):
            best = (c - a, b - a, c - b, val, float(np.abs(g).max()))
    cached = getattr(cov, "_ml_cache", None)
    if cached is not None:
        cached[1].close()
    return {"workload": "N=%d d=%d: _negativeloglikelihood + _d_nll_d_theta at a new theta (host arrays in)" % (N, d),
            "nll_ms": best[1] * 1e3, "gradient_ms": best[2] * 1e3, "iteration_ms": best[0] * 1e3, "nll": best[3], "max_abs_gradient": best[4],
            "algorithmic_flops": N ** 3 / 3.0 + 2.0 * N ** 3 / 3.0,       # Cholesky + K^-1 from the factor
            "tflops": (N ** 3 / 3.0 + 2.0 * N ** 3 / 3.0) / best[0] / 1e12}


def spgp_section(n=262144, m=2048, d=8, queries=16384, reps=2):
    """"next" row f3 at BASELINE config 5 (Snelson SPGP, M = 2048 pseudo-inputs, N = 262144, d = 8), outside the timed region:
    low-rank fit, estimate_many, Snelson's likelihood and its analytic gradient on one GPU (the same figures tools/bench_spgp.py prints)."""
    import skgpuppy_amd as sk
    rng = np.random.RandomState(20240 + n + d)
    x = rng.uniform(0, 10, (n, d))
    t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(n)
    xs = rng.uniform(0, 10, (queries, d))
    xb = x[rng.choice(n, m, replace=False)].copy()
    theta = np.concatenate([np.log(np.array([2.0, 0.01] + [0.04] * d)), xb.ravel()])
    cov = sk.SPGPCovariance(m)
    best = None
    for r in range(reps + 1):
        t0 = time.perf_counter()
        gp = sk.GaussianProcess(x, t, cov, theta)
        t1 = time.perf_counter()
        mu, _var = gp.estimate_many(xs)
        t2 = time.perf_counter()
        val = gp._dev().nll()             # on its own (estimate_many overwrote whatever a likelihood and a gradient share)
        t3 = time.perf_counter()
        g = gp._dev().nll_grad()          # right behind the likelihood at the same theta (an L-BFGS step): their common N m^2 part is there
        t4 = time.perf_counter()
        gp.estimate_many(xs[:16])         # (discards it again)
        t5 = time.perf_counter()
        g = gp._dev().nll_grad()          # on its own: everything it needs computed in the call
        t6 = time.perf_counter()
        gp._dev().close()
        cur = (t1 - t0, t2 - t1, t3 - t2, t6 - t5, t4 - t3)
        if r:
            best = cur if best is None else tuple(min(a_, b_) for a_, b_ in zip(best, cur))
    flops_fit = 2.0 * n * m * m + 2.0 * m ** 3 / 3.0          # TRSM + lower-only W^T W + two Cholesky
    return {"workload": "C5: SPGP N=%d M=%d d=%d, %d queries" % (n, m, d, queries), "fit_ms": best[0] * 1e3, "estimate_many_ms": best[1] * 1e3,
            "snelson_nll_ms": best[2] * 1e3, "analytic_gradient_ms": best[3] * 1e3, "gradient_after_nll_ms": best[4] * 1e3, "train_pts_per_s": n / best[0],
            "fit_tflops_algorithmic": flops_fit / best[0] / 1e12, "fit_frac_of_peak": flops_fit / best[0] / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "nll": val, "gradient_finite": bool(np.all(np.isfinite(g))),
            "mean_abs_residual": float(np.abs(mu - np.sin(0.3 * xs.sum(1))).mean())}


def workload_point(lib, _gpx, name, warm, reps):
    """Another BASELINE.json configuration on the same GPU, outside the timed region (the driver's record then holds a throughput
    figure for every single-GPU configuration): fit + estimate_many through the C-ABI on device-resident inputs, like the headline
    step; best of `reps` after `warm` warm-ups."""
    wl = WORKLOADS[name]
    N, d, M = wl["N"], wl["d"], wl["M"]
    x, t, xs, theta = recipe(N, d, M)
    dev = torch.device("cuda:0")
    xd, td, xsd = torch.as_tensor(x).to(dev), torch.as_tensor(t - t.mean()).to(dev), torch.as_tensor(xs).to(dev)
    mean_d = torch.empty(M, dtype=torch.float64, device=dev)
    var_d = torch.empty(M, dtype=torch.float64, device=dev)
    th = np.ascontiguousarray(theta)
    vp = lambda tt: ctypes.c_void_p(tt.data_ptr())  # noqa: E731
    best = None
    for r in range(warm + reps):
        h = ctypes.c_void_p()
        torch.cuda.synchronize()
        a = time.perf_counter()
        _gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "gpx_fit")
        b = time.perf_counter()
        _gpx.check(lib.gpx_predict(h, vp(xsd), M, vp(mean_d), vp(var_d)), "gpx_predict")
        torch.cuda.synchronize()
        c = time.perf_counter()
        lib.gpx_free(h)
        if r >= warm and (best is None or c - a < best[0]):
            best = (c - a, b - a, c - b)
    finite = bool(torch.isfinite(mean_d).all().item() and torch.isfinite(var_d).all().item() and (var_d > 0).all().item())
    del xd, td, xsd, mean_d, var_d
    flops = N ** 3 / 3.0 + float(N) * N * M
    return {"workload": "%s: N=%d d=%d M=%d fit + estimate_many, device-resident inputs" % (name.upper(), N, d, M),
            "warmup": warm, "best_of": reps, "fit_ms": best[1] * 1e3, "estimate_many_ms": best[2] * 1e3, "ms_per_step": best[0] * 1e3,
            "pts_per_s": (N + M) / best[0], "algorithmic_flops": flops, "tflops": flops / best[0] / 1e12,
            "frac_of_peak": flops / best[0] / 1e12 / FP64_MFMA_PEAK_TFLOPS, "fit_frac_of_peak": N ** 3 / 3.0 / best[1] / 1e12 / FP64_MFMA_PEAK_TFLOPS,
            "outputs_finite_and_var_positive": finite}


def multi_abi_point(lib, _gpx, name="c3", ranks=(1, 2), reps=3):
    """The single-process multi-device C-ABI (gpx_multi_*, csrc/multi.hip: the sharded schedule of SURVEY 8e without Python's launcher) on
    the ONE GPU of this run, with 1 and 2 logical ranks on it (a device ordinal may repeat): a driver-witnessed functional figure at full
    size -- ranks on one GPU share its CUs, so two cannot be faster here -- and the check that alpha equals gpx_fit's to the bit."""
    wl = WORKLOADS[name]
    N, d = wl["N"], wl["d"]
    x, t, xs, theta = recipe(N, d, 4096)
    x, tc, xs, th = np.ascontiguousarray(x), np.ascontiguousarray(t - t.mean()), np.ascontiguousarray(xs), np.ascontiguousarray(theta)
    h = ctypes.c_void_p()
    _gpx.check(lib.gpx_fit(_gpx.ptr(x), _gpx.ptr(tc), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "gpx_fit")
    ref = np.empty(N)
    _gpx.check(lib.gpx_alpha(h, _gpx.ptr(ref)), "gpx_alpha")
    lib.gpx_free(h)
    out = {"workload": "%s: N=%d d=%d through gpx_multi_fit / gpx_multi_predict (4096 queries), host arrays in, logical ranks on cuda:0" % (name.upper(), N, d)}
    for R in ranks:
        devs = (ctypes.c_int * R)(*([0] * R))
        best = None
        for _ in range(1 + reps):
            m = ctypes.c_void_p()
            a = time.perf_counter()
            _gpx.check(lib.gpx_multi_fit(_gpx.ptr(x), _gpx.ptr(tc), N, d, _gpx.ptr(th), devs, R, ctypes.byref(m)), "gpx_multi_fit")
            b = time.perf_counter()
            mean, var = np.empty(len(xs)), np.empty(len(xs))
            _gpx.check(lib.gpx_multi_predict(m, _gpx.ptr(xs), len(xs), _gpx.ptr(mean), _gpx.ptr(var)), "gpx_multi_predict")
            c = time.perf_counter()
            beta = np.empty(N)
            _gpx.check(lib.gpx_multi_alpha(m, _gpx.ptr(beta)), "gpx_multi_alpha")
            lib.gpx_multi_free(m)
            if best is None or b - a < best[0]:
                best = (b - a, c - b)
        out["ranks_%d" % R] = {"fit_ms": best[0] * 1e3, "predict_4096_ms": best[1] * 1e3, "max_abs_dalpha_vs_gpx_fit": float(np.abs(beta - ref).max()),
                               "outputs_finite_and_var_positive": bool(np.isfinite(mean).all() and (var > 0).all())}
    return out


def propagate_parity(cb, d):
    """Config C3's second half against the oracle at the benchmark size: propagate_GA (Approx and Exact, u = 5 1_d, Sigma = 0.01 I)
    on a GPU fit of the oracle's inputs against oracle.approx_propagate / exact_propagate on the oracle's own K^-1 (serial C
    loops, UncertaintyPropagation2.pyx order).  SURVEY 8a tolerances: mean 1e-9 absolute, variances 1e-8 v absolute."""
    import skgpuppy_amd as sk
    from oracle import oracle as orc
    xo, to_, _xso, tho = cb["inputs"]
    og = cb["oracle_gp"]
    u, S = np.full(d, 5.0), 0.01 * np.eye(d)
    t0 = time.perf_counter()
    oa = orc.approx_propagate(og, u, S)
    t1 = time.perf_counter()
    oe = orc.exact_propagate(og, u, S)
    t2 = time.perf_counter()
    gp = sk.GaussianProcess(xo, to_, sk.GaussianCovariance(), tho.copy())
    ga = sk.UncertaintyPropagationApprox(gp).propagate_GA(u, S)
    ge = sk.UncertaintyPropagationExact(gp).propagate_GA(u, S)
    gp._dev().close()
    v_ = float(np.exp(tho[0]))
    res = {"u": "5 * 1_d", "Sigma": "0.01 I", "oracle_approx_s": t1 - t0, "oracle_exact_s": t2 - t1,
           "approx": {"gpu": [float(ga[0]), float(ga[1])], "oracle": [float(oa[0]), float(oa[1])]},
           "exact": {"gpu": [float(ge[0]), float(ge[1])], "oracle": [float(oe[0]), float(oe[1])]},
           "tolerance": "mean 1e-9 abs, variance 1e-8 v abs"}
    res["ok"] = bool(abs(ga[0] - oa[0]) < 1e-9 and abs(ga[1] - oa[1]) < 1e-8 * v_ and abs(ge[0] - oe[0]) < 1e-9 and abs(ge[1] - oe[1]) < 1e-8 * v_)
    return res


def run_single(args):
    import skgpuppy_amd  # noqa: F401
    from skgpuppy_amd import _gpx

    wl = WORKLOADS[args.workload or "c3"]
    N, d, M = wl["N"], wl["d"], wl["M"]
    x, t, xs, theta = recipe(N, d, M)
    tc = t - t.mean()
    dev = torch.device("cuda:0")
    xd = torch.as_tensor(x).to(dev)
    td = torch.as_tensor(tc).to(dev)
    xsd = torch.as_tensor(xs).to(dev)
    mean_d = torch.empty(M, dtype=torch.float64, device=dev)
    var_d = torch.empty(M, dtype=torch.float64, device=dev)
    th = np.ascontiguousarray(theta)
    torch.cuda.synchronize()
    lib = _gpx.lib
    vp = lambda tt: ctypes.c_void_p(tt.data_ptr())  # noqa: E731

    NCLS = len(_gpx.KERNEL_CLASS_NAMES)
    prof = {k: [0, 0.0, 0.0] for k in range(NCLS)}
    t_fit = [0.0]
    t_pred = [0.0]

    def step(timed, clock=True):
        h = ctypes.c_void_p()
        a = time.perf_counter()
        _gpx.check(lib.gpx_fit(vp(xd), vp(td), N, d, _gpx.ptr(th), None, ctypes.byref(h)), "gpx_fit")
        b = time.perf_counter()
        if timed:
            lib.gpx_profile_enable(h, int(os.environ["GPX_PROFILE"]))
        _gpx.check(lib.gpx_predict(h, vp(xsd), M, vp(mean_d), vp(var_d)), "gpx_predict")
        c = time.perf_counter()
        if timed:
            if clock:
                t_fit[0] += b - a
                t_pred[0] += c - b
            for k in (range(NCLS) if os.environ["GPX_PROFILE"] == "2" else (_gpx.K_GEMM,)):
                n_, ms_, w_ = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
                lib.gpx_profile_read(h, k, ctypes.byref(n_), ctypes.byref(ms_), ctypes.byref(w_))
                prof[k][0] += n_.value
                prof[k][1] += ms_.value
                prof[k][2] += w_.value
        lib.gpx_free(h)

    # Timed region: only the dominant kernel (128x128-tile GEMM launches, ~106 per step) is bracketed with HIP events
    # on its stream (level 1; bracketing all ~1100 launches of a step costs ~9 % of the step).  The per-class table is
    # taken afterwards from ONE extra, untimed step at level 2.  The switch is process wide so gpx_fit is covered.
    os.environ["GPX_PROFILE"] = os.environ.get("GPX_BENCH_PROFILE", "1")
    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    value = (N + M) * args.steps / elapsed
    g_n, g_ms, g_flops = prof[_gpx.K_GEMM]
    table_prof = {k: list(v) for k, v in prof.items()}
    if os.environ["GPX_PROFILE"] != "2":
        for k in prof:
            prof[k] = [0, 0.0, 0.0]
        os.environ["GPX_PROFILE"] = "2"
        step(True, clock=False)
        torch.cuda.synchronize()
        table_prof, table_steps = prof, 1
    else:
        table_steps = args.steps
    # the sections below (propagation, Python classes, likelihood, SPGP, C2 / C4) are timed WITHOUT event bracketing: left at level 2
    # (all ~600 launches of a fit bracketed) the Python-API fit read 28.9 ms where it takes 27.1 (tools/probe_python_overhead.py)
    os.environ["GPX_PROFILE"] = "0"
    achieved = g_flops / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
    # known-good references on the same box: (i) the issue rate of v_mfma_f64_16x16x4_f64 itself (gpx_bench_mfma_f64: what `peak`
    # stands for, re-measured here), (ii) the vendor DGEMM (rocBLAS through torch) on an 8192^3 NT product -- best of 5 after 3
    # warm-ups (a single cold shot read 34 TFLOP/s in round 3's driver run against 74 warm)
    probe_tf = ctypes.c_double()
    mfma_probe = probe_tf.value if lib.gpx_bench_mfma_f64(4096, ctypes.byref(probe_tf)) == 0 else None
    a_ = torch.randn(8192, 8192, dtype=torch.float64, device=dev)
    for _ in range(3):
        c_ = a_ @ a_.T
    torch.cuda.synchronize()
    best_ms = 1e30
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        c_ = a_ @ a_.T
        e1.record()
        torch.cuda.synchronize()
        best_ms = min(best_ms, e0.elapsed_time(e1))
    vendor_tf = 2.0 * 8192 ** 3 / (best_ms * 1e-3) / 1e12
    del a_, c_

    traffic, traffic_src = None, None
    try:   # HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (same workload)
        pmc_name = "r05_pmc_traffic.json" if os.path.exists(os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")) else "r04_pmc_traffic.json"
        with open(os.path.join(ROOT, "profiles", pmc_name)) as fpm:
            pm = json.load(fpm)
        if (args.workload or "c3") == "c3" and g_n:
            # the counter passes serialise kernels, so the library runs its two-launch schedule there (55 dominant launches per step
            # instead of 44: the same tiles); the bytes of a whole step (2 steps in the passes) are spread over this run's launches
            per_step = pm["hbm_bytes_per_launch"] * pm["launches"] / 2.0
            traffic = per_step / (g_n / max(1, args.steps))
            traffic_src = ("profiles/" + pmc_name + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction): "
                           "%.1f GB per step over this run's %.0f dominant launches per step" % (per_step / 1e9, g_n / max(1, args.steps)))
    except Exception:
        pass
    out = {
        "metric": "GP fit+predict pts/sec (K+Cholesky, N=%d d=%d)" % (N, d),
        "value": value,
        "unit": "pts/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",     # the workload (total work) is fixed; N > 1 lines shard the SAME kind of step at config C4
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%s: N=%d d=%d M=%d fit(Gram+Cholesky+alpha)+estimate_many, theta fixed" % (
            (args.workload or "c3").upper(), N, d, M), "global_batch": N + M, "parallelism": "1 GPU"},
        "fit_ms": t_fit[0] / args.steps * 1e3,
        "predict_ms": t_pred[0] / args.steps * 1e3,
        # per-stage view: whole-stage algorithmic flops / stage wall time against the fp64 MFMA peak
        "stages": {
            "fit": {"pts_per_s": N * args.steps / t_fit[0], "algorithmic_flops": N ** 3 / 3.0,
                    "tflops": N ** 3 / 3.0 * args.steps / t_fit[0] / 1e12,
                    "frac_of_peak": N ** 3 / 3.0 * args.steps / t_fit[0] / 1e12 / FP64_MFMA_PEAK_TFLOPS},
            "estimate_many": {"pts_per_s": M * args.steps / t_pred[0], "algorithmic_flops": float(N) * N * M,
                              "tflops": float(N) * N * M * args.steps / t_pred[0] / 1e12,
                              "frac_of_peak": float(N) * N * M * args.steps / t_pred[0] / 1e12 / FP64_MFMA_PEAK_TFLOPS},
        },
        "roofline": {
            "kernel": "gemm_nt_f64_kernel (v_mfma_f64_16x16x4_f64)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": FP64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "launches_per_step": g_n / max(1, args.steps),
            "avg_launch_ms": g_ms / max(1, g_n),
            "flops_per_step": g_flops / max(1, args.steps),
            "mfma_probe_tflops": mfma_probe,           # back-to-back v_mfma_f64_16x16x4_f64 on this box (the denominator, re-measured)
            "vendor_dgemm_8192_tflops_same_box": vendor_tf,   # best of 5 after 3 warm-ups
        },
        "kernel_classes_from_untimed_profiling_step": {
            _gpx.KERNEL_CLASS_NAMES[k]: {"launches": table_prof[k][0] / table_steps, "ms": table_prof[k][1] / table_steps,
                                         "work": table_prof[k][2] / table_steps} for k in range(NCLS) if table_prof[k][0]},
    }
    if not args.no_propagate:
        out["propagate"] = propagate_section(lib, _gpx, vp, xd, td, th, N, d)
    if not args.no_python_api:
        out["python_api"], mean_py, var_py = python_api_section(N, d, M)
    if not args.no_extras:
        out["likelihood_f1"] = likelihood_section(N, d)
        if (args.workload or "c3") == "c3":
            _gpx.lib.gpx_pool_trim()
            out["spgp_c5_f3"] = spgp_section()
            _gpx.lib.gpx_pool_trim()
            # the other single-GPU configurations of BASELINE.json, so that every one of them has a driver-witnessed figure
            out["c2"] = workload_point(lib, _gpx, "c2", warm=3, reps=10)
            _gpx.lib.gpx_pool_trim()
            out["c4_one_gpu"] = workload_point(lib, _gpx, "c4", warm=1, reps=2)   # the one-GPU point of the N = 65536 scaling curve
            _gpx.lib.gpx_pool_trim()
            out["multi_abi"] = multi_abi_point(lib, _gpx)                        # e1-e4 behind the C-ABI, rehearsed with logical ranks on this GPU
            _gpx.lib.gpx_pool_trim()
    if not args.no_cpu:
        cb = cpu_baseline(N, d, M, budget_s=args.cpu_budget)
        Ns, Ms, tf, tp = cb["N"], cb["M"], cb["fit_s"], cb["predict_s"]
        hi = host_info()
        full = (Ns == N and Ms == M)
        out["cpu_baseline"] = {
            "value": (Ns + Ms) / (tf + tp),
            "unit": "pts/s",
            "cores": hi["blas_threads"],
            "kind": "port",
            "sample": ("the FULL workload (N=M=%d d=%d)" % (Ns, d) if full else
                       "N=M=%d d=%d of the same recipe (the full workload was predicted to exceed the %.0f s budget)" % (Ns, d, args.cpu_budget))
                      + ": oracle (numpy/scipy restatement of the reference algorithm: GEMM-expansion Gram, LU inverse, "
                        "GEMM estimate_many incl. the M x M products) fit %.2f s + estimate_many %.2f s" % (tf, tp),
            "full_workload": full,
            "fit_s": tf, "estimate_many_s": tp, "probe_2048_s": cb["probe_2048_s"], "probe_4096_s": cb["probe_4096_s"],
            "probe_exponent": cb["probe_exponent"],
            "host": hi,
        }
        if not full:
            scale = (N / Ns) ** cb["probe_exponent"]      # the exponent measured between the two probes, not the cubic flop count
            out["cpu_baseline"]["extrapolated_full_workload_value"] = (N + M) / ((tf + tp) * scale)
        # parity of what was just timed: the GPU path on the oracle's inputs against the oracle's outputs (SURVEY 8a tolerances)
        xo, to_, xso, tho = cb["inputs"]
        if full:
            gm, gv = mean_d.cpu().numpy() + to_.mean(), var_d.cpu().numpy()
        else:
            import skgpuppy_amd as sk
            gpo = sk.GaussianProcess(xo, to_, sk.GaussianCovariance(), tho.copy())
            gm, gv = gpo.estimate_many(xso)
            gpo._dev().close()
        v_ = float(np.exp(tho[0]))
        em, ev = float(np.abs(gm - cb["mean"]).max()), float(np.abs(gv - cb["var"]).max())
        ok = bool(np.allclose(gm, cb["mean"], rtol=1e-6, atol=1e-9 * v_) and np.allclose(gv, cb["var"], rtol=1e-6, atol=1e-9 * v_))
        out["parity_vs_oracle"] = {"workload": "N=M=%d d=%d" % (Ns, d), "max_abs_dmean": em, "max_abs_dvar": ev,
                                   "tolerance": "rtol 1e-6, atol 1e-9 v", "ok": ok}
        if not ok:
            print(json.dumps(out))
            raise SystemExit("bench.py: GPU outputs differ from the oracle beyond tolerance: %r" % (out["parity_vs_oracle"],))
        if not args.no_propagate:
            pp = propagate_parity(cb, d)
            out["parity_vs_oracle"]["propagate_GA"] = pp
            if not pp["ok"]:
                print(json.dumps(out))
                raise SystemExit("bench.py: propagate_GA differs from the oracle beyond tolerance: %r" % (pp,))
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg (and the parity check against it)")
    ap.add_argument("--cpu-budget", type=float, default=240.0, help="seconds the CPU baseline may take (full workload if it fits)")
    ap.add_argument("--no-python-api", action="store_true", help="skip the second figure through the Python classes")
    ap.add_argument("--no-propagate", action="store_true", help="skip the (untimed) propagate_GA section and its parity check against the oracle")
    ap.add_argument("--no-extras", action="store_true", help="skip the (untimed) likelihood (f1), SPGP config-5 (f3), C2 and one-GPU C4 figures")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not os.environ.get("GPX_BENCH_SHARDED"):
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))    # (imported and called as main(): the module-level launcher did not run)
    rehearsal = bool(os.environ.get("GPX_BENCH_SHARDED"))    # diagnostic: the sharded code path with ONE rank (world size 1)
    if args.gpus != world and not rehearsal:
        raise SystemExit("bench.py: --gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 or world > 1 or rehearsal:
        from skgpuppy_amd import distributed as dist_mod
        dist_mod.bench_main(args)
        return
    run_single(args)


if __name__ == "__main__":
    main()
