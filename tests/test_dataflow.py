"""The factorisation as one persistent dataflow launch (csrc/dflow.hip, gpx_dev_chol_dataflow; the schedule of gpx_fit's Cholesky up to
GPX_DFLOW_MAX_BLOCKS block rows, default none): the factor against numpy on ragged block counts, the trailing-part form, bit-identity of a
fit that hands everything to the kernel, and the stall protocol (an expired in-kernel wait aborts the launch and the fit repeats on the plain
schedule).  Replaces skgpuppy/Covariance.py:179 like the default schedule does."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import torch

from skgpuppy_amd import _gpx

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spd(n, seed):
    rng = np.random.RandomState(seed)
    B = rng.randn(n, 48)
    return B.dot(B.T) / 48.0 + np.diag(rng.uniform(1.0, 2.0, n))


def _run(A, nb, first=0):
    dev = torch.device("cuda")
    n = 128 * nb
    Ad = torch.as_tensor(A).to(dev).contiguous()
    dinv = torch.zeros(nb * 128 * 128, dtype=torch.float64, device=dev)
    diag = torch.zeros(n, dtype=torch.float64, device=dev)
    info = torch.zeros(4, dtype=torch.int32, device=dev)
    st = _gpx.lib.gpx_dev_chol_dataflow(ctypes.c_void_p(Ad.data_ptr()), n, nb, first, ctypes.c_void_p(dinv.data_ptr()),
                                        ctypes.c_void_p(diag.data_ptr()), ctypes.c_void_p(info.data_ptr()), None)
    _gpx.check(st, "gpx_dev_chol_dataflow")
    return Ad.cpu().numpy(), dinv.cpu().numpy().reshape(nb, 128, 128), diag.cpu().numpy(), info.cpu().numpy()


@pytest.mark.parametrize("nb", [1, 7, 9, 17, 26])
def test_dataflow_factor_against_numpy(nb):
    """one panel, a ragged last panel, two and four panels (trailing tiles, square updates, column queues all in play)"""
    n = 128 * nb
    A = _spd(n, nb)
    ref = np.linalg.cholesky(A)
    got, dinv, diag, info = _run(A, nb)
    assert info[0] == 0 and info[1] == 0
    np.testing.assert_allclose(np.tril(got), ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    np.testing.assert_allclose(diag, np.diag(ref), rtol=1e-13)
    for k in range(nb):
        Lkk = ref[128 * k:128 * k + 128, 128 * k:128 * k + 128]
        np.testing.assert_allclose(dinv[k].dot(Lkk), np.eye(128), rtol=0, atol=1e-10)


def test_dataflow_trailing_part_and_not_positive_definite():
    """first_block > 0: the block columns before it are final and applied (here by numpy); a non-positive pivot comes back as the 1-based
    column in the status word, not as a stall"""
    nb, first = 19, 8
    n, c = 128 * nb, 128 * first
    A = _spd(n, 5)
    ref = np.linalg.cholesky(A)
    M = A.copy()
    M[:, :c] = ref[:, :c]                                             # the finished block columns
    M[c:, c:] = A[c:, c:] - ref[c:, :c].dot(ref[c:, :c].T)           # ... applied to the trailing matrix
    got, dinv, diag, info = _run(M, nb, first)
    assert info[0] == 0 and info[1] == 0
    np.testing.assert_allclose(np.tril(got)[c:, c:], ref[c:, c:], rtol=0, atol=1e-12 * np.abs(ref).max())
    np.testing.assert_array_equal(got[:, :c], M[:, :c])               # untouched
    bad = _spd(128 * 10, 6)
    bad[700, 700] = -3.0
    _g, _d, _dg, info = _run(bad, 10)
    assert info[1] == 0 and info[0] == 701


_FIT_WORKER = r"""
import sys, numpy as np
sys.path.insert(0, %(pkg)r)
import torch
import skgpuppy_amd as sk
rng = np.random.RandomState(9)
N, d = 9000, 5
x = rng.uniform(0, 10, (N, d)); t = np.sin(0.3 * x.sum(1)) + 0.1 * rng.randn(N)
theta = np.log(np.array([2.0, 0.01] + [0.04] * d))
gp = sk.GaussianProcess(x, t, sk.GaussianCovariance(), theta.copy())
np.save(sys.argv[1], gp._get_beta())
print("JITTER %%g" %% gp._dev().jitter())
"""


def test_fit_with_dataflow_schedule_is_bit_identical(tmp_path):
    """the fit's Cholesky handed to the dataflow kernel as a whole (GPX_DFLOW_MAX_BLOCKS above the matrix' 71 block rows; by default only
    no fit goes there): the same tile arithmetic in the same order as the multi-stream schedule -> alpha agrees to the
    last bit; and the stall protocol: with a time limit of 2 us every in-kernel wait expires, the launch aborts, the fit repeats on the
    plain schedule and still returns that alpha"""
    code = _FIT_WORKER % {"pkg": os.path.join(ROOT, "scikit-gpuppy_amd")}
    betas = {}
    variants = {"default": {"GPX_DFLOW_MAX_BLOCKS": "0"}, "whole": {"GPX_DFLOW_MAX_BLOCKS": "1000"},
                "stalled": {"GPX_DFLOW_MAX_BLOCKS": "1000", "GPX_WAIT_LIMIT_MS": "0.002", "GPX_DEBUG": "1"}}
    for name, extra in variants.items():
        env = dict(os.environ)
        env.update(extra)
        out = tmp_path / (name + ".npy")
        r = subprocess.run([sys.executable, "-c", code, str(out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        assert [l for l in r.stdout.splitlines() if l.startswith("JITTER")][-1].split()[1] == "0", r.stdout
        if name == "stalled":
            assert "hand-off stalled: refit on the plain schedule" in r.stderr, r.stderr[-2000:]
        betas[name] = np.load(out)
    for name in ("whole", "stalled"):
        np.testing.assert_array_equal(betas[name], betas["default"])
