"""CPU-only: the C-ABI library loads, exports every symbol include/gpx.h declares, and fails loudly
(no CPU fallback) when no GPU is visible."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, have_gpu

import skgpuppy_amd
from skgpuppy_amd import _gpx


def _header_functions():
    text = open(os.path.join(ROOT, "include", "gpx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _header_functions()
    assert len(names) >= 25
    for name in names:
        assert hasattr(_gpx.lib, name), "libgpx.so does not export %s" % name
        assert name in _gpx.SIGNATURES, "ctypes binding lacks %s" % name
    # and nothing is bound that the header does not declare
    assert sorted(_gpx.SIGNATURES) == names


def test_abi_version():
    assert _gpx.lib.gpx_abi_version() == 1


def test_reference_surface_present():
    # names / signatures the reference exposes
    cov = skgpuppy_amd.GaussianCovariance()
    for m in ("__call__", "get_theta", "cov_matrix", "cov_matrix_ij", "inv_cov_matrix", "get_Jacobian", "get_Hessian"):
        assert callable(getattr(cov, m))
    for m in ("estimate_many", "estimate", "__call__", "get_realisation", "_get_beta", "_get_W_inv", "_get_v", "_get_vt",
              "_covariance", "_inv_cov_matrix", "_get_mean_t", "_get_Hessian", "_get_Jacobian"):
        assert hasattr(skgpuppy_amd.GaussianProcess, m)
    for m in ("propagate_GA", "propagate_mean", "_getFactor", "_get_variance_dv_h"):
        assert hasattr(skgpuppy_amd.UncertaintyPropagationApprox, m)
    for m in ("propagate_GA", "propagate_mean"):
        assert hasattr(skgpuppy_amd.UncertaintyPropagationExact, m)


def test_scalar_accessors_host_side():
    cov = skgpuppy_amd.GaussianCovariance()
    th = np.log(np.array([2.0, 0.01, 0.04, 0.04]))
    a = np.array([5.0, 5.0])
    assert cov(a, a.copy(), th) == pytest.approx(2.01, abs=1e-15)
    assert cov(a, a + 1.0, th) == pytest.approx(2.0 * np.exp(-0.04), rel=1e-15)
    J = cov.get_Jacobian(a, a + 1.0, th)
    H = cov.get_Hessian(a, a + 1.0, th)
    assert J.shape == (2, 1) and H.shape == (2, 2)
    np.testing.assert_allclose(H, H.T)


@pytest.mark.skipif(have_gpu(), reason="checks the no-device behaviour")
def test_no_device_is_an_error_not_a_fallback():
    cov = skgpuppy_amd.GaussianCovariance()
    x = np.random.RandomState(0).rand(5, 2)
    th = np.zeros(4)
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        cov.cov_matrix_ij(x, x, th)
    with pytest.raises(RuntimeError):
        skgpuppy_amd.GaussianProcess(x, np.zeros(5), cov, th)
    out = ctypes.c_double()
    assert _gpx.lib.gpx_bench_mfma_f64(10, ctypes.byref(out)) == _gpx.GPX_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "scikit-gpuppy_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
