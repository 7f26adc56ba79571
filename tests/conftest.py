"""pytest configuration: markers, import paths, shared fixtures."""
import os
import sys

import numpy as np
import pytest

try:  # torch first: its bundled HIP runtime and libgpx share one libamdhip64 (same SONAME)
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_PARENT = os.path.join(ROOT, "scikit-gpuppy_amd")
for p in (ROOT, PKG_PARENT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


GP_CASES = ["kat1_grid", "grid_int", "n203_d3", "n256_d8", "n1000_d4", "n300_d16", "metis"]


def have_gpu():
    try:
        import skgpuppy_amd._gpx as g
        return g.device_count() > 0
    except Exception:
        return False
