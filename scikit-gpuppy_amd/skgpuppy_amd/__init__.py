"""skgpuppy_amd -- MI355X-native drop-in for the GP hot path of scikit-gpuppy.

Same class names / signatures as the reference modules skgpuppy.Covariance, skgpuppy.GaussianProcess
and skgpuppy.UncertaintyPropagation; the arithmetic runs in hand-written HIP kernels (libgpx.so)
behind a ctypes C-ABI (include/gpx.h).  Importing this package without the built library fails.
"""
from . import _gpx  # noqa: F401  (fails loudly when libgpx.so is missing)
from .Covariance import Covariance, GaussianCovariance, SPGPCovariance, tracedot  # noqa: F401
from .GaussianProcess import GaussianProcess  # noqa: F401
from .UncertaintyPropagation import (  # noqa: F401
    UncertaintyPropagationApprox,
    UncertaintyPropagationExact,
    UncertaintyPropagationGA,
)

from .InverseUncertaintyPropagation import (  # noqa: F401,E402
    InverseUncertaintyPropagation,
    InverseUncertaintyPropagationApprox,
    InverseUncertaintyPropagationNumerical,
)

__all__ = [
    "Covariance", "GaussianCovariance", "SPGPCovariance", "GaussianProcess", "UncertaintyPropagationGA",
    "UncertaintyPropagationApprox", "UncertaintyPropagationExact", "tracedot",
    "InverseUncertaintyPropagation", "InverseUncertaintyPropagationApprox", "InverseUncertaintyPropagationNumerical",
]
