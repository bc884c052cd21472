"""hyperslam_amd — MI355X-native continuous-time NLLS backend for HyperSLAM (hot path only).

The product is the HIP library ``libhyperslam_hip.so`` behind the C ABI of ``include/hyperslam_hip.h``; this package is the
thin host-side mirror used by tests, bench.py and the integration examples.
"""
from ._lib import HS_BEARING, HS_INERTIAL, HS_INERTIAL_AS_REFERENCE, HS_INERTIAL_EXACT, HS_PIXEL, HS_PRIOR, load  # noqa: F401
from ._lib import (HS_MANIFOLD_BIAS_POINT, HS_MANIFOLD_CONSTANT, HS_MANIFOLD_CONTROL_POINT, HS_MANIFOLD_EUCLIDEAN,  # noqa: F401
                   HS_MANIFOLD_SE3, HS_MANIFOLD_SPHERE3)
from .problem import HsError, Problem, Window  # noqa: F401
