"""ctypes binding of the C ABI declared in include/hyperslam_hip.h.

The product library is ``hyperslam_amd/libhyperslam_hip.so`` (HIP, gfx950). There is no CPU fallback: if the
library is missing, ``load()`` raises. The same binding class can be pointed at the oracle (``oracle/liboracle.so``,
symbol prefix ``hso_``) — that is done by tests / bench.py only, never by the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HS_LIBRARY: another build of the SAME library (the profiling build tools/libhyperslam_hip_prof.so with the phase timestamps and the A/B
# kernels compiled in) for the measurement tools; unset in production.
PRODUCT_LIB = os.environ.get("HS_LIBRARY") or os.path.join(_HERE, "libhyperslam_hip.so")

HS_PIXEL, HS_BEARING, HS_PRIOR, HS_INERTIAL = 0, 1, 2, 3
HS_INERTIAL_AS_REFERENCE, HS_INERTIAL_EXACT = 0, 1  # hs_set_inertial_jacobian
(HS_MANIFOLD_CONSTANT, HS_MANIFOLD_EUCLIDEAN, HS_MANIFOLD_CONTROL_POINT, HS_MANIFOLD_SE3, HS_MANIFOLD_SPHERE3,
 HS_MANIFOLD_BIAS_POINT) = range(6)
HS_NO_CONVERGENCE, HS_CONVERGENCE, HS_FAILURE = 0, 1, 2

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class Iteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32),
        ("step_is_valid", C.c_int32),
        ("step_is_successful", C.c_int32),
        ("reserved", C.c_int32),
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double),
        ("relative_decrease", C.c_double),
        ("radius", C.c_double),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("termination", C.c_int32),
        ("num_residual_blocks", C.c_int32),
        ("linearize_ms", C.c_double),
        ("schur_ms", C.c_double),
        ("solve_ms", C.c_double),
        ("update_ms", C.c_double),
        ("total_ms", C.c_double),
    ]


class Linearization(C.Structure):
    _fields_ = [
        ("r", c_double_p),
        ("J_state", c_double_p),
        ("J_landmark", c_double_p),
        ("J_bias_g", c_double_p),
        ("J_bias_a", c_double_p),
        ("J_gravity", c_double_p),
        ("first_cp", c_int32_p),
        ("first_bias", c_int32_p),
        ("cost", c_double_p),
        # sensor parameter blocks (optional; constant in the solver)
        ("J_extrinsics", c_double_p),
        ("J_intrinsics", c_double_p),
        ("J_distortion", c_double_p),
        ("J_gyro_intrinsics", c_double_p),
        ("J_acc_intrinsics", c_double_p),
        ("J_gyro_sensitivity", c_double_p),
        ("J_acc_offsets", c_double_p),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

# name -> (restype, argtypes); the handle is a void*
_SIGNATURES = {
    "create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "destroy": (C.c_int, [C.c_void_p]),
    "last_error": (C.c_char_p, [C.c_void_p]),
    "set_spline": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, c_double_p, c_uint8_p, C.c_int, C.c_int]),
    "set_cameras": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p]),
    "set_sensors": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "set_landmarks": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_uint8_p]),
    "set_imu": (C.c_int, [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int, C.c_double, C.c_double,
                          C.c_int, c_double_p, c_double_p, C.c_int]),
    "set_gravity": (C.c_int, [C.c_void_p, c_double_p, C.c_int]),
    "set_inertial_jacobian": (C.c_int, [C.c_void_p, C.c_int]),
    "set_stage_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "set_weights": (C.c_int, [C.c_void_p, C.c_int, c_double_p]),
    "set_pixel_residuals": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_int32_p, c_int32_p]),
    "set_bearing_residuals": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_int32_p, c_int32_p]),
    "set_prior_residuals": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_int32_p]),
    "set_inertial_residuals": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p]),
    # delta interface (tables kept incrementally between solves)
    "append_landmarks": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_uint8_p, c_int32_p]),
    "append_pixel_residuals": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_int32_p, c_int32_p]),
    "append_bearing_residuals": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_int32_p, c_int32_p]),
    "append_prior_residuals": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_int32_p]),
    "append_inertial_residuals": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p]),
    "retire_landmarks": (C.c_int, [C.c_void_p, C.c_int, c_int32_p, c_int32_p]),
    "retire_residuals_before": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "stage": (C.c_int, [C.c_void_p]),
    "residual_layout": (C.c_int, [C.c_void_p, C.c_int, C.c_int, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    "num_residuals": (C.c_int, [C.c_void_p, C.c_int]),
    "dim_pose": (C.c_int, [C.c_void_p]),
    "linearize": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Linearization)]),
    "cost_function_evaluate": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(c_double_p), c_double_p, C.POINTER(c_double_p)]),
    "cost": (C.c_int, [C.c_void_p, c_double_p]),
    "reduced_system": (C.c_int, [C.c_void_p, C.c_double, c_double_p, c_double_p]),
    "solve": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Summary), C.POINTER(Iteration)]),
    "get_control_points": (C.c_int, [C.c_void_p, c_double_p]),
    "get_landmarks": (C.c_int, [C.c_void_p, c_double_p]),
    "get_bias": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "get_gravity": (C.c_int, [C.c_void_p, c_double_p]),
    "sample_trajectory": (C.c_int, [C.c_void_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
    "process_tracks": (C.c_int, [C.c_void_p, C.c_double, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "manifold_tangent_size": (C.c_int, [C.c_int, C.c_int]),
    "manifold_plus": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]),
    "manifold_plus_jacobian": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p]),
    "manifold_minus": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p]),
    "manifold_minus_jacobian": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, c_double_p, c_double_p]),
    "set_allreduce": (C.c_int, [C.c_void_p, ALLREDUCE_FN, C.c_void_p]),
    "set_shard": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "band_blocks": (C.c_int, [C.c_void_p]),
}
_PRODUCT_ONLY = {
    "set_guard": (C.c_int, [C.c_int]),
    "snapshot": (C.c_int, [C.c_void_p]),
    "restore": (C.c_int, [C.c_void_p]),
    "version": (C.c_int, []),
    "arch": (C.c_char_p, []),
    "rccl_unique_id": (C.c_int, [C.c_char_p]),
    "rccl_init": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "rccl_shutdown": (C.c_int, [C.c_void_p]),
    "exchange_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
}

# every symbol include/hyperslam_hip.h declares (checked by tests/test_oracle.py::test_product_library_exports_every_declared_symbol)
ABI_SYMBOLS = ["hs_" + n for n in list(_SIGNATURES) + list(_PRODUCT_ONLY)]


class Library:
    """A loaded C-ABI library with typed entry points (attribute access without the prefix)."""

    def __init__(self, path: str, prefix: str, strict: bool = True):
        """strict=False: a library that exports only part of the ABI (oracle/liboracle_ld.so — prefix hsl_ —, the long-double referee of the tests and
        tools/fuzz_parity.py): the entry points it lacks are simply not bound."""
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found — build it first (python -c 'import __graft_entry__ as g; g.build()'); "
                "hyperslam_amd has no CPU fallback")
        self.path, self.prefix = path, prefix
        self.cdll = C.CDLL(path)
        table = dict(_SIGNATURES)
        if prefix == "hs_":
            table.update(_PRODUCT_ONLY)
        for name, (res, args) in table.items():
            if not strict and not hasattr(self.cdll, prefix + name):
                continue
            fn = getattr(self.cdll, prefix + name)
            fn.restype, fn.argtypes = res, args
            setattr(self, name, fn)


_product = None


def load() -> Library:
    """The product library (HIP). Raises if it has not been built."""
    global _product
    if _product is None:
        _product = Library(PRODUCT_LIB, "hs_")
    return _product
