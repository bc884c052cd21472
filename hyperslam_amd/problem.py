"""Host-side mirror of the tables behind the C ABI: a sliding-window NLLS problem as flat numpy arrays.

``Window`` is plain data (what the reference keeps in ceres::Problem + Environment + AbstractState,
/root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:189-382); ``Problem`` owns a library handle and moves
the tables into HBM. Names follow the reference's domain (control points, landmarks, observations), not ML vocabulary.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import HS_BEARING, HS_INERTIAL, HS_PIXEL, HS_PRIOR, Iteration, Library, Linearization, Summary

_f64 = np.float64


def _d(a):
    return a.ctypes.data_as(_lib.c_double_p)


def _i(a):
    return a.ctypes.data_as(_lib.c_int32_p)


def _u8(a):
    return a.ctypes.data_as(_lib.c_uint8_p)


def _arr(x, dtype, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=dtype))
    if shape is not None:
        a = a.reshape(shape)
    return a


@dataclass
class Window:
    """One optimisation window (SURVEY.md §8(d) synthetic configs produce these)."""
    order: int = 4
    t0: float = 0.0
    dt: float = 0.1
    control_points: np.ndarray = field(default_factory=lambda: np.zeros((0, 8)))  # [qx qy qz qw px py pz t]
    cp_constant: np.ndarray | None = None
    rotation_constant: bool = False
    translation_constant: bool = False
    cam_T_bs: np.ndarray = field(default_factory=lambda: np.zeros((0, 7)))
    cam_intrinsics: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))
    cam_distortion: np.ndarray = field(default_factory=lambda: np.zeros((0, 4)))
    sensor_T_bs: np.ndarray = field(default_factory=lambda: np.zeros((0, 7)))
    landmarks: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    landmark_constant: np.ndarray | None = None
    # residual tables
    pixel_stamps: np.ndarray = field(default_factory=lambda: np.zeros(0))
    pixels: np.ndarray = field(default_factory=lambda: np.zeros((0, 2)))
    pixel_landmark: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    pixel_camera: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    bearing_stamps: np.ndarray = field(default_factory=lambda: np.zeros(0))
    bearings: np.ndarray = field(default_factory=lambda: np.zeros((0, 3)))
    bearing_landmark: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    bearing_camera: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    prior_stamps: np.ndarray = field(default_factory=lambda: np.zeros(0))
    prior_poses: np.ndarray = field(default_factory=lambda: np.zeros((0, 7)))
    prior_sensor: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    inertial_stamps: np.ndarray = field(default_factory=lambda: np.zeros(0))
    inertial_measurements: np.ndarray = field(default_factory=lambda: np.zeros((0, 6)))
    # IMU (optional)
    imu: dict | None = None  # keys: T_bs, i_g, i_a, S_g, X_a, bias_order, bias_t0, bias_dt, bias_g (n x 4), bias_a, bias_constant
    gravity: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, -9.80665]))
    gravity_constant: bool = True

    @property
    def n_cp(self):
        return int(self.control_points.shape[0])

    def num_residual_blocks(self):
        return len(self.pixel_stamps) + len(self.bearing_stamps) + len(self.prior_stamps) + len(self.inertial_stamps)

    def valid_range(self):
        """Stamps t with all k control points available: [t0 + ((k-1)//2) dt, t0 + (n_cp - k + (k-1)//2 + 1) dt)."""
        k = self.order
        lo = self.t0 + ((k - 1) // 2) * self.dt
        hi = self.t0 + (self.n_cp - k + (k - 1) // 2 + 1) * self.dt
        return lo, hi


class HsError(RuntimeError):
    pass


class Problem:
    """A library handle with the window's tables resident (HBM for the product library)."""

    def __init__(self, window: Window, lib: Library | None = None, device: int = 0, stream: int | None = None):
        self.lib = lib if lib is not None else _lib.load()
        self.window = window
        h = C.c_void_p()
        rc = self.lib.create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0 or not h:
            raise HsError(f"{self.lib.prefix}create failed with code {rc} (no usable GPU?)")
        self.h = h
        self._keep = []
        self.upload(window)

    # -- lifetime ---------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.last_error(self.h)
            raise HsError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    # -- tables ------------------------------------------------------------------------------------------------
    def upload(self, w: Window):
        L, h = self.lib, self.h
        cp = _arr(w.control_points, _f64, (-1, 8))
        cpc = None if w.cp_constant is None else _arr(w.cp_constant, np.uint8)
        self._check(L.set_spline(h, w.order, w.t0, w.dt, cp.shape[0], _d(cp), None if cpc is None else _u8(cpc),
                                 int(w.rotation_constant), int(w.translation_constant)), "set_spline")
        T, I, D = _arr(w.cam_T_bs, _f64, (-1, 7)), _arr(w.cam_intrinsics, _f64, (-1, 4)), _arr(w.cam_distortion, _f64, (-1, 4))
        self._check(L.set_cameras(h, T.shape[0], _d(T), _d(I), _d(D)), "set_cameras")
        S = _arr(w.sensor_T_bs, _f64, (-1, 7))
        self._check(L.set_sensors(h, S.shape[0], _d(S)), "set_sensors")
        lm = _arr(w.landmarks, _f64, (-1, 3))
        lmc = None if w.landmark_constant is None else _arr(w.landmark_constant, np.uint8)
        self._check(L.set_landmarks(h, lm.shape[0], _d(lm), None if lmc is None else _u8(lmc)), "set_landmarks")
        if w.imu is not None:
            m = w.imu
            a = [_arr(m[k], _f64) for k in ("T_bs", "i_g", "i_a", "S_g", "X_a")]
            bg, ba = _arr(m["bias_g"], _f64, (-1, 4)), _arr(m["bias_a"], _f64, (-1, 4))
            self._check(L.set_imu(h, *[_d(x) for x in a], int(m["bias_order"]), float(m["bias_t0"]), float(m["bias_dt"]), bg.shape[0],
                                  _d(bg), _d(ba), int(m.get("bias_constant", False))), "set_imu")
        g = _arr(w.gravity, _f64, (3,))
        self._check(L.set_gravity(h, _d(g), int(w.gravity_constant)), "set_gravity")
        st, px = _arr(w.pixel_stamps, _f64), _arr(w.pixels, _f64, (-1, 2))
        li, ci = _arr(w.pixel_landmark, np.int32), _arr(w.pixel_camera, np.int32)
        self._check(L.set_pixel_residuals(h, st.shape[0], _d(st), _d(px), _i(li), _i(ci)), "set_pixel_residuals")
        st, b = _arr(w.bearing_stamps, _f64), _arr(w.bearings, _f64, (-1, 3))
        li, ci = _arr(w.bearing_landmark, np.int32), _arr(w.bearing_camera, np.int32)
        self._check(L.set_bearing_residuals(h, st.shape[0], _d(st), _d(b), _i(li), _i(ci)), "set_bearing_residuals")
        st, po, si = _arr(w.prior_stamps, _f64), _arr(w.prior_poses, _f64, (-1, 7)), _arr(w.prior_sensor, np.int32)
        self._check(L.set_prior_residuals(h, st.shape[0], _d(st), _d(po), _i(si)), "set_prior_residuals")
        st, me = _arr(w.inertial_stamps, _f64), _arr(w.inertial_measurements, _f64, (-1, 6))
        self._check(L.set_inertial_residuals(h, st.shape[0], _d(st), _d(me)), "set_inertial_residuals")
        self.window = w

    # -- delta interface (tables kept incrementally between solves; include/hyperslam_hip.h) ---------------------
    def append_landmarks(self, xyz, constant=None):
        """addLandmark (optimizer.cpp:347-358): rows appended to the landmark table; returns the index of the first new row."""
        lm = _arr(xyz, _f64, (-1, 3))
        c = None if constant is None else _arr(constant, np.uint8)
        first = C.c_int32(-1)
        self._check(self.lib.append_landmarks(self.h, lm.shape[0], _d(lm), None if c is None else _u8(c), C.byref(first)), "append_landmarks")
        return first.value

    def append_residuals(self, ftype, stamps, values, landmark=None, camera=None, sensor=None):
        """add(observation) (optimizer.cpp:189-274): rows appended to the residual table of one factor type."""
        st = _arr(stamps, _f64)
        n = st.shape[0]
        if ftype in (HS_PIXEL, HS_BEARING):
            v = _arr(values, _f64, (-1, 2 if ftype == HS_PIXEL else 3))
            li, ci = _arr(landmark, np.int32), _arr(camera, np.int32)
            fn = self.lib.append_pixel_residuals if ftype == HS_PIXEL else self.lib.append_bearing_residuals
            self._check(fn(self.h, n, _d(st), _d(v), _i(li), _i(ci)), "append_residuals")
        elif ftype == HS_PRIOR:
            v, si = _arr(values, _f64, (-1, 7)), _arr(sensor if sensor is not None else np.zeros(n), np.int32)
            self._check(self.lib.append_prior_residuals(self.h, n, _d(st), _d(v), _i(si)), "append_residuals")
        else:
            v = _arr(values, _f64, (-1, 6))
            self._check(self.lib.append_inertial_residuals(self.h, n, _d(st), _d(v)), "append_residuals")

    def retire_landmarks(self, ids):
        """updateLandmarks (optimizer.cpp:360-382): the landmarks leave with their residual blocks; returns new index per old row (-1: retired)."""
        ids = _arr(ids, np.int32)
        rows = self.num_landmarks()
        remap = np.zeros(max(rows, 1), np.int32)
        self._check(self.lib.retire_landmarks(self.h, ids.shape[0], _i(ids), _i(remap)), "retire_landmarks")
        return remap[:rows].copy()

    def retire_residuals_before(self, ftype, stamp):
        self._check(self.lib.retire_residuals_before(self.h, int(ftype), float(stamp)), "retire_residuals_before")

    def stage(self):
        """Sorts and uploads what changed since the tables were last staged (nothing is awaited)."""
        self._check(self.lib.stage(self.h), "stage")

    def set_control_points(self, control_points, cp_constant=None):
        """The control-point table re-sent with the knots of the resident window (values + constancy mask only)."""
        w = self.window
        cp = _arr(control_points, _f64, (-1, 8))
        cpc = None if cp_constant is None else _arr(cp_constant, np.uint8)
        self._check(self.lib.set_spline(self.h, w.order, w.t0, w.dt, cp.shape[0], _d(cp), None if cpc is None else _u8(cpc),
                                        int(w.rotation_constant), int(w.translation_constant)), "set_spline")

    # -- structure ---------------------------------------------------------------------------------------------
    def num_residuals(self, ftype):
        return self.lib.num_residuals(self.h, ftype)

    def dim_pose(self):
        return self.lib.dim_pose(self.h)

    def residual_layout(self, ftype, idx):
        nb, npar, nres = C.c_int32(), C.c_int32(), C.c_int32()
        ind = np.zeros(4, np.int32)
        sizes, offs, ids = np.zeros(32, np.int32), np.zeros(32, np.int32), np.zeros(32, np.int32)
        self._check(self.lib.residual_layout(self.h, ftype, idx, C.byref(nb), _i(ind), _i(sizes), _i(offs), _i(ids), C.byref(npar), C.byref(nres)),
                    "residual_layout")
        n = nb.value
        return dict(num_blocks=n, indices=ind.copy(), sizes=sizes[:n].copy(), offsets=offs[:n].copy(), block_ids=ids[:n].copy(),
                    num_parameters=npar.value, num_residuals=nres.value)

    # -- evaluation --------------------------------------------------------------------------------------------
    def set_weights(self, ftype, weights):
        """CostConfiguration::weights of one factor type (n_res x n_res; None clears): honoured by linearize / cost_function_evaluate."""
        w = None if weights is None else np.ascontiguousarray(weights, dtype=_f64)
        self._check(self.lib.set_weights(self.h, int(ftype), None if w is None else _d(w)), "set_weights")

    def set_stage_timing(self, enabled=True):
        """Per-stage device times in the summary of solve() (four HIP events per iteration, ~5.7 us of idle device each): off by default."""
        self._check(self.lib.set_stage_timing(self.h, int(bool(enabled))), "set_stage_timing")

    def set_inertial_jacobian(self, mode):
        """HS_INERTIAL_AS_REFERENCE (0, default: inertial.cpp:131-198 as written) | HS_INERTIAL_EXACT (1: derivative of the prediction)."""
        self._check(self.lib.set_inertial_jacobian(self.h, int(mode)), "set_inertial_jacobian")

    def linearize(self, ftype, robustify=True, sensor_blocks=False):
        """Residuals + Ceres-local Jacobians of every residual block of one factor type (table order). sensor_blocks adds the
        Jacobians w.r.t. the sensor parameter blocks (extrinsics; pixel: intrinsics, distortion; inertial: i_g, i_a, S_g, X_a)."""
        w = self.window
        n = self.num_residuals(ftype)
        k = w.order
        nres = {HS_PIXEL: 2, HS_BEARING: 1, HS_PRIOR: 6, HS_INERTIAL: 6}[ftype]
        out = dict(r=np.zeros((n, nres)), J_state=np.zeros((n, nres, 6 * k)), first_cp=np.zeros(n, np.int32), cost=np.zeros(n))
        lin = Linearization()
        lin.r, lin.J_state, lin.first_cp, lin.cost = _d(out["r"]), _d(out["J_state"]), _i(out["first_cp"]), _d(out["cost"])
        if ftype in (HS_PIXEL, HS_BEARING):
            out["J_landmark"] = np.zeros((n, nres, 3))
            lin.J_landmark = _d(out["J_landmark"])
        if ftype == HS_INERTIAL:
            kb = int(w.imu["bias_order"])
            out["J_bias_g"], out["J_bias_a"] = np.zeros((n, 6, 3 * kb)), np.zeros((n, 6, 3 * kb))
            out["J_gravity"], out["first_bias"] = np.zeros((n, 6, 2)), np.zeros(n, np.int32)
            lin.J_bias_g, lin.J_bias_a, lin.J_gravity, lin.first_bias = _d(out["J_bias_g"]), _d(out["J_bias_a"]), _d(out["J_gravity"]), _i(out["first_bias"])
        if sensor_blocks:
            out["J_extrinsics"] = np.zeros((n, nres, 6))
            lin.J_extrinsics = _d(out["J_extrinsics"])
            if ftype == HS_PIXEL:
                out["J_intrinsics"], out["J_distortion"] = np.zeros((n, 2, 4)), np.zeros((n, 2, 4))
                lin.J_intrinsics, lin.J_distortion = _d(out["J_intrinsics"]), _d(out["J_distortion"])
            if ftype == HS_INERTIAL:
                for name, cols in (("J_gyro_intrinsics", 6), ("J_acc_intrinsics", 6), ("J_gyro_sensitivity", 9), ("J_acc_offsets", 9)):
                    out[name] = np.zeros((n, 6, cols))
                    setattr(lin, name, _d(out[name]))
        self._check(self.lib.linearize(self.h, ftype, int(robustify), C.byref(lin)), "linearize")
        return out

    def parameter_blocks(self, ftype, idx):
        """Values of the parameter blocks of residual idx in ExteroceptiveCost::update order (state || sensor || observation)."""
        w, L = self.window, self.residual_layout(ftype, idx)
        ids, k = L["block_ids"], w.order
        blocks = [np.array(w.control_points[ids[j]], float) for j in range(k)]
        if ftype in (HS_PIXEL, HS_BEARING):
            c = ids[k]
            blocks += [np.array(w.cam_T_bs[c], float), np.array(w.cam_intrinsics[c], float), np.array(w.cam_distortion[c], float),
                       np.array(w.landmarks[ids[k + 3]], float)]
        elif ftype == HS_PRIOR:
            blocks.append(np.array(w.sensor_T_bs[ids[k]], float))
        else:
            m, kb = w.imu, int(w.imu["bias_order"])
            blocks += [np.array(m[n], float) for n in ("T_bs", "i_g", "i_a", "S_g", "X_a")]
            blocks += [np.array(m["bias_g"][ids[k + 5 + j]], float) for j in range(kb)]
            blocks += [np.array(m["bias_a"][ids[k + 5 + kb + j]], float) for j in range(kb)]
            blocks.append(np.array(w.gravity, float))
        return blocks

    def cost_function_evaluate(self, ftype, idx, blocks, want=None):
        """ceres::CostFunction::Evaluate contract (exteroceptive.hpp:31): residuals + row-major per-block ambient Jacobians
        (None where `want[i]` is False, like Ceres' nullptr for constant blocks)."""
        L = self.residual_layout(ftype, idx)
        nb, nres = L["num_blocks"], L["num_residuals"]
        blocks = [np.ascontiguousarray(b, dtype=_f64) for b in blocks]
        params = (_lib.c_double_p * nb)(*[_d(b) for b in blocks])
        res = np.zeros(nres)
        if want is None:
            self._check(self.lib.cost_function_evaluate(self.h, ftype, idx, params, _d(res), None), "cost_function_evaluate")
            return res, None
        jac = [np.zeros((nres, int(L["sizes"][i]))) if want[i] else None for i in range(nb)]
        jptr = (_lib.c_double_p * nb)(*[(_d(j) if j is not None else None) for j in jac])
        self._check(self.lib.cost_function_evaluate(self.h, ftype, idx, params, _d(res), jptr), "cost_function_evaluate")
        return res, jac

    def cost(self):
        c = C.c_double()
        self._check(self.lib.cost(self.h, C.byref(c)), "cost")
        return c.value

    def reduced_system(self, radius=1e4):
        n = self.dim_pose()
        S, g = np.zeros((n, n)), np.zeros(n)
        self._check(self.lib.reduced_system(self.h, float(radius), _d(S), _d(g)), "reduced_system")
        return S, g

    def solve(self, max_iterations=5):
        """CeresOptimizer::optimize (optimizer.cpp:276-280; max_num_iterations = 5, optimizer.cpp:40)."""
        s = Summary()
        its = (Iteration * (max_iterations + 1))()
        self._check(self.lib.solve(self.h, max_iterations, C.byref(s), its), "solve")
        fields = [f[0] for f in Iteration._fields_]
        n = min(s.num_iterations, max_iterations)
        iterations = [{f: getattr(its[i], f) for f in fields} for i in range(n + 1)]
        summary = {f[0]: getattr(s, f[0]) for f in Summary._fields_}
        summary["iterations"] = iterations
        return summary

    def snapshot(self):
        self._check(self.lib.snapshot(self.h), "snapshot")

    def restore(self):
        self._check(self.lib.restore(self.h), "restore")

    # -- read back ---------------------------------------------------------------------------------------------
    def control_points(self):
        cp = np.zeros((self.window.n_cp, 8))
        self._check(self.lib.get_control_points(self.h, _d(cp)), "get_control_points")
        return cp

    def num_landmarks(self):
        """Rows of the library's landmark table (the delta interface appends / retires rows: not necessarily the window's)."""
        if not hasattr(self.lib, "append_landmarks"):  # (oracle/liboracle_ld.so: the referee binds the whole-table entry points only)
            return len(self.window.landmarks)
        rows = C.c_int32(0)
        self._check(self.lib.append_landmarks(self.h, 0, None, None, C.byref(rows)), "append_landmarks")  # (appending nothing reports the row count)
        return rows.value

    def landmarks(self):
        lm = np.zeros((self.num_landmarks(), 3))
        if len(lm):
            self._check(self.lib.get_landmarks(self.h, _d(lm)), "get_landmarks")
        return lm

    def gravity(self):
        g = np.zeros(3)
        self._check(self.lib.get_gravity(self.h, _d(g)), "get_gravity")
        return g

    def bias(self):
        n = len(self.window.imu["bias_g"])
        bg, ba = np.zeros((n, 4)), np.zeros((n, 4))
        self._check(self.lib.get_bias(self.h, _d(bg), _d(ba)), "get_bias")
        return bg, ba

    def process_tracks(self, stamp, pixels0, pixels1):
        """AbstractOptimizer::process(VisualTracks) front half: (bearings0, bearings1, positions_w) of n stereo tracks."""
        p0, p1 = _arr(pixels0, _f64).reshape(-1, 2), _arr(pixels1, _f64).reshape(-1, 2)
        n = len(p0)
        b0, b1, pw = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
        self._check(self.lib.process_tracks(self.h, float(stamp), n, _d(p0), _d(p1), _d(b0), _d(b1), _d(pw)), "process_tracks")
        return b0, b1, pw

    def manifold_plus(self, kind, x, delta):
        """Batched ceres::Manifold::Plus of variable class `kind` (HS_MANIFOLD_*): x (n, ambient), delta (n, tangent)."""
        x = _arr(x, _f64)
        x = x.reshape(-1, x.shape[-1])
        n, ambient = x.shape
        tangent = self.lib.manifold_tangent_size(int(kind), ambient)
        d = _arr(delta, _f64).reshape(n, max(tangent, 0)) if tangent > 0 else np.zeros((n, 0))
        out = np.zeros((n, ambient))
        self._check(self.lib.manifold_plus(self.h, int(kind), ambient, n, _d(x), _d(d) if tangent > 0 else None, _d(out)), "manifold_plus")
        return out

    def manifold_plus_jacobian(self, kind, x):
        """Batched ceres::Manifold::PlusJacobian: (n, ambient, tangent), row-major per element."""
        x = _arr(x, _f64)
        x = x.reshape(-1, x.shape[-1])
        n, ambient = x.shape
        tangent = self.lib.manifold_tangent_size(int(kind), ambient)
        if tangent < 0:
            raise ValueError("unknown manifold kind / ambient size")
        jac = np.zeros((n, ambient, tangent))
        if tangent > 0:
            self._check(self.lib.manifold_plus_jacobian(self.h, int(kind), ambient, n, _d(x), _d(jac)), "manifold_plus_jacobian")
        return jac

    def manifold_minus(self, kind, y, x):
        """Batched ceres::Manifold::Minus: y, x (n, ambient) -> (n, tangent), the tangent vector with Plus(x, .) = y."""
        x = _arr(x, _f64)
        x = x.reshape(-1, x.shape[-1])
        n, ambient = x.shape
        y = _arr(y, _f64).reshape(n, ambient)
        tangent = self.lib.manifold_tangent_size(int(kind), ambient)
        if tangent < 0:
            raise ValueError("unknown manifold kind / ambient size")
        out = np.zeros((n, tangent))
        if tangent > 0:
            self._check(self.lib.manifold_minus(self.h, int(kind), ambient, n, _d(y), _d(x), _d(out)), "manifold_minus")
        return out

    def manifold_minus_jacobian(self, kind, x):
        """Batched ceres::Manifold::MinusJacobian: (n, tangent, ambient), row-major per element."""
        x = _arr(x, _f64)
        x = x.reshape(-1, x.shape[-1])
        n, ambient = x.shape
        tangent = self.lib.manifold_tangent_size(int(kind), ambient)
        if tangent < 0:
            raise ValueError("unknown manifold kind / ambient size")
        jac = np.zeros((n, tangent, ambient))
        if tangent > 0:
            self._check(self.lib.manifold_minus_jacobian(self.h, int(kind), ambient, n, _d(x), _d(jac)), "manifold_minus_jacobian")
        return jac

    def sample_trajectory(self, stamps, derivatives=False):
        st = _arr(stamps, _f64)
        pose = np.zeros((len(st), 7))
        vel = np.zeros((len(st), 6)) if derivatives else None
        acc = np.zeros((len(st), 6)) if derivatives else None
        self._check(self.lib.sample_trajectory(self.h, len(st), _d(st), _d(pose), None if vel is None else _d(vel),
                                               None if acc is None else _d(acc)), "sample_trajectory")
        return (pose, vel, acc) if derivatives else pose
