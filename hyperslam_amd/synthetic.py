"""Deterministic synthetic windows mirroring BASELINE.json:configs.

RNG = SplitMix64 -> (x >> 11) * 2^-53 uniforms, Box-Muller normals; seed = 0x48595045 ^ config index. Ground truth:
p(t) = (2 sin 0.8t, 2 cos 0.6t, sin 0.4t) m, rotation vector phi(t) = 0.5 (sin 0.5t, cos 0.3t, sin 0.7t) rad.
Camera / IMU parameters are the EuRoC values of
/root/reference/resources/datasets/euroc/setups/stereo_inertial/settings.yaml:20-109 (restated here as constants; the
GPU box has no /root/reference).
"""
from __future__ import annotations

import numpy as np

from .problem import Window

SEED = 0x48595045

# settings.yaml:33-45 / :60-72
EUROC_CAM_T_BS = np.array([
    [-0.007707179755532, 0.010499323370595, 0.701752800292141, 0.712301460668946, -0.0216401454975, -0.064676986768, 0.00981073058949],
    [-0.002550236745188, 0.015323927487975, 0.702486685782579, 0.711527321918909, -0.0198435579556, 0.0453689425024, 0.00786212447038],
])
EUROC_CAM_INTRINSICS = np.array([[367.215, 248.375, 458.654, 457.296], [379.999, 255.238, 457.587, 456.134]])  # [cx cy fx fy]
EUROC_CAM_DISTORTION = np.array([[-0.28340811, 0.07395907, 1.76187114e-05, 0.00019359], [-0.28368365, 0.07451284, -3.55590700e-05, -0.00010473]])
GYRO_NOISE_DENSITY, ACCEL_NOISE_DENSITY, IMU_RATE = 1.6968e-04, 2.0000e-3, 200.0  # settings.yaml:96-97,108-109,81
GRAVITY_NORM = 9.80665


class SplitMix64:
    """Counter-based view of SplitMix64: output n = mix(seed + n * gamma); vectorised draws consume the counter."""
    GAMMA = np.uint64(0x9E3779B97F4A7C15)

    def __init__(self, seed: int):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def _raw(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + idx * self.GAMMA
            self.state = np.uint64(z[-1]) if n else self.state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def uniform(self, *shape, lo=0.0, hi=1.0) -> np.ndarray:
        n = int(np.prod(shape)) if shape else 1
        u = (self._raw(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        u = lo + (hi - lo) * u
        return u.reshape(shape) if shape else float(u[0])

    def normal(self, *shape, sigma=1.0) -> np.ndarray:
        n = int(np.prod(shape)) if shape else 1
        u1 = 1.0 - self.uniform(n)  # (0, 1]
        u2 = self.uniform(n)
        z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2) * sigma
        return z.reshape(shape) if shape else float(z[0])


# ---- small batched SO(3) / quaternion helpers (x, y, z, w) -------------------------------------------------------------
def quat_mul(a, b):
    ax, ay, az, aw = np.moveaxis(a, -1, 0)
    bx, by, bz, bw = np.moveaxis(b, -1, 0)
    return np.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz], -1)


def quat_exp(phi):
    phi = np.asarray(phi, float)
    t = np.linalg.norm(phi, axis=-1, keepdims=True)
    half = 0.5 * t
    s = np.where(t < 1e-8, 0.5 - t * t / 48.0, np.sin(half) / np.where(t < 1e-8, 1.0, t))
    return np.concatenate([s * phi, np.cos(half)], -1)


def quat_to_matrix(q):
    x, y, z, w = np.moveaxis(np.asarray(q, float), -1, 0)
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 0, 2] = 1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)
    R[..., 1, 0], R[..., 1, 1], R[..., 1, 2] = 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)
    R[..., 2, 0], R[..., 2, 1], R[..., 2, 2] = 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)
    return R


def gt_position(t):
    t = np.asarray(t, float)
    return np.stack([2 * np.sin(0.8 * t), 2 * np.cos(0.6 * t), np.sin(0.4 * t)], -1)


def gt_rotvec(t):
    t = np.asarray(t, float)
    return 0.5 * np.stack([np.sin(0.5 * t), np.cos(0.3 * t), np.sin(0.7 * t)], -1)


def gt_pose(t):
    """(q_wb (.., 4), p_wb (.., 3)) of the ground-truth trajectory."""
    return quat_exp(gt_rotvec(t)), gt_position(t)


def compose(qa, pa, qb, pb):
    """T_a o T_b."""
    Ra = quat_to_matrix(qa)
    return quat_mul(qa, qb), np.einsum("...ij,...j->...i", Ra, pb) + pa


def project_radtan(p_s, intr, dist):
    x, y = p_s[..., 0] / p_s[..., 2], p_s[..., 1] / p_s[..., 2]
    k1, k2, p1, p2 = np.moveaxis(dist, -1, 0)
    cx, cy, fx, fy = np.moveaxis(intr, -1, 0)
    r2 = x * x + y * y
    rad = 1 + k1 * r2 + k2 * r2 * r2
    xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([cx + fx * xd, cy + fy * yd], -1)


def make_control_points(rng, order, dt, n_cp, noise=1e-2):
    """Control points at t_j = (j - (k-1)//2) dt (bootstrap rule, abstract.cpp:89) = GT sampled at the knots + tangent noise."""
    t0 = -((order - 1) // 2) * dt
    t = t0 + dt * np.arange(n_cp)
    q, p = gt_pose(t)
    q = quat_mul(quat_exp(rng.normal(n_cp, 3, sigma=noise)), q)
    p = p + rng.normal(n_cp, 3, sigma=noise)
    cp = np.concatenate([q, p, t[:, None]], -1)
    return t0, cp


def _visual_window(seed, order, n_cp, n_lm, obs_pairs, bearing=False, dt=0.1, pixel_noise=0.5, lm_noise=0.05, span=1.0):
    rng = SplitMix64(seed)
    t0, cp = make_control_points(rng, order, dt, n_cp)
    w = Window(order=order, t0=t0, dt=dt, control_points=cp, cam_T_bs=EUROC_CAM_T_BS.copy(), cam_intrinsics=EUROC_CAM_INTRINSICS.copy(),
               cam_distortion=EUROC_CAM_DISTORTION.copy())
    lo, hi = w.valid_range()
    eps = 1e-9
    # landmarks: anchor stamp, pixel in cam0, depth -> back-projected through the ground truth
    ta = rng.uniform(n_lm, lo=lo, hi=hi - eps)
    px = np.stack([rng.uniform(n_lm, lo=0.0, hi=752.0), rng.uniform(n_lm, lo=0.0, hi=480.0)], -1)
    depth = rng.uniform(n_lm, lo=2.0, hi=10.0)
    cx, cy, fx, fy = EUROC_CAM_INTRINSICS[0]
    ps = np.stack([(px[:, 0] - cx) / fx * depth, (px[:, 1] - cy) / fy * depth, depth], -1)
    qa, pa = gt_pose(ta)
    q_ws, p_ws = compose(qa, pa, np.broadcast_to(EUROC_CAM_T_BS[0, :4], (n_lm, 4)), np.broadcast_to(EUROC_CAM_T_BS[0, 4:], (n_lm, 3)))
    lm_gt = np.einsum("nij,nj->ni", quat_to_matrix(q_ws), ps) + p_ws
    # observations: obs_pairs stereo pairs per landmark at stamps within +-span/2 of the anchor
    st = ta[:, None] + rng.uniform(n_lm, obs_pairs, lo=-0.5 * span, hi=0.5 * span)
    st[:, 0] = ta
    st = np.clip(st, lo, hi - eps)
    stamps = np.repeat(st[:, :, None], 2, axis=2)                       # (n_lm, pairs, 2 cams)
    cams = np.broadcast_to(np.array([0, 1], np.int32), stamps.shape).copy()
    lmi = np.broadcast_to(np.arange(n_lm, dtype=np.int32)[:, None, None], stamps.shape).copy()

    def sensor_points(stf, camf, lmf):
        qb, pb = gt_pose(stf)
        qs, psw = compose(qb, pb, EUROC_CAM_T_BS[camf, :4], EUROC_CAM_T_BS[camf, 4:])
        return np.einsum("nji,nj->ni", quat_to_matrix(qs), lm_gt[lmf] - psw)

    stf, camf, lmf = stamps.reshape(-1), cams.reshape(-1), lmi.reshape(-1)
    p_s = sensor_points(stf, camf, lmf)
    bad = p_s[:, 2] < 0.5  # behind / too close: fall back to the anchor stamp (always in front of cam0; cam1 is 11 cm away)
    stf = np.where(bad, ta[lmf], stf)
    p_s = sensor_points(stf, camf, lmf)
    n_obs = len(stf)
    if bearing:
        b = p_s / np.linalg.norm(p_s, axis=-1, keepdims=True)
        b = b + rng.normal(n_obs, 3, sigma=pixel_noise / 458.0)  # ~ one-pixel angular noise (optimizer.cpp:203 comment)
        b /= np.linalg.norm(b, axis=-1, keepdims=True)
        w.bearing_stamps, w.bearings, w.bearing_landmark, w.bearing_camera = stf, b, lmf, camf
    else:
        meas = project_radtan(p_s, EUROC_CAM_INTRINSICS[camf], EUROC_CAM_DISTORTION[camf]) + rng.normal(n_obs, 2, sigma=pixel_noise)
        w.pixel_stamps, w.pixels, w.pixel_landmark, w.pixel_camera = stf, meas, lmf, camf
    w.landmarks = lm_gt + rng.normal(n_lm, 3, sigma=lm_noise)
    return w, rng


def config0(n_cp=32, n_prior=1000):
    """configs[0]: cubic SE3 B-spline, 32 control points, 1k pose-prior residuals (CPU plumbing case)."""
    rng = SplitMix64(SEED ^ 0)
    t0, cp = make_control_points(rng, 4, 0.1, n_cp)
    w = Window(order=4, t0=t0, dt=0.1, control_points=cp)
    lo, hi = w.valid_range()
    q_bs = quat_exp(rng.uniform(1, 3, lo=-1.0, hi=1.0))
    p_bs = rng.uniform(1, 3, lo=-1.0, hi=1.0)
    w.sensor_T_bs = np.concatenate([q_bs, p_bs], -1)
    st = rng.uniform(n_prior, lo=lo, hi=hi - 1e-9)
    qb, pb = gt_pose(st)
    qm, pm = compose(qb, pb, np.broadcast_to(q_bs[0], (n_prior, 4)), np.broadcast_to(p_bs[0], (n_prior, 3)))
    qm = quat_mul(quat_exp(rng.normal(n_prior, 3, sigma=1e-2)), qm)
    pm = pm + rng.normal(n_prior, 3, sigma=1e-2)
    w.prior_stamps, w.prior_poses, w.prior_sensor = st, np.concatenate([qm, pm], -1), np.zeros(n_prior, np.int32)
    return w


def config1(n_cp=128, n_landmarks=5000, obs_pairs=5, bearing=False):
    """configs[1]: order-4 spline, 128 control points, 50k visual reprojection residuals + 5k landmarks."""
    w, _ = _visual_window(SEED ^ 1, 4, n_cp, n_landmarks, obs_pairs, bearing=bearing)
    return w


def config3(n_cp=512, n_landmarks=20000, obs_pairs=5):
    """configs[3]: 512 control points, 200k residual blocks, 20k landmarks (sharded by landmark over 8 GPUs)."""
    w, _ = _visual_window(SEED ^ 3, 4, n_cp, n_landmarks, obs_pairs)
    return w


def small_visual(order=4, n_cp=16, n_landmarks=40, obs_pairs=3, bearing=False, seed=7, with_priors=0, span=1.0):
    """Small window for oracle-sized parity tests (`span` = seconds a landmark's observations are spread over)."""
    w, rng = _visual_window(SEED ^ (0x100 + seed), order, n_cp, n_landmarks, obs_pairs, bearing=bearing, span=span)
    if with_priors:
        lo, hi = w.valid_range()
        q_bs = quat_exp(rng.uniform(1, 3, lo=-0.5, hi=0.5))
        p_bs = rng.uniform(1, 3, lo=-0.2, hi=0.2)
        w.sensor_T_bs = np.concatenate([q_bs, p_bs], -1)
        st = rng.uniform(with_priors, lo=lo, hi=hi - 1e-9)
        qb, pb = gt_pose(st)
        qm, pm = compose(qb, pb, np.broadcast_to(q_bs[0], (with_priors, 4)), np.broadcast_to(p_bs[0], (with_priors, 3)))
        qm = quat_mul(quat_exp(rng.normal(with_priors, 3, sigma=1e-2)), qm)
        pm = pm + rng.normal(with_priors, 3, sigma=1e-2)
        w.prior_stamps, w.prior_poses, w.prior_sensor = st, np.concatenate([qm, pm], -1), np.zeros(with_priors, np.int32)
    return w


def add_imu(w: Window, rng: SplitMix64, n_inertial: int, bias_dt=1.0, bias_order=4, identity=True, gravity_constant=False):
    """Adds an IMU (EuRoC parameters, settings.yaml:83-109), its R^3 bias splines (free, optimizer.cpp:62-63), gravity and
    `n_inertial` direct inertial residuals (model of inertial.cpp:200-203 evaluated on the ground truth + noise)."""
    lo, hi = w.valid_range()
    kb = bias_order
    n_bias = int(np.ceil((hi - lo) / bias_dt)) + kb
    bias_t0 = lo - ((kb - 1) // 2) * bias_dt - 1e-3
    stamps_b = bias_t0 + bias_dt * np.arange(n_bias)
    bg = np.concatenate([rng.normal(n_bias, 3, sigma=1e-3), stamps_b[:, None]], -1)
    ba = np.concatenate([rng.normal(n_bias, 3, sigma=1e-2), stamps_b[:, None]], -1)
    if identity:
        T_bs = np.array([0, 0, 0, 1, 0, 0, 0.0])
        i_g = np.array([1, 1, 1, 0, 0, 0.0])
        i_a = i_g.copy()
        S_g, X_a = np.zeros(9), np.zeros(9)
    else:
        T_bs = np.concatenate([quat_exp(rng.uniform(3, lo=-0.3, hi=0.3)), rng.uniform(3, lo=-0.1, hi=0.1)])
        i_g = np.array([1, 1, 1, 0, 0, 0.0]) + rng.uniform(6, lo=-0.05, hi=0.05)
        i_a = np.array([1, 1, 1, 0, 0, 0.0]) + rng.uniform(6, lo=-0.05, hi=0.05)
        S_g, X_a = rng.uniform(9, lo=-1e-3, hi=1e-3), rng.uniform(9, lo=-0.02, hi=0.02)
    w.imu = dict(T_bs=T_bs, i_g=i_g, i_a=i_a, S_g=S_g, X_a=X_a, bias_order=kb, bias_t0=bias_t0, bias_dt=bias_dt, bias_g=bg, bias_a=ba,
                 bias_constant=False)
    g_true = np.array([0.0, 0.0, -GRAVITY_NORM])
    # initial gravity slightly tilted (the reference initialises (-kNorm, 0, 0)-style and frees it while window == state range)
    tilt = quat_to_matrix(quat_exp(rng.normal(3, sigma=0.02)))
    w.gravity, w.gravity_constant = tilt @ g_true, gravity_constant
    st = rng.uniform(n_inertial, lo=lo, hi=min(hi, stamps_b[-1] - (kb // 2) * bias_dt) - 1e-9)
    # ground-truth body rates by differentiating the analytic trajectory numerically (central differences, h = 1e-4)
    h = 1e-4
    q0, _ = gt_pose(st)
    qp, _ = gt_pose(st + h)
    qm, _ = gt_pose(st - h)
    R0 = quat_to_matrix(q0)
    dR = (quat_to_matrix(qp) - quat_to_matrix(qm)) / (2 * h)
    Om = np.einsum("nji,njk->nik", R0, dR)
    w_b = np.stack([Om[:, 2, 1], Om[:, 0, 2], Om[:, 1, 0]], -1)
    a_w = (gt_position(st + h) - 2 * gt_position(st) + gt_position(st - h)) / (h * h)
    a_b = np.einsum("nji,nj->ni", R0, a_w - g_true)
    meas = np.concatenate([w_b, a_b], -1)  # identity-IMU model at the ground truth (lever arm 0)
    meas[:, :3] += rng.normal(n_inertial, 3, sigma=GYRO_NOISE_DENSITY * np.sqrt(IMU_RATE))
    meas[:, 3:] += rng.normal(n_inertial, 3, sigma=ACCEL_NOISE_DENSITY * np.sqrt(IMU_RATE))
    w.inertial_stamps, w.inertial_measurements = st, meas
    return w


def config2(n_cp=128, n_landmarks=5000, obs_pairs=5, n_inertial=10000):
    """configs[2]: stereo-inertial, order-6 spline, 128 control points, 10k IMU + 50k stereo residuals, Schur on landmarks."""
    w, rng = _visual_window(SEED ^ 2, 6, n_cp, n_landmarks, obs_pairs)
    return add_imu(w, rng, n_inertial)


def small_inertial(order=4, n_cp=16, n_landmarks=30, obs_pairs=3, n_inertial=80, seed=21, identity=False):
    w, rng = _visual_window(SEED ^ (0x200 + seed), order, n_cp, n_landmarks, obs_pairs)
    return add_imu(w, rng, n_inertial, identity=identity)


def shard_by_landmark(w: Window, rank: int, world: int) -> Window:
    """Residual-block sharding of SURVEY.md §8(e): all observations of a landmark stay on one rank (landmark l -> rank
    l % world); pose-prior / inertial residuals are split by contiguous index ranges. Tables that are replicated
    (control points, sensors) are shared."""
    import copy
    s = copy.copy(w)
    if len(w.pixel_stamps):
        m = (w.pixel_landmark % world) == rank
        s.pixel_stamps, s.pixels, s.pixel_landmark, s.pixel_camera = w.pixel_stamps[m], w.pixels[m], w.pixel_landmark[m], w.pixel_camera[m]
    if len(w.bearing_stamps):
        m = (w.bearing_landmark % world) == rank
        s.bearing_stamps, s.bearings, s.bearing_landmark, s.bearing_camera = w.bearing_stamps[m], w.bearings[m], w.bearing_landmark[m], w.bearing_camera[m]
    for name, cols in (("prior", ("prior_stamps", "prior_poses", "prior_sensor")), ("inertial", ("inertial_stamps", "inertial_measurements"))):
        n = len(getattr(w, cols[0]))
        lo, hi = (n * rank) // world, (n * (rank + 1)) // world
        for c in cols:
            setattr(s, c, getattr(w, c)[lo:hi])
    return s
