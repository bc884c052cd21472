// factors.hpp — per-residual factor algebra on the device, one residual block per lane.
//
// Replaces, for a whole table at once, what the reference does per ceres residual block:
//   VisualPixelEvaluator<SE3>::evaluate      /root/reference/internal/hyper/optimizers/evaluators/pixel.cpp:16-146
//   VisualBearingEvaluator<SE3>::evaluate    .../evaluators/bearing.cpp:14-79  (+ EXTERNAL AngularMetric, optimizer.cpp:192)
//   ManifoldEvaluator<SE3>::evaluate         .../evaluators/manifold.cpp:12-61 (+ EXTERNAL ManifoldMetric, optimizer.cpp:237)
//   ExteroceptiveCost::Evaluate              .../ceres/costs/exteroceptive.cpp:101-160 (metric, J_w = J_m J_e)
//   Ceres' PlusJacobian projection and loss corrector
// The chain groupPlus -> groupInverse -> vectorPlus with 6x6 Jacobian products (bearing.cpp:62-75) collapses to the
// closed form of SURVEY.md A.2c:   p_s = R_sw (p_w - p_ws),
//   d p_s / d theta_wb = R_sw hat(p_w - p_wb),  d p_s / d p_wb = -R_sw,  d p_s / d p_w = R_sw,
// and Ceres-local control-point columns  J_rot_j = 2 (J_proj R_sw hat(v)) G_j,  J_trans_j = -B_j J_proj R_sw.
#pragma once
#include "device_spline.hpp"
#include "problem.hpp"

namespace hs {
using namespace hsd;

constexpr double kHuberPixel = 0.5;       // optimizer.cpp:226
constexpr double kHuberBearing = 1.6e-3;  // optimizer.cpp:204
constexpr double kScaleInertial = 1.6e-5; // optimizer.cpp:267

template <int K>
struct VisualOut {
  double r[2];
  double Jl[6];          // 2 x 3
  double Jp[2 * 6 * K];  // 2 x 6K
  double cost;
};

/// Projection chain of one visual residual given the sensor-frame point: residual rows and d r / d p_s (2 x 3).
/// type 0: pixel (pixel.cpp:86-99 ProjectToPlane -> radtan distort -> denormalize; CartesianMetric),
/// type 1: bearing (bearing.cpp:77 returns p_s; AngularMetric atan2(|p x b|, p . b), second row unused).
HSD void visual_measure(int type, V3 ps, const double* cam, const double* meas, bool jac, double* r, double* Jps) {
  if (type == 0) {
    const double cx = cam[7], cy = cam[8], fx = cam[9], fy = cam[10];
    const double k1 = cam[11], k2 = cam[12], p1 = cam[13], p2 = cam[14];
    const double iz = 1.0 / ps.z;
    const double x = ps.x * iz, y = ps.y * iz;
    const double r2 = x * x + y * y, r4 = r2 * r2;
    const double rad = 1.0 + k1 * r2 + k2 * r4;
    const double xd = x * rad + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
    const double yd = y * rad + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
    r[0] = cx + fx * xd - meas[0];
    r[1] = cy + fy * yd - meas[1];
    if (jac) {
      const double drx = 2.0 * k1 * x + 4.0 * k2 * r2 * x, dry = 2.0 * k1 * y + 4.0 * k2 * r2 * y;
      const double d00 = fx * (rad + x * drx + 2.0 * p1 * y + 6.0 * p2 * x);
      const double d01 = fx * (x * dry + 2.0 * p1 * x + 2.0 * p2 * y);
      const double d10 = fy * (y * drx + 2.0 * p1 * x + 2.0 * p2 * y);
      const double d11 = fy * (rad + y * dry + 6.0 * p1 * y + 2.0 * p2 * x);
      // d n / d p_s = [[iz, 0, -x iz], [0, iz, -y iz]]
      Jps[0] = d00 * iz, Jps[1] = d01 * iz, Jps[2] = -(d00 * x + d01 * y) * iz;
      Jps[3] = d10 * iz, Jps[4] = d11 * iz, Jps[5] = -(d10 * x + d11 * y) * iz;
    }
  } else {
    const V3 b = V3{meas[0], meas[1], meas[2]};
    const V3 c = cross(ps, b);
    const double n = sqrt(dot(c, c)), d = dot(ps, b);
    r[0] = atan2(n, d);
    r[1] = 0.0;
    if (jac) {
      Jps[3] = Jps[4] = Jps[5] = 0.0;
      if (n > 0.0) {
        const V3 bxc = cross(b, c);
        const double den = 1.0 / (n * n + d * d), dn = d / n;
        Jps[0] = (dn * bxc.x - n * b.x) * den, Jps[1] = (dn * bxc.y - n * b.y) * den, Jps[2] = (dn * bxc.z - n * b.z) * den;
      } else {
        Jps[0] = Jps[1] = Jps[2] = 0.0;
      }
    }
  }
}

/// Sensor-frame point of landmark p_w seen from pose (q_wb, p_wb) through extrinsics cam[0..6].
HSD V3 to_sensor(Quat q_wb, V3 p_wb, const double* cam, V3 p_w, M3* R_sw_out, V3* v_out) {
  const M3 R_wb = qmat(q_wb);
  const M3 R_bs = qmat(Quat{cam[0], cam[1], cam[2], cam[3]});
  const V3 v = p_w - p_wb;
  const V3 vb = mul_t(R_wb, v);
  const V3 ps = mul_t(R_bs, vb - V3{cam[4], cam[5], cam[6]});
  if (R_sw_out) *R_sw_out = mul_tn(R_bs, transpose(R_wb));
  if (v_out) *v_out = v;
  return ps;
}

/// Value-only visual residual cost 0.5*rho(|r|^2) (residual-only branch, exteroceptive.cpp:104-122).
template <int K>
HSD double visual_cost(const Tables& T, const double* cps, const double* lms, int q) {
  const int info = T.v_info[q];
  const int type = info >> 16, camid = info & 0xffff;
  const double* cam = T.cam + kCamStride * camid;
  double u;
  segment_of(T.v_stamp[q], T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dl[1], ddl[1];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dl, ddl, 0);
  Quat qw;
  V3 pw;
  spline_pose<K>(cps + 8 * T.v_first[q], lam, &qw, &pw);
  const double* l = lms + 3 * T.v_lm[q];
  const V3 ps = to_sensor(qw, pw, cam, V3{l[0], l[1], l[2]}, nullptr, nullptr);
  double r[2], dummy[6];
  visual_measure(type, ps, cam, T.v_meas + 3 * q, false, r, dummy);
  const double s = r[0] * r[0] + r[1] * r[1];
  double sr;
  return 0.5 * loss_huber(s, type == 0 ? kHuberPixel : kHuberBearing, &sr);
}

/// Full linearisation of visual residual q (landmark-major index) in Ceres-local coordinates.
template <int K>
HSD void visual_linearize(const Tables& T, const double* cps, int q, bool robustify, VisualOut<K>* o) {
  const int info = T.v_info[q];
  const int type = info >> 16, camid = info & 0xffff;
  const double* cam = T.cam + kCamStride * camid;
  const int first = T.v_first[q];
  double u;
  segment_of(T.v_stamp[q], T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dl[1], ddl[1];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dl, ddl, 0);
  Quat qw;
  V3 pw;
  M3 G[K];
  spline_pose_jac<K>(cps + 8 * first, lam, &qw, &pw, G);
  const int lmid = T.v_lm[q];
  const double* l = T.lm + 3 * lmid;
  M3 R_sw;
  V3 v;
  const V3 ps = to_sensor(qw, pw, cam, V3{l[0], l[1], l[2]}, &R_sw, &v);
  double Jps[6];
  visual_measure(type, ps, cam, T.v_meas + 3 * q, true, o->r, Jps);
  const double s = o->r[0] * o->r[0] + o->r[1] * o->r[1];
  double sr;
  o->cost = 0.5 * loss_huber(s, type == 0 ? kHuberPixel : kHuberBearing, &sr);
  if (!robustify) sr = 1.0;
  o->r[0] *= sr, o->r[1] *= sr;
  // A = sr * Jps * R_sw (2x3);  M = A * hat(v)
  double A[6], Mh[6];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) A[3 * i + j] = sr * (Jps[3 * i] * R_sw.m[j] + Jps[3 * i + 1] * R_sw.m[3 + j] + Jps[3 * i + 2] * R_sw.m[6 + j]);
    // row * hat(v) = (row x v)^T ... (a^T hat(v))_j : a x v with sign: a^T hat(v) = (hat(v)^T a)^T = -(v x a)^T = (a x v)^T
    const V3 a = V3{A[3 * i], A[3 * i + 1], A[3 * i + 2]};
    const V3 axv = cross(a, v);
    Mh[3 * i] = axv.x, Mh[3 * i + 1] = axv.y, Mh[3 * i + 2] = axv.z;
  }
  const bool lm_free = !T.lm_const[lmid];
#pragma unroll
  for (int i = 0; i < 6; ++i) o->Jl[i] = lm_free ? A[i] : 0.0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const bool frozen = T.cp_const[first + j] != 0;
    const double Bj = lam[j] - (j + 1 < K ? lam[j + 1] : 0.0);
    const bool rot_free = !frozen && !T.sp.rot_const, tr_free = !frozen && !T.sp.trans_const;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double jr = 2.0 * (Mh[3 * i] * G[j].m[c] + Mh[3 * i + 1] * G[j].m[3 + c] + Mh[3 * i + 2] * G[j].m[6 + c]);
        o->Jp[i * 6 * K + 6 * j + c] = rot_free ? jr : 0.0;
        o->Jp[i * 6 * K + 6 * j + 3 + c] = tr_free ? -Bj * A[3 * i + c] : 0.0;
      }
    }
  }
}

// ---- pose prior (manifold.cpp:12-61 + ManifoldMetric<SE3>: r = [Log(R_m^T R_ws) ; p_ws - p_m]) ------------------
template <int K>
struct PriorOut {
  double r[6];
  double Jp[6 * 6 * K];
  double cost;
};

HSD void prior_residual(Quat qw, V3 pw, const double* T_bs, const double* meas, double* r, V3* rot_out, M3* R_ws_out, V3* Rt_out) {
  const Quat q_ws = qmul(qw, Quat{T_bs[0], T_bs[1], T_bs[2], T_bs[3]});
  const M3 R_wb = qmat(qw);
  const V3 Rt = mul(R_wb, V3{T_bs[4], T_bs[5], T_bs[6]});
  const V3 rot = so3_log(qmul(qconj(Quat{meas[0], meas[1], meas[2], meas[3]}), q_ws));
  r[0] = rot.x, r[1] = rot.y, r[2] = rot.z;
  r[3] = Rt.x + pw.x - meas[4], r[4] = Rt.y + pw.y - meas[5], r[5] = Rt.z + pw.z - meas[6];
  if (rot_out) *rot_out = rot;
  if (R_ws_out) *R_ws_out = qmat(q_ws);
  if (Rt_out) *Rt_out = Rt;
}

template <int K>
HSD double prior_cost(const Tables& T, const double* cps, int i) {
  double u;
  segment_of(T.p_stamp[i], T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dl[1], ddl[1];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dl, ddl, 0);
  Quat qw;
  V3 pw;
  spline_pose<K>(cps + 8 * T.p_first[i], lam, &qw, &pw);
  double r[6];
  prior_residual(qw, pw, T.sensor + 8 * T.p_sensor[i], T.p_meas + 7 * i, r, nullptr, nullptr, nullptr);
  double s = 0;
#pragma unroll
  for (int c = 0; c < 6; ++c) s += r[c] * r[c];
  return 0.5 * s;  // no loss (optimizer.cpp:250)
}

template <int K>
HSD void prior_linearize(const Tables& T, const double* cps, int i, PriorOut<K>* o) {
  const int first = T.p_first[i];
  double u;
  segment_of(T.p_stamp[i], T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dl[1], ddl[1];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dl, ddl, 0);
  Quat qw;
  V3 pw;
  M3 G[K];
  spline_pose_jac<K>(cps + 8 * first, lam, &qw, &pw, G);
  V3 rot, Rt;
  M3 R_ws;
  prior_residual(qw, pw, T.sensor + 8 * T.p_sensor[i], T.p_meas + 7 * i, o->r, &rot, &R_ws, &Rt);
  double s = 0;
#pragma unroll
  for (int c = 0; c < 6; ++c) s += o->r[c] * o->r[c];
  o->cost = 0.5 * s;
  // d r_rot / d theta = J_r^-1(r_rot) R_ws^T ;  d r_p / d theta = -hat(R_wb t_bs)
  const So3Coef sc = so3_coef(dot(rot, rot), true);
  const M3 Jri = rodrigues_poly(rot, 0.5, sc.D);
  const M3 Arot = mul_nt(Jri, R_ws);  // J_r^-1 * R_ws^T
  const M3 Apos = scale(-1.0, hat(Rt));
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const bool frozen = T.cp_const[first + j] != 0;
    const bool rot_free = !frozen && !T.sp.rot_const, tr_free = !frozen && !T.sp.trans_const;
    const double Bj = lam[j] - (j + 1 < K ? lam[j + 1] : 0.0);
    const M3 Jr = scale(2.0, mul(Arot, G[j]));
    const M3 Jp = scale(2.0, mul(Apos, G[j]));
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o->Jp[r * 6 * K + 6 * j + c] = rot_free ? Jr.m[3 * r + c] : 0.0;
        o->Jp[r * 6 * K + 6 * j + 3 + c] = 0.0;
        o->Jp[(3 + r) * 6 * K + 6 * j + c] = rot_free ? Jp.m[3 * r + c] : 0.0;
        o->Jp[(3 + r) * 6 * K + 6 * j + 3 + c] = (tr_free && r == c) ? Bj : 0.0;
      }
  }
}

}  // namespace hs
