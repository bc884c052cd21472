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

/// Inputs of one visual residual block (whoever calls visual_core decides where they come from: the tables in HBM, or LDS copies
/// a workgroup staged for its chunk).
struct VisualIn {
  double stamp, meas[3];
  int type, first;
  const double* cam;  // 16 doubles: T_bs(7) | cx cy fx fy | k1 k2 p1 p2
  double lm[3];       // landmark position
};
HSD VisualIn visual_input(const Tables& T, int q, const double* lms) {
  VisualIn in;
  const int info = T.v_info[q];
  in.type = info >> 16, in.cam = T.cam + kCamStride * (info & 0xffff);
  in.first = T.v_first[q], in.stamp = T.v_stamp[q];
  in.meas[0] = T.v_meas[3 * q], in.meas[1] = T.v_meas[3 * q + 1], in.meas[2] = T.v_meas[3 * q + 2];
  const double* l = (lms ? lms : T.lm) + 3 * T.v_lm[q];
  in.lm[0] = l[0], in.lm[1] = l[1], in.lm[2] = l[2];
  return in;
}

/// Shared front half of the visual linearisation: spline pose + rotation Jacobian blocks, projection chain, loss corrector.
///   A = sr * J_proj * R_sw (2 x 3) = d r / d p_w,   Mh = A * hat(p_w - p_wb) (2 x 3).
template <int K>
struct VisualCore {
  double r[2];     // corrected residual rows
  double A[6], Mh[6];
  M3 G[K];         // d theta / d phi_j
  double lam[K];   // cumulative basis weights
  double cost;
};

/// rel (optional): relative rotations of consecutive control points, indexed like `cps` (rel[j] = pair j -> j + 1, device_spline.hpp RelPre).
template <int K>
HSD void visual_core(const Tables& T, const double* cps, const VisualIn& in, bool robustify, VisualCore<K>* o, const RelPre* rel = nullptr) {
  double u;
  segment_of(in.stamp, T.sp.t0, T.sp.dt, K, &u);
  double dl[1], ddl[1];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, o->lam, dl, ddl, 0);
  Quat qw;
  V3 pw;
  if (rel)
    spline_pose_jac_pre<K>(cps + 8 * in.first, rel + in.first, o->lam, &qw, &pw, o->G);
  else
    spline_pose_jac<K>(cps + 8 * in.first, o->lam, &qw, &pw, o->G);
  M3 R_sw;
  V3 v;
  const V3 ps = to_sensor(qw, pw, in.cam, V3{in.lm[0], in.lm[1], in.lm[2]}, &R_sw, &v);
  double Jps[6];
  visual_measure(in.type, ps, in.cam, in.meas, true, o->r, Jps);
  const double s = o->r[0] * o->r[0] + o->r[1] * o->r[1];
  double sr;
  o->cost = 0.5 * loss_huber(s, in.type == 0 ? kHuberPixel : kHuberBearing, &sr);
  if (!robustify) sr = 1.0;
  o->r[0] *= sr, o->r[1] *= sr;
  // A = sr * Jps * R_sw (2x3);  M = A * hat(v)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) o->A[3 * i + j] = sr * (Jps[3 * i] * R_sw.m[j] + Jps[3 * i + 1] * R_sw.m[3 + j] + Jps[3 * i + 2] * R_sw.m[6 + j]);
    // row * hat(v) = (row x v)^T ... (a^T hat(v))_j : a x v with sign: a^T hat(v) = (hat(v)^T a)^T = -(v x a)^T = (a x v)^T
    const V3 a = V3{o->A[3 * i], o->A[3 * i + 1], o->A[3 * i + 2]};
    const V3 axv = cross(a, v);
    o->Mh[3 * i] = axv.x, o->Mh[3 * i + 1] = axv.y, o->Mh[3 * i + 2] = axv.z;
  }
}

/// Value-only cost 0.5 * rho(|r|^2) of a visual residual block from preloaded inputs (residual-only branch, exteroceptive.cpp:104-122):
/// what visual_cost computes, for callers that staged the inputs themselves.
template <int K>
HSD double visual_cost_in(const Tables& T, const double* cps, const VisualIn& in, const RelPre* rel = nullptr /* indexed like cps: pair (j, j + 1) at rel[j] */) {
  double u;
  segment_of(in.stamp, T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dl[1], ddl[1];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dl, ddl, 0);
  Quat qw;
  V3 pw;
  if (rel)
    spline_pose_pre<K>(cps + 8 * in.first, rel + in.first, lam, &qw, &pw);
  else
    spline_pose<K>(cps + 8 * in.first, lam, &qw, &pw);
  const V3 ps = to_sensor(qw, pw, in.cam, V3{in.lm[0], in.lm[1], in.lm[2]}, nullptr, nullptr);
  double r[2], dummy[6];
  visual_measure(in.type, ps, in.cam, in.meas, false, r, dummy);
  const double s = r[0] * r[0] + r[1] * r[1];
  double sr;
  return 0.5 * loss_huber(s, in.type == 0 ? kHuberPixel : kHuberBearing, &sr);
}

/// Full linearisation of visual residual q (landmark-major index) in Ceres-local coordinates.
template <int K>
HSD void visual_linearize(const Tables& T, const double* cps, int q, bool robustify, VisualOut<K>* o, const double* lms = nullptr) {
  const VisualIn in = visual_input(T, q, lms);
  VisualCore<K> c;
  visual_core<K>(T, cps, in, robustify, &c);
  o->r[0] = c.r[0], o->r[1] = c.r[1], o->cost = c.cost;
  const bool lm_free = !T.lm_const[T.v_lm[q]];
#pragma unroll
  for (int i = 0; i < 6; ++i) o->Jl[i] = lm_free ? c.A[i] : 0.0;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const bool frozen = T.cp_const[in.first + j] != 0;
    const double Bj = c.lam[j] - (j + 1 < K ? c.lam[j + 1] : 0.0);
    const bool rot_free = !frozen && !T.sp.rot_const, tr_free = !frozen && !T.sp.trans_const;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const double jr = 2.0 * (c.Mh[3 * i] * c.G[j].m[cc] + c.Mh[3 * i + 1] * c.G[j].m[3 + cc] + c.Mh[3 * i + 2] * c.G[j].m[6 + cc]);
        o->Jp[i * 6 * K + 6 * j + cc] = rot_free ? jr : 0.0;
        o->Jp[i * 6 * K + 6 * j + 3 + cc] = tr_free ? -Bj * c.A[3 * i + cc] : 0.0;
      }
    }
  }
}

/// Compact record of a visual residual (the fused build keeps a workgroup's records in LDS, kernels_build.hpp). The translation columns of
/// the state Jacobian are -B_j A and the landmark Jacobian is A itself, so the 2 x 6K state block is stored as its rotation half plus the
/// K scalars B_j, laid out for 16-byte LDS accesses:
///   [r(2) | A(2 x 3) | per control point j: J_rot row 0 (3), B_eff_j, J_rot row 1 (3), B_eff_j]  = 8 + 8K doubles   (full record: 8 + 12K).
/// B_eff_j = 0 for a constant control point / constant translations, the rotation columns of such a point are stored as zeros. A is
/// stored unmasked: a constant landmark's flag is applied by the consumers (its W, H_ll and b_l are zero, its translation columns are not).
/// cp_frozen: constancy flags of the control points, indexed like `cps` (absolute index).
template <int K>
constexpr int compact_record() { return 8 + 8 * K; }

template <int K>
HSD double visual_linearize_compact(const Tables& T, const double* cps, const RelPre* rel, const uint8_t* cp_frozen, const VisualIn& in, bool robustify,
                                    double* rec) {
  VisualCore<K> c;
  visual_core<K>(T, cps, in, robustify, &c, rel);
  *reinterpret_cast<double2*>(rec) = make_double2(c.r[0], c.r[1]);
#pragma unroll
  for (int i = 0; i < 6; i += 2) *reinterpret_cast<double2*>(rec + 2 + i) = make_double2(c.A[i], c.A[i + 1]);
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const bool frozen = cp_frozen[in.first + j] != 0;
    const double Bj = c.lam[j] - (j + 1 < K ? c.lam[j + 1] : 0.0);
    const bool rot_free = !frozen && !T.sp.rot_const, tr_free = !frozen && !T.sp.trans_const;
    const double be = tr_free ? Bj : 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      double jr[3];
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) {
        const double v = 2.0 * (c.Mh[3 * i] * c.G[j].m[cc] + c.Mh[3 * i + 1] * c.G[j].m[3 + cc] + c.Mh[3 * i + 2] * c.G[j].m[6 + cc]);
        jr[cc] = rot_free ? v : 0.0;
      }
      double* blk = rec + 8 + 8 * j + 4 * i;
      *reinterpret_cast<double2*>(blk) = make_double2(jr[0], jr[1]);
      *reinterpret_cast<double2*>(blk + 2) = make_double2(jr[2], be);
    }
  }
  return c.cost;
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

namespace hs {

// ---- inertial factor (inertial.cpp:13-205; CartesianMetric<6>; ScaledLoss(1.6e-5), optimizer.cpp:267-268) ---------------
// prediction = [ I_g R_sb w_b + S_g a_m + b_g ;  I_a R_sb a_m + b_a ],   a_m[i] = a_i[i] + F_a.row(i) (X_a.col(i) + t_bs),
// a_i = R_bw (p'' - g),  F_a = hat(w)^2 + hat(alpha).
// Jacobian, two forms selected by Tables::inertial_literal (hs_set_inertial_jacobian):
//   as written upstream (default): linear rows of the rotational state columns carry I_g (inertial.cpp:136,142,148), their lever arm
//     is t_bs alone, and the S_g terms of the state and gravity columns are absent (:134-153,198);
//   exact: the derivative of the prediction (I_a, per-row lever arms X_a.col(i) + t_bs, S_g terms kept).
// Identical for I_g = I_a, S_g = 0, X_a = 0 — wherever the reference is exercised (settings.yaml:87-96).
struct ImuParams {
  double T_bs[7], i_g[6], i_a[6], S_g[9], X_a[9];
};

template <int K, int KB>
struct InertialOut {
  double r[6];
  double* Jp;            // 6 x 6K local state Jacobian, written as it is produced: the caller points it at the record in memory (held
                         // in registers, its 36 K doubles alone exceed the register file at K = 6)
  double wg[KB], wa[KB]; // bias-spline weights (d r_ang / d b_g,j = wg[j] I, d r_lin / d b_a,j = wa[j] I)
  double Jg[12];         // 6 x 2 gravity (SphereManifold<3> tangent)
  double cost;
};

HSD M3 lower_tri(const double* c) { return M3{{c[0], 0, 0, c[3], c[1], 0, c[4], c[5], c[2]}}; }
HSD M3 colmajor3(const double* c) { return M3{{c[0], c[3], c[6], c[1], c[4], c[7], c[2], c[5], c[8]}}; }

template <int KB>
HSD V3 bias_value(const BasisCoef& bb, const double* cps, double u, double* wts) {
  double lam[KB], dl[1], ddl[1];
  basis_weights<KB>(bb, u, 1.0, lam, dl, ddl, 0);
  V3 b = V3{0, 0, 0};
#pragma unroll
  for (int j = 0; j < KB; ++j) {
    const double Bj = lam[j] - (j + 1 < KB ? lam[j + 1] : 0.0);
    wts[j] = Bj;
    b = b + Bj * V3{cps[4 * j], cps[4 * j + 1], cps[4 * j + 2]};
  }
  return b;
}

/// Shared by value-only and full paths. JAC selects the Jacobian outputs.
template <int K, int KB, bool JAC>
HSD void inertial_evaluate(const Tables& T, const double* cps, const double* bias_g, const double* bias_a, const double* gravity, int i,
                           bool robustify, InertialOut<K, KB>* o) {
  const ImuParams& P = *T.imu;
  const int first = T.i_first[i], fb = T.i_first_bias[i];
  double u;
  segment_of(T.i_stamp[i], T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dlam[K], ddlam[K];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dlam, ddlam, 2);
  SplineFull<K> S;
  spline_full<K, JAC>(cps + 8 * first, lam, dlam, ddlam, &S);
  double ub;
  segment_of(T.i_stamp[i], T.bias_t0, T.bias_dt, KB, &ub);
  const V3 b_g = bias_value<KB>(T.bias_basis, bias_g + 4 * fb, ub, o->wg);
  const V3 b_a = bias_value<KB>(T.bias_basis, bias_a + 4 * fb, ub, o->wa);

  const M3 R = qmat(S.q);                       // R_wb
  const M3 R_bs = qmat(Quat{P.T_bs[0], P.T_bs[1], P.T_bs[2], P.T_bs[3]});
  const V3 t_bs = V3{P.T_bs[4], P.T_bs[5], P.T_bs[6]};
  const M3 I_g = lower_tri(P.i_g), I_a = lower_tri(P.i_a), S_g = colmajor3(P.S_g), X_a = colmajor3(P.X_a);
  const V3 g = V3{gravity[0], gravity[1], gravity[2]};
  const V3 a_i = mul_t(R, S.a - g);             // R_bw (p'' - g)
  const M3 wx = hat(S.w);
  const M3 F_a = add(mul(wx, wx), hat(S.al));
  V3 lever[3];
  double am[3];
  const double ai[3] = {a_i.x, a_i.y, a_i.z};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    lever[r] = V3{X_a.m[r] + t_bs.x, X_a.m[3 + r] + t_bs.y, X_a.m[6 + r] + t_bs.z};  // X_a.col(r) + t_bs
    am[r] = ai[r] + F_a.m[3 * r] * lever[r].x + F_a.m[3 * r + 1] * lever[r].y + F_a.m[3 * r + 2] * lever[r].z;
  }
  const V3 a_m = V3{am[0], am[1], am[2]};
  const M3 IgRsb = mul_nt(I_g, R_bs), IaRsb = mul_nt(I_a, R_bs);  // I * R_sb = I * R_bs^T
  const V3 ang = mul(IgRsb, S.w) + mul(S_g, a_m) + b_g;
  const V3 lin = mul(IaRsb, a_m) + b_a;
  const double* m = T.i_meas + 6 * i;
  o->r[0] = ang.x - m[0], o->r[1] = ang.y - m[1], o->r[2] = ang.z - m[2];
  o->r[3] = lin.x - m[3], o->r[4] = lin.y - m[4], o->r[5] = lin.z - m[5];
  double s = 0;
#pragma unroll
  for (int c = 0; c < 6; ++c) s += o->r[c] * o->r[c];
  o->cost = 0.5 * kScaleInertial * s;
  const double sr = robustify ? sqrt(kScaleInertial) : 1.0;
#pragma unroll
  for (int c = 0; c < 6; ++c) o->r[c] *= sr;
  if (!JAC) return;

  // d a_m / d w (L_w) and d a_m / d alpha (L_al) with per-row lever arms
  const bool lit = T.inertial_literal != 0;
  const M3 S_gJ = lit ? zero3() : S_g;        // S_g as it enters the state / gravity columns
  const M3 IxRsb = lit ? IgRsb : IaRsb;       // (:136,142,148) vs the matrix of the prediction
  M3 L_w, L_al;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const M3 lx = hat(lit ? t_bs : lever[r]);
    const M3 mw = sub(mul(lx, wx), scale(2.0, mul(wx, lx)));  // -(2 wx lx - lx wx)
#pragma unroll
    for (int c = 0; c < 3; ++c) L_w.m[3 * r + c] = mw.m[3 * r + c], L_al.m[3 * r + c] = -lx.m[3 * r + c];
  }
  const M3 Rt = transpose(R);
  const M3 HaRt = mul(hat(a_i), Rt);  // hat(a_i) R_bw
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const bool frozen = T.cp_const[first + j] != 0;
    const bool rot_free = !frozen && !T.sp.rot_const, tr_free = !frozen && !T.sp.trans_const;
    // d a_m / d phi_j, d a_m / d dp_j
    const M3 dam_rot = add(mul(HaRt, S.dth[j]), add(mul(L_w, S.dw[j]), mul(L_al, S.dal[j])));
    const M3 dam_tr = scale(S.Bdd[j], Rt);
    const M3 ang_rot = add(mul(IgRsb, S.dw[j]), mul(S_gJ, dam_rot));
    const M3 ang_tr = mul(S_gJ, dam_tr);
    const M3 lin_rot = mul(IxRsb, dam_rot);
    const M3 lin_tr = mul(IaRsb, dam_tr);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        o->Jp[r * 6 * K + 6 * j + c] = rot_free ? 2.0 * sr * ang_rot.m[3 * r + c] : 0.0;
        o->Jp[r * 6 * K + 6 * j + 3 + c] = tr_free ? sr * ang_tr.m[3 * r + c] : 0.0;
        o->Jp[(3 + r) * 6 * K + 6 * j + c] = rot_free ? 2.0 * sr * lin_rot.m[3 * r + c] : 0.0;
        o->Jp[(3 + r) * 6 * K + 6 * j + 3 + c] = tr_free ? sr * lin_tr.m[3 * r + c] : 0.0;
      }
  }
  const double bscale = T.bias_const ? 0.0 : sr;
#pragma unroll
  for (int j = 0; j < KB; ++j) o->wg[j] *= bscale, o->wa[j] *= bscale;
  // gravity: d a_m / d g = -R_bw, through the SphereManifold<3> tangent basis
  double Pg[6];
  sphere_plus_jacobian(gravity, Pg);
  const M3 nRt = scale(-1.0, Rt);
  const M3 ang_g = mul(S_gJ, nRt), lin_g = mul(IaRsb, nRt);
  const double gs = T.gravity_const ? 0.0 : sr;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      o->Jg[2 * r + c] = gs * (ang_g.m[3 * r] * Pg[c] + ang_g.m[3 * r + 1] * Pg[2 + c] + ang_g.m[3 * r + 2] * Pg[4 + c]);
      o->Jg[2 * (3 + r) + c] = gs * (lin_g.m[3 * r] * Pg[c] + lin_g.m[3 * r + 1] * Pg[2 + c] + lin_g.m[3 * r + 2] * Pg[4 + c]);
    }
}

/// The linearisation of inertial residual i for ONE control point m of its segment (wave-uniform m: k_linearize_inertial spreads a
/// residual over K lanes of K different waves): the prediction and the residual — by every lane, they are cheap — and the 6 x 6 block of
/// the local state Jacobian that belongs to control point m, written straight into the record. m == 0 also writes the residual, the bias
/// weights and the gravity block and returns the cost (other lanes return 0). Same expressions as inertial_evaluate.
template <int K, int KB>
HSD double inertial_linearize_col(const Tables& T, const double* cps, const double* bias_g, const double* bias_a, const double* gravity, const int i,
                                  const int m, const bool robustify, double* rec) {
  const ImuParams& P = *T.imu;
  const int first = T.i_first[i], fb = T.i_first_bias[i];
  double u;
  segment_of(T.i_stamp[i], T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dlam[K], ddlam[K];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dlam, ddlam, 2);
  SplineCol<K> S;
  spline_full_col<K>(cps + 8 * first, lam, dlam, ddlam, m, &S);
  const M3 R = qmat(S.q);  // R_wb
  const M3 R_bs = qmat(Quat{P.T_bs[0], P.T_bs[1], P.T_bs[2], P.T_bs[3]});
  const V3 t_bs = V3{P.T_bs[4], P.T_bs[5], P.T_bs[6]};
  const M3 I_g = lower_tri(P.i_g), I_a = lower_tri(P.i_a), S_g = colmajor3(P.S_g), X_a = colmajor3(P.X_a);
  const V3 g = V3{gravity[0], gravity[1], gravity[2]};
  const V3 a_i = mul_t(R, S.a - g);  // R_bw (p'' - g)
  const M3 wx = hat(S.w);
  const M3 F_a = add(mul(wx, wx), hat(S.al));
  V3 lever[3];
  double am[3];
  const double ai[3] = {a_i.x, a_i.y, a_i.z};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    lever[r] = V3{X_a.m[r] + t_bs.x, X_a.m[3 + r] + t_bs.y, X_a.m[6 + r] + t_bs.z};  // X_a.col(r) + t_bs
    am[r] = ai[r] + F_a.m[3 * r] * lever[r].x + F_a.m[3 * r + 1] * lever[r].y + F_a.m[3 * r + 2] * lever[r].z;
  }
  const M3 IgRsb = mul_nt(I_g, R_bs), IaRsb = mul_nt(I_a, R_bs);  // I * R_sb = I * R_bs^T
  const double sr = robustify ? sqrt(kScaleInertial) : 1.0;
  const M3 Rt = transpose(R);
  const bool lit = T.inertial_literal != 0;
  const M3 S_gJ = lit ? zero3() : S_g;   // S_g as it enters the state / gravity columns
  const M3 IxRsb = lit ? IgRsb : IaRsb;  // (:136,142,148) vs the matrix of the prediction
  double cost = 0.0;
  if (m == 0) {
    const V3 a_m = V3{am[0], am[1], am[2]};
    double ub, wg[KB], wa[KB];
    segment_of(T.i_stamp[i], T.bias_t0, T.bias_dt, KB, &ub);
    const V3 b_g = bias_value<KB>(T.bias_basis, bias_g + 4 * fb, ub, wg);
    const V3 b_a = bias_value<KB>(T.bias_basis, bias_a + 4 * fb, ub, wa);
    const V3 ang = mul(IgRsb, S.w) + mul(S_g, a_m) + b_g;
    const V3 lin = mul(IaRsb, a_m) + b_a;
    const double* ms = T.i_meas + 6 * i;
    double r[6] = {ang.x - ms[0], ang.y - ms[1], ang.z - ms[2], lin.x - ms[3], lin.y - ms[4], lin.z - ms[5]};
    double s = 0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s += r[c] * r[c];
    cost = 0.5 * kScaleInertial * s;
#pragma unroll
    for (int c = 0; c < 6; ++c) rec[c] = r[c] * sr;
    const double bscale = T.bias_const ? 0.0 : sr;
#pragma unroll
    for (int j = 0; j < KB; ++j) rec[6 + 36 * K + j] = wg[j] * bscale, rec[6 + 36 * K + KB + j] = wa[j] * bscale;
    // gravity: d a_m / d g = -R_bw, through the SphereManifold<3> tangent basis
    double Pg[6];
    sphere_plus_jacobian(gravity, Pg);
    const M3 nRt = scale(-1.0, Rt);
    const M3 ang_g = mul(S_gJ, nRt), lin_g = mul(IaRsb, nRt);
    const double gs = T.gravity_const ? 0.0 : sr;
    double* Jg = rec + 6 + 36 * K + 2 * KB;
#pragma unroll
    for (int r2 = 0; r2 < 3; ++r2)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        Jg[2 * r2 + c] = gs * (ang_g.m[3 * r2] * Pg[c] + ang_g.m[3 * r2 + 1] * Pg[2 + c] + ang_g.m[3 * r2 + 2] * Pg[4 + c]);
        Jg[2 * (3 + r2) + c] = gs * (lin_g.m[3 * r2] * Pg[c] + lin_g.m[3 * r2 + 1] * Pg[2 + c] + lin_g.m[3 * r2 + 2] * Pg[4 + c]);
      }
  }
  // d a_m / d w (L_w) and d a_m / d alpha (L_al) with per-row lever arms
  M3 L_w, L_al;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const M3 lx = hat(lit ? t_bs : lever[r]);
    const M3 mw = sub(mul(lx, wx), scale(2.0, mul(wx, lx)));  // -(2 wx lx - lx wx)
#pragma unroll
    for (int c = 0; c < 3; ++c) L_w.m[3 * r + c] = mw.m[3 * r + c], L_al.m[3 * r + c] = -lx.m[3 * r + c];
  }
  const M3 HaRt = mul(hat(a_i), Rt);  // hat(a_i) R_bw
  const bool frozen = T.cp_const[first + m] != 0;
  const bool rot_free = !frozen && !T.sp.rot_const, tr_free = !frozen && !T.sp.trans_const;
  const M3 dam_rot = add(mul(HaRt, S.dth), add(mul(L_w, S.dw), mul(L_al, S.dal)));  // d a_m / d phi_m
  const M3 dam_tr = scale(S.Bdd_m, Rt);                                               // d a_m / d dp_m
  const M3 ang_rot = add(mul(IgRsb, S.dw), mul(S_gJ, dam_rot));
  const M3 ang_tr = mul(S_gJ, dam_tr);
  const M3 lin_rot = mul(IxRsb, dam_rot);
  const M3 lin_tr = mul(IaRsb, dam_tr);
  double* Jp = rec + 6;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Jp[r * 6 * K + 6 * m + c] = rot_free ? 2.0 * sr * ang_rot.m[3 * r + c] : 0.0;
      Jp[r * 6 * K + 6 * m + 3 + c] = tr_free ? sr * ang_tr.m[3 * r + c] : 0.0;
      Jp[(3 + r) * 6 * K + 6 * m + c] = rot_free ? 2.0 * sr * lin_rot.m[3 * r + c] : 0.0;
      Jp[(3 + r) * 6 * K + 6 * m + 3 + c] = tr_free ? sr * lin_tr.m[3 * r + c] : 0.0;
    }
  return cost;
}

}  // namespace hs
