// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_math.hpp header). PARITY UNPINNED.
//
// hs_factors.hpp: CPU restatement of the per-residual path
//   ExteroceptiveCost::update / ::Evaluate            (internal/hyper/optimizers/ceres/costs/exteroceptive.cpp:25-160)
//   Visual{Bearing,Pixel}Evaluator, ManifoldEvaluator, InertialEvaluator
//                                                    (internal/hyper/optimizers/evaluators/{bearing,pixel,manifold,inertial}.cpp)
//   metrics (EXTERNAL Angular/Cartesian/ManifoldMetric; call sites optimizer.cpp:192,215,237,256)
//   Ceres manifolds Plus / PlusJacobian              (include/hyper/optimizers/ceres/manifolds/**)
// following the reference structure: evaluator -> ambient J_e -> metric J_m -> J_w = J_m J_e -> row-major per-block
// scatter -> (Ceres) local Jacobian = J_block * PlusJacobian.
//
// Tangent conventions chosen for the EXTERNAL Lie-group pieces (documented in DESIGN.md):
//   spline value T_wb, T_ws, T_sw : rotation perturbed on the left (world frame), translation additively
//   extrinsics T_bs              : rotation perturbed on the right (sensor frame), translation additively
// (these are exactly the conventions under which inertial.cpp:136 and inertial.cpp:155-161 are exact).
#pragma once
#include <algorithm>
#include <vector>

#include "hs_math.hpp"

namespace hso {

// ---------------------------------------------------------------------------------------------------------
// Dynamic row-major matrix (stand-in for Eigen DynamicJacobian).
// ---------------------------------------------------------------------------------------------------------
struct DMat {
  int rows = 0, cols = 0;
  std::vector<double> a;
  void set_zero(int r, int c) {
    rows = r, cols = c;
    a.assign(size_t(r) * c, 0.0);
  }
  double& operator()(int i, int j) { return a[size_t(i) * cols + j]; }
  const double& operator()(int i, int j) const { return a[size_t(i) * cols + j]; }
  template <int R, int C>
  void set_block(int i0, int j0, const Mat<R, C>& m) {
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) (*this)(i0 + i, j0 + j) = m(i, j);
  }
};
inline DMat dmul(const DMat& x, const DMat& y) {
  DMat z;
  z.set_zero(x.rows, y.cols);
  for (int i = 0; i < x.rows; ++i)
    for (int k = 0; k < x.cols; ++k) {
      const double v = x(i, k);
      if (v == 0.0) continue;
      for (int j = 0; j < y.cols; ++j) z(i, j) += v * y(k, j);
    }
  return z;
}

// ---------------------------------------------------------------------------------------------------------
// Residual kinds / block layout (exteroceptive.cpp:25-99, evaluators/forward.hpp:22-41).
// ---------------------------------------------------------------------------------------------------------
enum FactorType : int { kPixel = 0, kBearing = 1, kPrior = 2, kInertial = 3 };

struct Layout {
  int num_parameters = 0;
  int static_state_idx = 0, static_sensor_idx = 0, dynamic_sensor_idx = 0, static_observation_idx = 0;
  std::vector<int> offsets;
  std::vector<int> sizes;
  int num_residuals = 0;
};

// Block sizes: Stamped<SE3> = 8, SE3 = 7, camera intrinsics 4, radtan distortion 4,
// IMU {7,6,6,9,9} (inertial.cpp:35-39), Stamped<R3> bias = 4, landmark 3, gravity 3.
constexpr int kCpSize = 8, kSe3Size = 7, kIntrinsicsSize = 4, kDistortionSize = 4;
constexpr int kImuAlignSize = 6, kImuMat9 = 9, kBiasCpSize = 4, kLandmarkSize = 3, kGravitySize = 3;

/// Restates ExteroceptiveCost::update (exteroceptive.cpp:25-99): block order state || sensor || observation,
/// indices {0, n_state, n_state + |sensor.variables()|, n_state + n_sensor}, sizes, offsets (exclusive prefix sum),
/// num_residuals = metric->outputSize().
inline Layout make_layout(FactorType type, int k, int k_bias) {
  Layout L;
  std::vector<int>& s = L.sizes;
  for (int j = 0; j < k; ++j) s.push_back(kCpSize);
  const int n_state = k;
  int n_sensor_static = 0, n_sensor = 0;
  switch (type) {
    case kPixel:
    case kBearing:
      s.push_back(kSe3Size), s.push_back(kIntrinsicsSize), s.push_back(kDistortionSize);
      n_sensor_static = n_sensor = 3;
      s.push_back(kLandmarkSize);
      L.num_residuals = (type == kPixel) ? 2 : 1;
      break;
    case kPrior:
      s.push_back(kSe3Size);
      n_sensor_static = n_sensor = 1;
      L.num_residuals = 6;
      break;
    case kInertial:
      s.push_back(kSe3Size), s.push_back(kImuAlignSize), s.push_back(kImuAlignSize), s.push_back(kImuMat9), s.push_back(kImuMat9);
      n_sensor_static = 5;
      for (int j = 0; j < 2 * k_bias; ++j) s.push_back(kBiasCpSize);
      n_sensor = 5 + 2 * k_bias;
      s.push_back(kGravitySize);
      L.num_residuals = 6;
      break;
  }
  L.static_state_idx = 0;
  L.static_sensor_idx = n_state;
  L.dynamic_sensor_idx = n_state + n_sensor_static;
  L.static_observation_idx = n_state + n_sensor;
  L.offsets.assign(s.size(), 0);
  for (size_t i = 1; i < s.size(); ++i) L.offsets[i] = L.offsets[i - 1] + s[i - 1];
  L.num_parameters = 0;
  for (int v : s) L.num_parameters += v;
  return L;
}

// ---------------------------------------------------------------------------------------------------------
// Tangent -> ambient adapters (EXTERNAL SE3JacobianAdapter, call sites bearing.cpp:74, se3.cpp:15-17).
// ---------------------------------------------------------------------------------------------------------
/// 3x4: d theta / d q for R <- Exp(theta) R (left):  theta = 2 vec(dq (x) q*).
inline Mat<3, 4> quat_adapter_left(const Quat& q) {
  Mat<3, 4> A;
  const Quat qc = qconj(q);
  for (int i = 0; i < 4; ++i) {
    Quat e = {0, 0, 0, 0};
    (&e.x)[i] = 1.0;
    const Quat r = qmul(e, qc);
    A(0, i) = 2 * r.x, A(1, i) = 2 * r.y, A(2, i) = 2 * r.z;
  }
  return A;
}
/// 3x4: d phi / d q for R <- R Exp(phi) (right):  phi = 2 vec(q* (x) dq).
inline Mat<3, 4> quat_adapter_right(const Quat& q) {
  Mat<3, 4> A;
  const Quat qc = qconj(q);
  for (int i = 0; i < 4; ++i) {
    Quat e = {0, 0, 0, 0};
    (&e.x)[i] = 1.0;
    const Quat r = qmul(qc, e);
    A(0, i) = 2 * r.x, A(1, i) = 2 * r.y, A(2, i) = 2 * r.z;
  }
  return A;
}

/// Expands a Jacobian given in tangent columns [theta(3) dp(3)] at column j0 of `Jt` into 7 ambient columns
/// [q(4) p(3)] of J_e at column c0.
inline void scatter_se3(const DMat& Jt, int j0, const Mat<3, 4>& A, DMat* Je, int c0) {
  for (int r = 0; r < Jt.rows; ++r) {
    for (int c = 0; c < 4; ++c) {
      double s = 0;
      for (int m = 0; m < 3; ++m) s += Jt(r, j0 + m) * A(m, c);
      (*Je)(r, c0 + c) = s;
    }
    for (int c = 0; c < 3; ++c) (*Je)(r, c0 + 4 + c) = Jt(r, j0 + 3 + c);
  }
}

// ---------------------------------------------------------------------------------------------------------
// State evaluation in the shape the evaluators consume: derivatives[d] and jacobians[d] (6 x 6k tangent columns,
// rows [angular; linear], Tangent<SE3> layout of inertial.cpp:135-150).
//   d=0 rows: [theta ; p]                 (left/additive tangent of T_wb)
//   d=1 rows: [w_b ; R^T v]               (body-frame twist; linear rows' rotation dependence NOT included, see below)
//   d=2 rows: [alpha_b ; R^T a]           (ditto: carried by J_value, inertial.cpp:136 -- SURVEY.md §8a conventions)
// ---------------------------------------------------------------------------------------------------------
struct StateResult {
  SplineValue s;
  DMat J[3];
};

inline void cp_u_invdt(const double* const* cps, int k, double stamp, double* u, double* inv_dt) {
  const int i = (k - 1) / 2;
  const double ti = cps[i][7];
  const double dt = cps[i + 1][7] - ti;
  *inv_dt = 1.0 / dt;
  *u = (stamp - ti) / dt;
}

inline void state_evaluate(const Basis& basis, const double* const* cps, double stamp, int derivative, bool jac, StateResult* out) {
  const int k = basis.k;
  double u, inv_dt;
  cp_u_invdt(cps, k, stamp, &u, &inv_dt);
  spline_evaluate(basis, cps, u, inv_dt, derivative, jac, &out->s);
  if (!jac) return;
  const SplineValue& s = out->s;
  const M3 Rt = T(s.R);
  for (int d = 0; d <= derivative; ++d) out->J[d].set_zero(6, 6 * k);
  for (int j = 0; j < k; ++j) {
    out->J[0].set_block(0, 6 * j, s.dth[j]);
    out->J[0].set_block(3, 6 * j + 3, s.B[j] * M3::identity());
    if (derivative >= 1) {
      out->J[1].set_block(0, 6 * j, s.dw[j]);
      out->J[1].set_block(3, 6 * j + 3, s.Bd[j] * Rt);
    }
    if (derivative >= 2) {
      out->J[2].set_block(0, 6 * j, s.dal[j]);
      out->J[2].set_block(3, 6 * j + 3, s.Bdd[j] * Rt);
    }
  }
}

/// Writes `Jt` (n x 6k, tangent columns) into J_e's k state blocks (8 ambient columns each; stamp column zero).
inline void scatter_state(const DMat& Jt, const double* const* cps, int k, const Layout& L, DMat* Je) {
  for (int j = 0; j < k; ++j) {
    const Quat q = {cps[j][0], cps[j][1], cps[j][2], cps[j][3]};
    scatter_se3(Jt, 6 * j, quat_adapter_left(q), Je, L.offsets[L.static_state_idx + j]);
  }
}

inline bool any_non_null(const double* const* p_js, int begin, int end) {
  if (!p_js) return false;
  for (int i = begin; i < end; ++i)
    if (p_js[i]) return true;
  return false;
}

// ---------------------------------------------------------------------------------------------------------
// SE3 group operations with Jacobians (EXTERNAL groupPlus / groupInverse / vectorPlus; bearing.cpp:62-69).
// ---------------------------------------------------------------------------------------------------------
struct Pose {
  Quat q;
  M3 R;
  V3 p;
};
inline Pose make_pose(const double* raw) {
  Pose T;
  T.q = {raw[0], raw[1], raw[2], raw[3]};
  T.R = qmat(T.q);
  T.p = v3(raw[4], raw[5], raw[6]);
  return T;
}
/// T_ws = T_wb o T_bs.  J_this: d tangent(T_ws)/d tangent(T_wb) (left/additive); J_other: .../d tangent(T_bs) (right/additive).
inline Pose group_plus(const Pose& a, const Pose& b, Mat<6, 6>* J_this, Mat<6, 6>* J_other) {
  Pose c;
  c.q = qmul(a.q, b.q);
  c.R = a.R * b.R;
  const V3 Rt = a.R * b.p;
  c.p = Rt + a.p;
  if (J_this) {
    *J_this = Mat<6, 6>::identity();
    const M3 m = -hat(Rt);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) (*J_this)(3 + i, j) = m(i, j);
  }
  if (J_other) {
    *J_other = Mat<6, 6>::zero();
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) (*J_other)(i, j) = c.R(i, j), (*J_other)(3 + i, 3 + j) = a.R(i, j);
  }
  return c;
}
/// T^-1 with d tangent(T^-1)/d tangent(T), both left/additive.
inline Pose group_inverse(const Pose& a, Mat<6, 6>* J) {
  Pose c;
  c.q = qconj(a.q);
  c.R = T(a.R);
  c.p = -(c.R * a.p);
  if (J) {
    *J = Mat<6, 6>::zero();
    const M3 m = -(c.R * hat(a.p));
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) (*J)(i, j) = -c.R(i, j), (*J)(3 + i, j) = m(i, j), (*J)(3 + i, 3 + j) = -c.R(i, j);
  }
  return c;
}
/// R p + t (pinned by d p_s / d p_w = R_sw, bearing.cpp:75) with d/d tangent(T) (left/additive), 3x6.
inline V3 vector_plus(const Pose& a, const V3& p, Mat<3, 6>* J) {
  const V3 Rp = a.R * p;
  if (J) {
    const M3 m = -hat(Rp);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) (*J)(i, j) = m(i, j), (*J)(i, 3 + j) = (i == j) ? 1.0 : 0.0;
  }
  return Rp + a.p;
}

// ---------------------------------------------------------------------------------------------------------
// Camera primitives (EXTERNAL Camera::ProjectToPlane, RadialTangentialDistortion<.,2>::distort, Intrinsics::denormalize;
// pixel.cpp:86-99; parameter orders settings.yaml:38-45).
// ---------------------------------------------------------------------------------------------------------
inline Mat<2, 1> project_to_plane(const V3& p, Mat<2, 3>* J) {
  Mat<2, 1> n;
  const double iz = 1.0 / p[2];
  n[0] = p[0] * iz, n[1] = p[1] * iz;
  if (J) {
    (*J)(0, 0) = iz, (*J)(0, 1) = 0, (*J)(0, 2) = -p[0] * iz * iz;
    (*J)(1, 0) = 0, (*J)(1, 1) = iz, (*J)(1, 2) = -p[1] * iz * iz;
  }
  return n;
}
inline Mat<2, 1> distort_radtan(const double* d, const Mat<2, 1>& n, Mat<2, 2>* J_n, Mat<2, 4>* J_d) {
  const double k1 = d[0], k2 = d[1], p1 = d[2], p2 = d[3];
  const double x = n[0], y = n[1], r2 = x * x + y * y, r4 = r2 * r2;
  const double rad = 1 + k1 * r2 + k2 * r4;
  Mat<2, 1> o;
  o[0] = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
  o[1] = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
  if (J_n) {
    const double drad_dx = 2 * k1 * x + 4 * k2 * r2 * x, drad_dy = 2 * k1 * y + 4 * k2 * r2 * y;
    (*J_n)(0, 0) = rad + x * drad_dx + 2 * p1 * y + 6 * p2 * x;
    (*J_n)(0, 1) = x * drad_dy + 2 * p1 * x + 2 * p2 * y;
    (*J_n)(1, 0) = y * drad_dx + 2 * p1 * x + 2 * p2 * y;
    (*J_n)(1, 1) = rad + y * drad_dy + 6 * p1 * y + 2 * p2 * x;
  }
  if (J_d) {
    (*J_d)(0, 0) = x * r2, (*J_d)(0, 1) = x * r4, (*J_d)(0, 2) = 2 * x * y, (*J_d)(0, 3) = r2 + 2 * x * x;
    (*J_d)(1, 0) = y * r2, (*J_d)(1, 1) = y * r4, (*J_d)(1, 2) = r2 + 2 * y * y, (*J_d)(1, 3) = 2 * x * y;
  }
  return o;
}
/// intrinsics [cx, cy, fx, fy] (settings.yaml:38-40).
inline Mat<2, 1> denormalize(const double* c, const Mat<2, 1>& n, Mat<2, 2>* J_n, Mat<2, 4>* J_i) {
  Mat<2, 1> o;
  o[0] = c[0] + c[2] * n[0], o[1] = c[1] + c[3] * n[1];
  if (J_n) (*J_n)(0, 0) = c[2], (*J_n)(0, 1) = 0, (*J_n)(1, 0) = 0, (*J_n)(1, 1) = c[3];
  if (J_i) {
    *J_i = Mat<2, 4>::zero();
    (*J_i)(0, 0) = 1, (*J_i)(1, 1) = 1, (*J_i)(0, 2) = n[0], (*J_i)(1, 3) = n[1];
  }
  return o;
}

// ---------------------------------------------------------------------------------------------------------
// Evaluators. Each returns the prediction and (optionally) the dense ambient J_e (n_out x num_parameters).
// ---------------------------------------------------------------------------------------------------------
struct Prediction {
  int n = 0;
  double v[7];
};

/// bearing.cpp:14-79 (p_J_e == nullptr -> value-only branch :42-51).
inline Prediction evaluate_bearing(const Basis& basis, double stamp, const double* const* p_ps, const Layout& L, DMat* p_J_e,
                                   const double* const* p_js) {
  const int k = basis.k;
  const double* const* p_S_wb_0 = p_ps + L.static_state_idx;
  const int o_T_bs = L.static_sensor_idx + 0, o_p_w = L.static_observation_idx + 0;
  const Pose T_bs = make_pose(p_ps[o_T_bs]);
  const V3 p_w = v3(p_ps[o_p_w][0], p_ps[o_p_w][1], p_ps[o_p_w][2]);
  Prediction out;
  out.n = 3;
  StateResult S;
  if (!p_J_e) {
    state_evaluate(basis, p_S_wb_0, stamp, 0, false, &S);
    const Pose T_wb = {S.s.q, S.s.R, S.s.p};
    const Pose T_ws = group_plus(T_wb, T_bs, nullptr, nullptr);
    const Pose T_sw = group_inverse(T_ws, nullptr);
    const V3 p_s = vector_plus(T_sw, p_w, nullptr);
    out.v[0] = p_s[0], out.v[1] = p_s[1], out.v[2] = p_s[2];
    return out;
  }
  p_J_e->set_zero(3, L.num_parameters);
  const bool J_S_wb = any_non_null(p_js, L.static_state_idx, L.static_state_idx + k);
  state_evaluate(basis, p_S_wb_0, stamp, 0, J_S_wb, &S);
  const Pose T_wb = {S.s.q, S.s.R, S.s.p};
  Mat<6, 6> J_T_ws_S_wb, J_T_ws_T_bs, J_T_sw_T_ws;
  const Pose T_ws = group_plus(T_wb, T_bs, &J_T_ws_S_wb, &J_T_ws_T_bs);
  const Pose T_sw = group_inverse(T_ws, &J_T_sw_T_ws);
  Mat<3, 6> J_p_s_T_sw;
  const V3 p_s = vector_plus(T_sw, p_w, &J_p_s_T_sw);
  const Mat<3, 6> J_p_s_T_ws = J_p_s_T_sw * J_T_sw_T_ws;
  if (J_S_wb) {
    DMat A;
    A.set_zero(3, 6);
    A.set_block(0, 0, J_p_s_T_ws * J_T_ws_S_wb);
    scatter_state(dmul(A, S.J[0]), p_S_wb_0, k, L, p_J_e);
  }
  if (p_js[o_T_bs]) {
    DMat A;
    A.set_zero(3, 6);
    A.set_block(0, 0, J_p_s_T_ws * J_T_ws_T_bs);
    scatter_se3(A, 0, quat_adapter_right(T_bs.q), p_J_e, L.offsets[o_T_bs]);
  }
  if (p_js[o_p_w]) p_J_e->set_block(0, L.offsets[o_p_w], T_sw.R);
  out.v[0] = p_s[0], out.v[1] = p_s[1], out.v[2] = p_s[2];
  return out;
}

/// pixel.cpp:16-146 (the four distortion/intrinsics branches :91-135 collapse to per-block null checks).
inline Prediction evaluate_pixel(const Basis& basis, double stamp, const double* const* p_ps, const Layout& L, DMat* p_J_e,
                                 const double* const* p_js) {
  const int k = basis.k;
  const double* const* p_S_wb_0 = p_ps + L.static_state_idx;
  const int o_T_bs = L.static_sensor_idx + 0, o_i = L.static_sensor_idx + 1, o_d = L.static_sensor_idx + 2;
  const int o_p_w = L.static_observation_idx + 0;
  const Pose T_bs = make_pose(p_ps[o_T_bs]);
  const double* c_i = p_ps[o_i];
  const double* c_d = p_ps[o_d];
  const V3 p_w = v3(p_ps[o_p_w][0], p_ps[o_p_w][1], p_ps[o_p_w][2]);
  Prediction out;
  out.n = 2;
  StateResult S;
  if (!p_J_e) {
    state_evaluate(basis, p_S_wb_0, stamp, 0, false, &S);
    const Pose T_wb = {S.s.q, S.s.R, S.s.p};
    const Pose T_sw = group_inverse(group_plus(T_wb, T_bs, nullptr, nullptr), nullptr);
    const V3 p_s = vector_plus(T_sw, p_w, nullptr);
    const auto n_px_s = project_to_plane(p_s, nullptr);
    const auto d_n_px_s = distort_radtan(c_d, n_px_s, nullptr, nullptr);
    const auto d_px_s = denormalize(c_i, d_n_px_s, nullptr, nullptr);
    out.v[0] = d_px_s[0], out.v[1] = d_px_s[1];
    return out;
  }
  p_J_e->set_zero(2, L.num_parameters);
  const bool J_S_wb = any_non_null(p_js, L.static_state_idx, L.static_state_idx + k);
  state_evaluate(basis, p_S_wb_0, stamp, 0, J_S_wb, &S);
  const Pose T_wb = {S.s.q, S.s.R, S.s.p};
  Mat<6, 6> J_T_ws_S_wb, J_T_ws_T_bs, J_T_sw_T_ws;
  const Pose T_ws = group_plus(T_wb, T_bs, &J_T_ws_S_wb, &J_T_ws_T_bs);
  const Pose T_sw = group_inverse(T_ws, &J_T_sw_T_ws);
  Mat<3, 6> J_p_s_T_sw;
  const V3 p_s = vector_plus(T_sw, p_w, &J_p_s_T_sw);
  Mat<2, 3> J_n_px_s_p_s;
  const auto n_px_s = project_to_plane(p_s, &J_n_px_s_p_s);
  Mat<2, 2> J_d_n_px_s_n_px_s, J_r_d_n_px_s;
  Mat<2, 4> J_d_n_px_s_d, J_d_px_s_c_i;
  const auto d_n_px_s = distort_radtan(c_d, n_px_s, &J_d_n_px_s_n_px_s, &J_d_n_px_s_d);
  const auto d_px_s = denormalize(c_i, d_n_px_s, &J_r_d_n_px_s, &J_d_px_s_c_i);
  if (p_js[o_d]) p_J_e->set_block(0, L.offsets[o_d], J_r_d_n_px_s * J_d_n_px_s_d);
  if (p_js[o_i]) p_J_e->set_block(0, L.offsets[o_i], J_d_px_s_c_i);
  const Mat<2, 2> J_r_n_px_s = J_r_d_n_px_s * J_d_n_px_s_n_px_s;
  const Mat<2, 3> J_r_p_s = J_r_n_px_s * J_n_px_s_p_s;
  const Mat<2, 6> J_r_T_ws = J_r_p_s * J_p_s_T_sw * J_T_sw_T_ws;
  if (J_S_wb) {
    DMat A;
    A.set_zero(2, 6);
    A.set_block(0, 0, J_r_T_ws * J_T_ws_S_wb);
    scatter_state(dmul(A, S.J[0]), p_S_wb_0, k, L, p_J_e);
  }
  if (p_js[o_T_bs]) {
    DMat A;
    A.set_zero(2, 6);
    A.set_block(0, 0, J_r_T_ws * J_T_ws_T_bs);
    scatter_se3(A, 0, quat_adapter_right(T_bs.q), p_J_e, L.offsets[o_T_bs]);
  }
  if (p_js[o_p_w]) p_J_e->set_block(0, L.offsets[o_p_w], J_r_p_s * T_sw.R);
  out.v[0] = d_px_s[0], out.v[1] = d_px_s[1];
  return out;
}

/// manifold.cpp:12-61 — returns T_ws as 7-vector; J_e has the 6 tangent rows of T_ws.
inline Prediction evaluate_manifold(const Basis& basis, double stamp, const double* const* p_ps, const Layout& L, DMat* p_J_e,
                                    const double* const* p_js) {
  const int k = basis.k;
  const double* const* p_S_wb_0 = p_ps + L.static_state_idx;
  const int o_T_bs = L.static_sensor_idx + 0;
  const Pose T_bs = make_pose(p_ps[o_T_bs]);
  Prediction out;
  out.n = 7;
  StateResult S;
  const bool J_S_wb = p_J_e && any_non_null(p_js, L.static_state_idx, L.static_state_idx + k);
  state_evaluate(basis, p_S_wb_0, stamp, 0, J_S_wb, &S);
  const Pose T_wb = {S.s.q, S.s.R, S.s.p};
  Mat<6, 6> J_r_S_wb, J_r_T_bs;
  const Pose T_ws = group_plus(T_wb, T_bs, p_J_e ? &J_r_S_wb : nullptr, p_J_e ? &J_r_T_bs : nullptr);
  if (p_J_e) {
    p_J_e->set_zero(6, L.num_parameters);
    if (J_S_wb) {
      DMat A;
      A.set_zero(6, 6);
      A.set_block(0, 0, J_r_S_wb);
      scatter_state(dmul(A, S.J[0]), p_S_wb_0, k, L, p_J_e);
    }
    if (p_js[o_T_bs]) {
      DMat A;
      A.set_zero(6, 6);
      A.set_block(0, 0, J_r_T_bs);
      scatter_se3(A, 0, quat_adapter_right(T_bs.q), p_J_e, L.offsets[o_T_bs]);
    }
  }
  out.v[0] = T_ws.q.x, out.v[1] = T_ws.q.y, out.v[2] = T_ws.q.z, out.v[3] = T_ws.q.w;
  out.v[4] = T_ws.p[0], out.v[5] = T_ws.p[1], out.v[6] = T_ws.p[2];
  return out;
}

/// OrthonormalityAlignment 6-parameter lower-triangular matrix [c00,c11,c22,c10,c20,c21] (settings.yaml:87-89,
/// inertial.cpp:118-119 asMatrix, :166,172 align).
inline M3 align_matrix(const double* c) {
  M3 m = M3::zero();
  m(0, 0) = c[0], m(1, 1) = c[1], m(2, 2) = c[2], m(1, 0) = c[3], m(2, 0) = c[4], m(2, 1) = c[5];
  return m;
}
inline Mat<3, 6> align_param_jacobian(const V3& v) {
  Mat<3, 6> J = Mat<3, 6>::zero();
  J(0, 0) = v[0], J(1, 1) = v[1], J(2, 2) = v[2], J(1, 3) = v[0], J(2, 4) = v[0], J(2, 5) = v[1];
  return J;
}

/// inertial.cpp:13-205. Two forms of the Jacobian (the prediction :200-203 is the same):
///   literal (default, HS_INERTIAL_AS_REFERENCE): the in-tree text — the linear rows of the state and extrinsic-rotation columns carry
///     I_g where the prediction has I_a (:136,142,148,158), the lever arm of those columns is t_bs alone, and the S_g / X_a
///     contributions to the state, extrinsic and gravity columns are absent (:134-153,155-162,198);
///   exact (HS_INERTIAL_EXACT): the derivative of the prediction (I_a in the linear rows, per-row lever arms X_a.col(i) + t_bs,
///     S_g terms in the angular rows).
/// They coincide for I_g = I_a, S_g = 0, X_a = 0, i.e. wherever the reference is exercised (SURVEY.md §8a "Latent reference bug").
inline Prediction evaluate_inertial(const Basis& basis, const Basis& bias_basis, double stamp, const double* const* p_ps, const Layout& L,
                                    DMat* p_J_e, const double* const* p_js, bool literal) {
  const int k = basis.k, kb = bias_basis.k;
  const int o_T_bs = L.static_sensor_idx + 0, o_i_g = L.static_sensor_idx + 1, o_i_a = L.static_sensor_idx + 2;
  const int o_S_g = L.static_sensor_idx + 3, o_X_a = L.static_sensor_idx + 4;
  const int o_b_g = L.dynamic_sensor_idx, o_b_a = o_b_g + kb, o_g_w = L.static_observation_idx;
  const Pose T_bs = make_pose(p_ps[o_T_bs]);
  const M3 I_g = align_matrix(p_ps[o_i_g]), I_a = align_matrix(p_ps[o_i_a]);
  M3 S_g, X_a;  // column-major maps (inertial.cpp:48-49)
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) S_g(r, c) = p_ps[o_S_g][3 * c + r], X_a(r, c) = p_ps[o_X_a][3 * c + r];
  const V3 g_w = v3(p_ps[o_g_w][0], p_ps[o_g_w][1], p_ps[o_g_w][2]);

  const bool want_J = p_J_e != nullptr;
  const bool J_S_wb = want_J && any_non_null(p_js, L.static_state_idx, L.static_state_idx + k);
  const bool J_b_g = want_J && any_non_null(p_js, o_b_g, o_b_g + kb);
  const bool J_b_a = want_J && any_non_null(p_js, o_b_a, o_g_w);
  StateResult S;
  state_evaluate(basis, p_ps + L.static_state_idx, stamp, 2, J_S_wb, &S);

  // Bias splines (value only; inertial.cpp:58-60, :100-108).
  double Bg[kMaxOrder], Ba[kMaxOrder];
  auto bias_u = [&](int o) {
    const int i = (kb - 1) / 2;
    const double ti = p_ps[o + i][3], dt = p_ps[o + i + 1][3] - ti;
    return (stamp - ti) / dt;
  };
  const V3 b_g = r3_spline_evaluate(bias_basis, p_ps + o_b_g, bias_u(o_b_g), Bg);
  const V3 b_a = r3_spline_evaluate(bias_basis, p_ps + o_b_a, bias_u(o_b_a), Ba);

  const M3 R_bw = T(S.s.R), R_sb = T(T_bs.R);
  const V3 w_b = S.s.w;
  const M3 w_b_x = hat(w_b);
  const V3 A_lin = R_bw * S.s.a;               // body-frame linear acceleration (A_wb.linear())
  const V3 a_b_i = A_lin - R_bw * g_w;         // ideal acceleration (:125)
  const M3 F_a = w_b_x * w_b_x + hat(S.s.al);  // (:127)
  V3 lever[3], a_b_m;
  for (int i = 0; i < 3; ++i) {
    lever[i] = v3(X_a(0, i) + T_bs.p[0], X_a(1, i) + T_bs.p[1], X_a(2, i) + T_bs.p[2]);
    a_b_m[i] = a_b_i[i] + F_a(i, 0) * lever[i][0] + F_a(i, 1) * lever[i][1] + F_a(i, 2) * lever[i][2];  // (:128)
  }
  const M3 I_g_R_sb = I_g * R_sb, I_a_R_sb = I_a * R_sb;
  const V3 w_s = R_sb * w_b, a_s = R_sb * a_b_m;

  Prediction out;
  out.n = 6;
  const V3 ang = I_g_R_sb * w_b + S_g * a_b_m + b_g;  // (:200-203)
  const V3 lin = I_a_R_sb * a_b_m + b_a;
  for (int i = 0; i < 3; ++i) out.v[i] = ang[i], out.v[3 + i] = lin[i];
  if (!want_J) return out;

  p_J_e->set_zero(6, L.num_parameters);
  // d a_m / d w and d a_m / d alpha with per-row lever arms l_i = X_a.col(i) + t_bs.
  M3 L_w, L_al;
  const M3 S_gJ = literal ? M3::zero() : S_g;             // S_g as it enters the state / extrinsic / gravity columns
  const M3 I_x_R_sb = literal ? I_g_R_sb : I_a_R_sb;      // (:136,142,148) vs the matrix of the prediction
  const M3 I_x = literal ? I_g : I_a;                     // (:158)
  for (int i = 0; i < 3; ++i) {
    const M3 lx = hat(literal ? T_bs.p : lever[i]);
    const M3 mw = -1.0 * (2.0 * (w_b_x * lx) - lx * w_b_x);  // (:142) generalised to per-row lever arms
    const M3 ma = -1.0 * lx;                                  // (:148)
    for (int j = 0; j < 3; ++j) L_w(i, j) = mw(i, j), L_al(i, j) = ma(i, j);
  }
  if (J_S_wb) {
    // d a_m / d tangent columns = hat(a_i) R_bw * J0.ang + L_w * J1.ang + L_al * J2.ang + J2.lin   (J2.lin = R_bw Bdd)
    Mat<3, 6> Ja_value = Mat<3, 6>::zero(), Ja_vel = Mat<3, 6>::zero(), Ja_acc = Mat<3, 6>::zero();
    const M3 hv = hat(a_b_i) * R_bw;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) Ja_value(i, j) = hv(i, j), Ja_vel(i, j) = L_w(i, j), Ja_acc(i, j) = L_al(i, j), Ja_acc(i, 3 + j) = (i == j);
    Mat<6, 6> J_value = Mat<6, 6>::zero(), J_velocity = Mat<6, 6>::zero(), J_acceleration = Mat<6, 6>::zero();
    const Mat<3, 6> gv = S_gJ * Ja_value, gw = S_gJ * Ja_vel, ga = S_gJ * Ja_acc;
    const Mat<3, 6> lv = I_x_R_sb * Ja_value, lw = I_x_R_sb * Ja_vel, la = I_x_R_sb * Ja_acc, la_lin = I_a_R_sb * Ja_acc;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 6; ++j) {
        J_value(i, j) = gv(i, j), J_value(3 + i, j) = lv(i, j);
        J_velocity(i, j) = gw(i, j) + (j < 3 ? I_g_R_sb(i, j) : 0.0), J_velocity(3 + i, j) = lw(i, j);
        J_acceleration(i, j) = ga(i, j), J_acceleration(3 + i, j) = j < 3 ? la(i, j) : la_lin(i, j);  // (:148 | :150)
      }
    DMat Jv, Jw, Ja;
    Jv.set_zero(6, 6), Jw.set_zero(6, 6), Ja.set_zero(6, 6);
    Jv.set_block(0, 0, J_value), Jw.set_block(0, 0, J_velocity), Ja.set_block(0, 0, J_acceleration);
    DMat Jt = dmul(Jv, S.J[0]);
    const DMat t1 = dmul(Jw, S.J[1]), t2 = dmul(Ja, S.J[2]);
    for (size_t i = 0; i < Jt.a.size(); ++i) Jt.a[i] += t1.a[i] + t2.a[i];
    scatter_state(Jt, p_ps + L.static_state_idx, k, L, p_J_e);
  }
  if (p_js[o_T_bs]) {  // (:155-162) right/additive tangent of T_bs
    const M3 amt = F_a;  // d a_m / d t_bs
    const M3 ang_ang = I_g * hat(w_s);
    const M3 lin_ang = I_x * hat(a_s);
    const M3 ang_lin = S_gJ * amt;
    const M3 lin_lin = I_a_R_sb * amt;
    DMat A;
    A.set_zero(6, 6);
    A.set_block(0, 0, ang_ang), A.set_block(3, 0, lin_ang), A.set_block(0, 3, ang_lin), A.set_block(3, 3, lin_lin);
    scatter_se3(A, 0, quat_adapter_right(T_bs.q), p_J_e, L.offsets[o_T_bs]);
  }
  if (p_js[o_i_g]) p_J_e->set_block(0, L.offsets[o_i_g], align_param_jacobian(w_s));  // (:164-168)
  if (p_js[o_i_a]) p_J_e->set_block(3, L.offsets[o_i_a], align_param_jacobian(a_s));  // (:170-174)
  if (p_js[o_S_g]) {                                                                  // (:176-187)
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) (*p_J_e)(r, L.offsets[o_S_g] + 3 * c + r) = a_b_m[c];
  }
  if (p_js[o_X_a]) {  // (:189-194) + gyro rows through S_g
    for (int i = 0; i < 3; ++i)
      for (int r = 0; r < 3; ++r)
        for (int row = 0; row < 3; ++row) {
          (*p_J_e)(3 + row, L.offsets[o_X_a] + 3 * i + r) = I_a_R_sb(row, i) * F_a(i, r);
          (*p_J_e)(row, L.offsets[o_X_a] + 3 * i + r) = S_gJ(row, i) * F_a(i, r);
        }
  }
  for (int j = 0; j < kb; ++j) {  // (:196-197)
    if (J_b_g && p_js[o_b_g + j]) p_J_e->set_block(0, L.offsets[o_b_g + j], Bg[j] * M3::identity());
    if (J_b_a && p_js[o_b_a + j]) p_J_e->set_block(3, L.offsets[o_b_a + j], Ba[j] * M3::identity());
  }
  if (p_js[o_g_w]) {  // (:198)
    p_J_e->set_block(3, L.offsets[o_g_w], -1.0 * (I_a_R_sb * R_bw));
    p_J_e->set_block(0, L.offsets[o_g_w], -1.0 * (S_gJ * R_bw));
  }
  return out;
}

// ---------------------------------------------------------------------------------------------------------
// Metrics (EXTERNAL; interface distance(lhs, rhs, J_lhs*, J_rhs*), exteroceptive.cpp:111-113,136).
// ---------------------------------------------------------------------------------------------------------
/// AngularMetric<Bearing>: scalar angle between prediction and measured bearing, atan2(|p x b|, p.b).
inline void metric_angular(const double* p, const double* b, double* out, DMat* J) {
  const V3 P = v3(p[0], p[1], p[2]), Bv = v3(b[0], b[1], b[2]);
  const V3 c = cross(P, Bv);
  const double n = norm(c), d = dot(P, Bv);
  out[0] = std::atan2(n, d);
  if (J) {
    J->set_zero(1, 3);
    if (n > 0) {
      const V3 bxc = cross(Bv, c);
      const double den = n * n + d * d;
      for (int i = 0; i < 3; ++i) (*J)(0, i) = (d * bxc[i] / n - n * Bv[i]) / den;
    }
  }
}
/// ManifoldMetric<SE3>: [Log(R_m^T R) ; p - p_m], Jacobian w.r.t. the left/additive tangent of the prediction.
inline void metric_se3(const double* pred, const double* meas, double* out, DMat* J) {
  const Quat q = {pred[0], pred[1], pred[2], pred[3]}, qm = {meas[0], meas[1], meas[2], meas[3]};
  const V3 r = so3_log_q(qmul(qconj(qm), q));
  for (int i = 0; i < 3; ++i) out[i] = r[i], out[3 + i] = pred[4 + i] - meas[4 + i];
  if (J) {
    J->set_zero(6, 6);
    const M3 Jm = so3_Jr_inv(r) * T(qmat(q));
    J->set_block(0, 0, Jm);
    J->set_block(3, 3, M3::identity());
  }
}

// ---------------------------------------------------------------------------------------------------------
// ExteroceptiveCost::Evaluate (exteroceptive.cpp:101-160) for weights == nullptr, metric != nullptr
// (all production call sites, optimizer.cpp:191-194,214-217,236-239,255-258).
// jacobians[i] row-major num_residuals x size_i, individually nullable.
// ---------------------------------------------------------------------------------------------------------
struct CostContext {
  FactorType type;
  const Basis* basis;
  const Basis* bias_basis;
  double stamp;
  const double* measurement;  // pixel 2 | bearing 3 | pose 7 | [w; a] 6
  bool inertial_literal = true;  // HS_INERTIAL_AS_REFERENCE
  const double* weights = nullptr;  // CostConfiguration::weights: n_res x n_res, row-major (exteroceptive.cpp:109-121,129-147)
};

inline bool cost_evaluate(const CostContext& ctx, const Layout& L, const double* const* parameters, double* residuals, double** jacobians) {
  DMat J_e;
  DMat* p_J_e = jacobians ? &J_e : nullptr;
  Prediction pred;
  switch (ctx.type) {
    case kPixel: pred = evaluate_pixel(*ctx.basis, ctx.stamp, parameters, L, p_J_e, jacobians); break;
    case kBearing: pred = evaluate_bearing(*ctx.basis, ctx.stamp, parameters, L, p_J_e, jacobians); break;
    case kPrior: pred = evaluate_manifold(*ctx.basis, ctx.stamp, parameters, L, p_J_e, jacobians); break;
    case kInertial: pred = evaluate_inertial(*ctx.basis, *ctx.bias_basis, ctx.stamp, parameters, L, p_J_e, jacobians, ctx.inertial_literal); break;
  }
  DMat J_m;
  DMat* p_J_m = jacobians ? &J_m : nullptr;
  switch (ctx.type) {
    case kPixel:
      for (int i = 0; i < 2; ++i) residuals[i] = pred.v[i] - ctx.measurement[i];
      break;
    case kInertial:
      for (int i = 0; i < 6; ++i) residuals[i] = pred.v[i] - ctx.measurement[i];
      break;
    case kBearing: metric_angular(pred.v, ctx.measurement, residuals, p_J_m); break;
    case kPrior: metric_se3(pred.v, ctx.measurement, residuals, p_J_m); break;
  }
  const int nr = L.num_residuals;
  if (ctx.weights) {  // output = weights * distance(..)
    double t[6];
    for (int r = 0; r < nr; ++r) {
      t[r] = 0;
      for (int c = 0; c < nr; ++c) t[r] += ctx.weights[r * nr + c] * residuals[c];
    }
    for (int r = 0; r < nr; ++r) residuals[r] = t[r];
  }
  if (!jacobians) return true;
  DMat J_w = (ctx.type == kBearing || ctx.type == kPrior) ? dmul(J_m, J_e) : J_e;  // Cartesian metric: J_m = I
  if (ctx.weights) {  // J_w = weights * J_m * J_e
    DMat W;
    W.set_zero(nr, nr);
    for (int r = 0; r < nr; ++r)
      for (int c = 0; c < nr; ++c) W(r, c) = ctx.weights[r * nr + c];
    J_w = dmul(W, J_w);
  }
  for (size_t i = 0; i < L.sizes.size(); ++i) {
    if (!jacobians[i]) continue;
    const int size = L.sizes[i];
    for (int r = 0; r < L.num_residuals; ++r)
      for (int c = 0; c < size; ++c) jacobians[i][r * size + c] = J_w(r, L.offsets[i] + c);
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// Ceres manifolds: Plus and PlusJacobian (ambient x local, row-major).
// ---------------------------------------------------------------------------------------------------------
enum ManifoldKind : int {
  kManifoldConstant = 0,   // SubsetManifold with all indices fixed (euclidean.hpp:35-36,45-50): local size 0
  kManifoldEuclidean = 1,  // EuclideanManifold (euclidean.hpp:38)
  kManifoldControlPoint = 2,  // Product(Product(EigenQuaternion, R3), stamp const)  (stamped.hpp:35-36, se3.cpp:22-23)
  kManifoldSE3 = 3,        // Product(EigenQuaternion, R3)  (sensor extrinsics, sensors/sensor.cpp:26-29)
  kManifoldSphere3 = 4,    // SphereManifold<3> (manifolds/variables/bearing.cpp:15, gravity.hpp:11-17)
  kManifoldBiasPoint = 5,  // Product(R3, stamp const)  (imu.cpp:64-66)
};
inline int manifold_local_size(ManifoldKind m, int ambient) {
  switch (m) {
    case kManifoldConstant: return 0;
    case kManifoldEuclidean: return ambient;
    case kManifoldControlPoint: return 6;
    case kManifoldSE3: return 6;
    case kManifoldSphere3: return 2;
    case kManifoldBiasPoint: return 3;
  }
  return 0;
}
/// EigenQuaternionManifold::Plus: x+ = [sin|d|/|d| d ; cos|d|] (x) x.
inline void quat_plus(const double* x, const double* d, double* out) {
  const double n = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  Quat dq;
  if (n == 0.0) {
    dq = {0, 0, 0, 1};
  } else {
    const double s = std::sin(n) / n;
    dq = {s * d[0], s * d[1], s * d[2], std::cos(n)};
  }
  const Quat r = qmul(dq, Quat{x[0], x[1], x[2], x[3]});
  out[0] = r.x, out[1] = r.y, out[2] = r.z, out[3] = r.w;
}
/// 4x3 row-major: column i = (e_i, 0) (x) x.
inline void quat_plus_jacobian(const double* x, double* J) {
  const Quat q = {x[0], x[1], x[2], x[3]};
  for (int i = 0; i < 3; ++i) {
    Quat e = {0, 0, 0, 0};
    (&e.x)[i] = 1.0;
    const Quat r = qmul(e, q);
    J[0 * 3 + i] = r.x, J[1 * 3 + i] = r.y, J[2 * 3 + i] = r.z, J[3 * 3 + i] = r.w;
  }
}
/// Ceres SphereManifold<3> Householder vector (ceres/internal/sphere_manifold_functions.h).
inline void sphere_householder(const double* x, double* v, double* beta) {
  const double sigma = x[0] * x[0] + x[1] * x[1];
  v[0] = x[0], v[1] = x[1], v[2] = 1.0;
  *beta = 0.0;
  const double x_pn = x[2];
  if (sigma <= 2.220446049250313e-16) {  // std::numeric_limits<double>::epsilon()
    if (x_pn < 0) *beta = 2.0;
    return;
  }
  const double mu = std::sqrt(x_pn * x_pn + sigma);
  double v_pivot;
  if (x_pn <= 0.0)
    v_pivot = x_pn - mu;
  else
    v_pivot = -sigma / (x_pn + mu);
  *beta = 2.0 * v_pivot * v_pivot / (sigma + v_pivot * v_pivot);
  v[0] /= v_pivot, v[1] /= v_pivot;
}
inline void sphere_plus(const double* x, const double* d, double* out) {
  const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1]);
  if (nd == 0.0) {
    out[0] = x[0], out[1] = x[1], out[2] = x[2];
    return;
  }
  double v[3], beta;
  sphere_householder(x, v, &beta);
  const double s = std::sin(nd) / nd;
  const double y[3] = {s * d[0], s * d[1], std::cos(nd)};
  const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  const double vy = beta * (v[0] * y[0] + v[1] * y[1] + v[2] * y[2]);
  for (int i = 0; i < 3; ++i) out[i] = nx * (y[i] - v[i] * vy);
}
/// 3x2 row-major.
inline void sphere_plus_jacobian(const double* x, double* J) {
  double v[3], beta;
  sphere_householder(x, v, &beta);
  const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 3; ++r) J[r * 2 + i] = nx * ((r == i ? 1.0 : 0.0) - beta * v[r] * v[i]);
}

inline void manifold_plus(ManifoldKind m, int ambient, const double* x, const double* d, double* out) {
  switch (m) {
    case kManifoldConstant:
      for (int i = 0; i < ambient; ++i) out[i] = x[i];
      break;
    case kManifoldEuclidean:
      for (int i = 0; i < ambient; ++i) out[i] = x[i] + d[i];
      break;
    case kManifoldControlPoint:
      quat_plus(x, d, out);
      for (int i = 0; i < 3; ++i) out[4 + i] = x[4 + i] + d[3 + i];
      out[7] = x[7];
      break;
    case kManifoldSE3:
      quat_plus(x, d, out);
      for (int i = 0; i < 3; ++i) out[4 + i] = x[4 + i] + d[3 + i];
      break;
    case kManifoldSphere3: sphere_plus(x, d, out); break;
    case kManifoldBiasPoint:
      for (int i = 0; i < 3; ++i) out[i] = x[i] + d[i];
      out[3] = x[3];
      break;
  }
}
/// ambient x local row-major PlusJacobian.
inline void manifold_plus_jacobian(ManifoldKind m, int ambient, const double* x, double* J) {
  const int local = manifold_local_size(m, ambient);
  for (int i = 0; i < ambient * local; ++i) J[i] = 0.0;
  switch (m) {
    case kManifoldConstant: break;
    case kManifoldEuclidean:
      for (int i = 0; i < ambient; ++i) J[i * local + i] = 1.0;
      break;
    case kManifoldControlPoint:
    case kManifoldSE3: {
      double Jq[12];
      quat_plus_jacobian(x, Jq);
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 3; ++c) J[r * local + c] = Jq[r * 3 + c];
      for (int i = 0; i < 3; ++i) J[(4 + i) * local + 3 + i] = 1.0;
      break;
    }
    case kManifoldSphere3: sphere_plus_jacobian(x, J); break;
    case kManifoldBiasPoint:
      for (int i = 0; i < 3; ++i) J[i * local + i] = 1.0;
      break;
  }
}

/// ceres::Manifold::Minus of the variable classes (wrapper.hpp:44-46), restated from Ceres' public definitions (third party, 2.1):
/// quaternion: [v ; w] = y (x) conj(x), delta = atan2(|v|, w) / |v| v; sphere: h = H y / |x|, delta = atan2(|h_t|, h_last) / |h_t| h_t;
/// Euclidean: y - x; products: block-wise; constant blocks have no tangent.
inline void manifold_minus(ManifoldKind m, int ambient, const double* y, const double* x, double* out) {
  auto quat = [&](double* o) {
    // r = y (x) conj(x), Hamilton, storage (x, y, z, w)
    const double ax = y[0], ay = y[1], az = y[2], aw = y[3], bx = -x[0], by = -x[1], bz = -x[2], bw = x[3];
    const double rx = aw * bx + ax * bw + ay * bz - az * by, ry = aw * by - ax * bz + ay * bw + az * bx, rz = aw * bz + ax * by - ay * bx + az * bw,
                 rw = aw * bw - ax * bx - ay * by - az * bz;
    const double n = std::sqrt(rx * rx + ry * ry + rz * rz);
    const double s = n == 0.0 ? 0.0 : std::atan2(n, rw) / n;
    o[0] = s * rx, o[1] = s * ry, o[2] = s * rz;
  };
  switch (m) {
    case kManifoldConstant: break;
    case kManifoldEuclidean:
      for (int i = 0; i < ambient; ++i) out[i] = y[i] - x[i];
      break;
    case kManifoldControlPoint:
    case kManifoldSE3:
      quat(out);
      for (int i = 0; i < 3; ++i) out[3 + i] = y[4 + i] - x[4 + i];
      break;
    case kManifoldSphere3: {
      double v[3], beta;
      sphere_householder(x, v, &beta);
      const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      const double vy = beta * (v[0] * y[0] + v[1] * y[1] + v[2] * y[2]);
      double h[3];
      for (int i = 0; i < 3; ++i) h[i] = (y[i] - v[i] * vy) / nx;
      const double n = std::sqrt(h[0] * h[0] + h[1] * h[1]);
      const double s = n == 0.0 ? 0.0 : std::atan2(n, h[2]) / n;
      out[0] = s * h[0], out[1] = s * h[1];
      break;
    }
    case kManifoldBiasPoint:
      for (int i = 0; i < 3; ++i) out[i] = y[i] - x[i];
      break;
  }
}
/// local x ambient row-major MinusJacobian (wrapper.hpp:48-50): d Minus(y, x) / dy at y = x.
inline void manifold_minus_jacobian(ManifoldKind m, int ambient, const double* x, double* J) {
  const int local = manifold_local_size(m, ambient);
  for (int i = 0; i < ambient * local; ++i) J[i] = 0.0;
  switch (m) {
    case kManifoldConstant: break;
    case kManifoldEuclidean:
      for (int i = 0; i < ambient; ++i) J[i * ambient + i] = 1.0;
      break;
    case kManifoldControlPoint:
    case kManifoldSE3: {
      double P[12];  // PlusJacobian of the quaternion (4 x 3): MinusJacobian is its transpose
      quat_plus_jacobian(x, P);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) J[r * ambient + c] = P[c * 3 + r];
      for (int i = 0; i < 3; ++i) J[(3 + i) * ambient + 4 + i] = 1.0;
      break;
    }
    case kManifoldSphere3: {
      double v[3], beta;
      sphere_householder(x, v, &beta);
      const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      for (int i = 0; i < 2; ++i)
        for (int c = 0; c < 3; ++c) J[i * 3 + c] = ((c == i ? 1.0 : 0.0) - beta * v[i] * v[c]) / nx;
      break;
    }
    case kManifoldBiasPoint:
      for (int i = 0; i < 3; ++i) J[i * ambient + i] = 1.0;
      break;
  }
}

/// Local Jacobian of one block: J_local (n_res x local) = J_block (n_res x ambient, row-major) * PlusJacobian.
inline void to_local(ManifoldKind m, int ambient, int n_res, const double* x, const double* J_block, double* J_local) {
  const int local = manifold_local_size(m, ambient);
  double P[9 * 9];  // largest block: 9-parameter Euclidean (S_g, X_a)
  manifold_plus_jacobian(m, ambient, x, P);
  for (int r = 0; r < n_res; ++r)
    for (int c = 0; c < local; ++c) {
      double s = 0;
      for (int a = 0; a < ambient; ++a) s += J_block[r * ambient + a] * P[a * local + c];
      J_local[r * local + c] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Robust losses (Ceres semantics; constants optimizer.cpp:204,226,250,267-268).
// rho[0..2] = rho(s), rho'(s), rho''(s) for s = |r|^2.
// ---------------------------------------------------------------------------------------------------------
enum LossKind : int { kLossNone = 0, kLossHuber = 1, kLossScaled = 2 };
struct Loss {
  LossKind kind;
  double a;  // Huber threshold or scale
};
inline Loss loss_for(FactorType t) {
  switch (t) {
    case kPixel: return {kLossHuber, 0.5};
    case kBearing: return {kLossHuber, 1.6e-3};
    case kPrior: return {kLossNone, 0.0};
    case kInertial: return {kLossScaled, 1.6e-5};
  }
  return {kLossNone, 0.0};
}
inline void loss_evaluate(const Loss& l, double s, double rho[3]) {
  switch (l.kind) {
    case kLossNone: rho[0] = s, rho[1] = 1, rho[2] = 0; break;
    case kLossScaled: rho[0] = l.a * s, rho[1] = l.a, rho[2] = 0; break;
    case kLossHuber: {
      const double b = l.a * l.a;
      if (s > b) {
        const double r = std::sqrt(s);
        rho[0] = 2 * l.a * r - b;
        rho[1] = std::max<double>(2.2250738585072014e-308, l.a / r);
        rho[2] = -rho[1] / (2 * s);
      } else {
        rho[0] = s, rho[1] = 1, rho[2] = 0;
      }
      break;
    }
  }
}
/// Ceres Corrector (corrector.cc): with rho'' <= 0 both residual and Jacobian are scaled by sqrt(rho').
/// All three in-tree losses have rho'' <= 0, so alpha = 0 always.
inline double corrector_scale(const double rho[3]) { return std::sqrt(rho[1]); }

}  // namespace hso
