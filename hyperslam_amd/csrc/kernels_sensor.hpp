// kernels_sensor.hpp — Jacobians w.r.t. the SENSOR parameter blocks (part of kernels.hpp; included once by capi.hip through it).
//
// The evaluators of the reference fill the extrinsic / intrinsic / distortion / IMU-intrinsic / S_g / X_a columns whenever Ceres hands
// them a non-null block pointer:
//   VisualBearingEvaluator   /root/reference/internal/hyper/optimizers/evaluators/bearing.cpp:74   (T_bs)
//   VisualPixelEvaluator     .../evaluators/pixel.cpp:91-135 (intrinsics, distortion), :141 (T_bs)
//   ManifoldEvaluator        .../evaluators/manifold.cpp:57                                      (T_bs)
//   InertialEvaluator        .../evaluators/inertial.cpp:155-162 (T_bs), :164-174 (i_g, i_a), :176-187 (S_g), :189-194 (X_a)
// In the optimizer these blocks are constant (camera.hpp:18, imu.hpp:18, optimizer.cpp:59-64), so the solver's linearisation kernels
// (kernels_linearize.hpp) do not carry them; the reference's own tests, however, probe every block
// (tests/include/tests/optimizers/evaluators/evaluator.hpp:38-65). These kernels produce them for hs_linearize's optional outputs and for
// hs_cost_function_evaluate, one residual block per lane, in Ceres-LOCAL coordinates:
//   T_bs = Product(EigenQuaternion, R3) (sensors/sensor.cpp:26-29): q_bs <- dq(delta) (x) q_bs, i.e. R_bs <- Exp(2 delta) R_bs; t_bs additive.
//   Every other sensor block is Euclidean.
// With y = R_wb^T (p_w - p_wb) - t_bs = R_bs p_s:
//   visual      d p_s / d delta = 2 hat(p_s) R_sb          d p_s / d t_bs = -R_sb
//   prior       d r_rot / d delta = 2 J_r^-1(r_rot) R_sb   d r_p / d t_bs = R_wb
//   inertial    d (I R_sb x) / d delta = 2 I R_sb hat(x)   d a_m / d t_bs = F_a
#pragma once
#include "kernels_common.hpp"

namespace hs {

constexpr int kSensorRecVisual = 12 + 8 + 8;                // J_ext 2 x 6 | J_intrinsics 2 x 4 | J_distortion 2 x 4
constexpr int kSensorRecPrior = 36;                         // J_ext 6 x 6
constexpr int kSensorRecInertial = 36 + 36 + 36 + 54 + 54;  // J_ext | J_i_g | J_i_a | J_S_g | J_X_a   (6 rows each)

template <int K>
HSD void visual_sensor_jacobians(const Tables& T, const double* cps, int q, bool robustify, double* rec) {
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
  const double* l = T.lm + 3 * T.v_lm[q];
  const V3 ps = to_sensor(qw, pw, cam, V3{l[0], l[1], l[2]}, nullptr, nullptr);
  double r[2], Jps[6];
  visual_measure(type, ps, cam, T.v_meas + 3 * q, true, r, Jps);
  double sr;
  loss_huber(r[0] * r[0] + r[1] * r[1], type == 0 ? kHuberPixel : kHuberBearing, &sr);
  if (!robustify) sr = 1.0;
  const M3 R_sb = transpose(qmat(Quat{cam[0], cam[1], cam[2], cam[3]}));
  const M3 d_rot = scale(2.0, mul(hat(ps), R_sb));
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rec[6 * i + c] = sr * (Jps[3 * i] * d_rot.m[c] + Jps[3 * i + 1] * d_rot.m[3 + c] + Jps[3 * i + 2] * d_rot.m[6 + c]);
      rec[6 * i + 3 + c] = -sr * (Jps[3 * i] * R_sb.m[c] + Jps[3 * i + 1] * R_sb.m[3 + c] + Jps[3 * i + 2] * R_sb.m[6 + c]);
    }
#pragma unroll
  for (int c = 0; c < 16; ++c) rec[12 + c] = 0.0;
  if (type == 0) {  // pixel.cpp:91-135: denormalize [cx cy fx fy] and radtan [k1 k2 p1 p2] parameter Jacobians
    const double fx = cam[9], fy = cam[10];
    const double k1 = cam[11], k2 = cam[12], p1 = cam[13], p2 = cam[14];
    const double iz = 1.0 / ps.z, x = ps.x * iz, y = ps.y * iz;
    const double r2 = x * x + y * y, r4 = r2 * r2, rad = 1.0 + k1 * r2 + k2 * r4;
    const double xd = x * rad + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x);
    const double yd = y * rad + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y;
    double* Ji = rec + 12;
    double* Jd = rec + 20;
    Ji[0] = sr, Ji[2] = sr * xd, Ji[4 + 1] = sr, Ji[4 + 3] = sr * yd;
    Jd[0] = sr * fx * x * r2, Jd[1] = sr * fx * x * r4, Jd[2] = sr * fx * 2.0 * x * y, Jd[3] = sr * fx * (r2 + 2.0 * x * x);
    Jd[4] = sr * fy * y * r2, Jd[5] = sr * fy * y * r4, Jd[6] = sr * fy * (r2 + 2.0 * y * y), Jd[7] = sr * fy * 2.0 * x * y;
  }
}

template <int K>
HSD void prior_sensor_jacobians(const Tables& T, const double* cps, int i, double* rec) {
  double u;
  segment_of(T.p_stamp[i], T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dl[1], ddl[1];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dl, ddl, 0);
  Quat qw;
  V3 pw;
  spline_pose<K>(cps + 8 * T.p_first[i], lam, &qw, &pw);
  const double* T_bs = T.sensor + 8 * T.p_sensor[i];
  double r[6];
  V3 rot;
  prior_residual(qw, pw, T_bs, T.p_meas + 7 * i, r, &rot, nullptr, nullptr);
  const So3Coef sc = so3_coef(dot(rot, rot), true);
  const M3 Jri = rodrigues_poly(rot, 0.5, sc.D);
  const M3 R_bs = qmat(Quat{T_bs[0], T_bs[1], T_bs[2], T_bs[3]});
  const M3 d_rot = scale(2.0, mul_nt(Jri, R_bs));  // 2 J_r^-1 R_bs^T
  const M3 R_wb = qmat(qw);
#pragma unroll
  for (int rr = 0; rr < 3; ++rr)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rec[6 * rr + c] = d_rot.m[3 * rr + c], rec[6 * rr + 3 + c] = 0.0;
      rec[6 * (3 + rr) + c] = 0.0, rec[6 * (3 + rr) + 3 + c] = R_wb.m[3 * rr + c];
    }
}

template <int K, int KB>
HSD void inertial_sensor_jacobians(const Tables& T, const double* cps, int i, bool robustify, double* rec) {
  const ImuParams& P = *T.imu;
  const int first = T.i_first[i];
  double u;
  segment_of(T.i_stamp[i], T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dlam[K], ddlam[K];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dlam, ddlam, 2);
  SplineFull<K> S;
  spline_full<K, false>(cps + 8 * first, lam, dlam, ddlam, &S);
  const M3 R = qmat(S.q);
  const M3 R_sb = transpose(qmat(Quat{P.T_bs[0], P.T_bs[1], P.T_bs[2], P.T_bs[3]}));
  const V3 t_bs = V3{P.T_bs[4], P.T_bs[5], P.T_bs[6]};
  const M3 I_g = lower_tri(P.i_g), I_a = lower_tri(P.i_a), S_g = colmajor3(P.S_g), X_a = colmajor3(P.X_a);
  const V3 g = V3{T.gravity[0], T.gravity[1], T.gravity[2]};
  const V3 a_i = mul_t(R, S.a - g);
  const M3 wx = hat(S.w);
  const M3 F_a = add(mul(wx, wx), hat(S.al));
  double am[3];
  const double ai[3] = {a_i.x, a_i.y, a_i.z};
#pragma unroll
  for (int r = 0; r < 3; ++r)
    am[r] = ai[r] + F_a.m[3 * r] * (X_a.m[r] + t_bs.x) + F_a.m[3 * r + 1] * (X_a.m[3 + r] + t_bs.y) + F_a.m[3 * r + 2] * (X_a.m[6 + r] + t_bs.z);
  const V3 a_m = V3{am[0], am[1], am[2]};
  const bool lit = T.inertial_literal != 0;  // inertial.cpp:155-162,189-194 as written | derivative of the prediction
  const M3 IgRsb = mul(I_g, R_sb), IaRsb = mul(I_a, R_sb), IxRsb = lit ? IgRsb : IaRsb, S_gJ = lit ? zero3() : S_g;
  const double sr = robustify ? sqrt(kScaleInertial) : 1.0;
  const M3 ang_rot = scale(2.0, mul(IgRsb, hat(S.w))), lin_rot = scale(2.0, mul(IxRsb, hat(a_m)));
  const M3 ang_tr = mul(S_gJ, F_a), lin_tr = mul(IaRsb, F_a);
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rec[6 * r + c] = sr * ang_rot.m[3 * r + c], rec[6 * r + 3 + c] = sr * ang_tr.m[3 * r + c];
      rec[6 * (3 + r) + c] = sr * lin_rot.m[3 * r + c], rec[6 * (3 + r) + 3 + c] = sr * lin_tr.m[3 * r + c];
    }
  // i_g, i_a: OrthonormalityAlignment::align parameter Jacobian of [c00 c11 c22 c10 c20 c21] at w_s / a_s (:164-174)
  const V3 w_s = mul(R_sb, S.w), a_s = mul(R_sb, a_m);
  double* Jig = rec + 36;
  double* Jia = rec + 72;
#pragma unroll
  for (int c = 0; c < 36; ++c) Jig[c] = 0.0, Jia[c] = 0.0;
  Jig[0 * 6 + 0] = sr * w_s.x, Jig[1 * 6 + 1] = sr * w_s.y, Jig[2 * 6 + 2] = sr * w_s.z;
  Jig[1 * 6 + 3] = sr * w_s.x, Jig[2 * 6 + 4] = sr * w_s.x, Jig[2 * 6 + 5] = sr * w_s.y;
  Jia[3 * 6 + 0] = sr * a_s.x, Jia[4 * 6 + 1] = sr * a_s.y, Jia[5 * 6 + 2] = sr * a_s.z;
  Jia[4 * 6 + 3] = sr * a_s.x, Jia[5 * 6 + 4] = sr * a_s.x, Jia[5 * 6 + 5] = sr * a_s.y;
  double* JS = rec + 108;
  double* JX = rec + 162;
#pragma unroll
  for (int c = 0; c < 54; ++c) JS[c] = 0.0, JX[c] = 0.0;
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 3; ++r) JS[9 * r + 3 * c + r] = sr * am[c];  // (:176-187)
#pragma unroll
  for (int col = 0; col < 3; ++col)  // X_a.col(col) enters a_m[col] through F_a.row(col) (:189-194; gyro rows through S_g)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int row = 0; row < 3; ++row) {
        JX[9 * (3 + row) + 3 * col + r] = sr * IaRsb.m[3 * row + col] * F_a.m[3 * col + r];
        JX[9 * row + 3 * col + r] = sr * S_gJ.m[3 * row + col] * F_a.m[3 * col + r];
      }
}

template <int K>
__global__ void __launch_bounds__(kBlock) k_sensor_visual(Tables T, double* out_rec, const int* out_pos, int robustify) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= T.n_vis) return;
  double rec[kSensorRecVisual];
  visual_sensor_jacobians<K>(T, T.cp, q, robustify != 0, rec);
  double* dst = out_rec + size_t(out_pos[q]) * kSensorRecVisual;
#pragma unroll
  for (int c = 0; c < kSensorRecVisual; ++c) dst[c] = rec[c];
}

template <int K>
__global__ void __launch_bounds__(kBlock) k_sensor_prior(Tables T, double* out_rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T.n_pri) return;
  double rec[kSensorRecPrior];
  prior_sensor_jacobians<K>(T, T.cp, i, rec);
  double* dst = out_rec + size_t(i) * kSensorRecPrior;
#pragma unroll
  for (int c = 0; c < kSensorRecPrior; ++c) dst[c] = rec[c];
}

template <int K, int KB>
__global__ void __launch_bounds__(kBlock) k_sensor_inertial(Tables T, double* out_rec, int robustify) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T.n_ine) return;
  inertial_sensor_jacobians<K, KB>(T, T.cp, i, robustify != 0, out_rec + size_t(i) * kSensorRecInertial);
}

}  // namespace hs
