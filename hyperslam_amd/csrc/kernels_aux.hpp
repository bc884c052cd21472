// kernels_aux.hpp — kernels outside the solve loop: trajectory sampling, track processing, state reset, manifolds (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include "kernels_common.hpp"

namespace hs {

/// Batched trajectory sampling (state.evaluate(StateQuery{t, derivative}) loop of apps/hyperslam/main.cpp:72-79):
/// pose n x 7, velocity / acceleration n x 6 [angular (body) ; linear (world)], nullable.
template <int K>
__global__ void __launch_bounds__(kBlock) k_sample_trajectory(Tables T, int n, const double* stamps, double* pose, double* vel, double* acc) {
  HS_DYNAMIC_LDS(smem);
  double* cps = smem;
  stage_cps(T.cp, cps, 8 * T.sp.n_cp);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double u;
  const int first = segment_of(stamps[i], T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dlam[K], ddlam[K];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dlam, ddlam, 2);
  SplineFull<K> S;
  spline_full<K, false>(cps + 8 * first, lam, dlam, ddlam, &S);
  double* o = pose + 7 * i;
  o[0] = S.q.x, o[1] = S.q.y, o[2] = S.q.z, o[3] = S.q.w, o[4] = S.p.x, o[5] = S.p.y, o[6] = S.p.z;
  if (vel) vel[6 * i] = S.w.x, vel[6 * i + 1] = S.w.y, vel[6 * i + 2] = S.w.z, vel[6 * i + 3] = S.v.x, vel[6 * i + 4] = S.v.y, vel[6 * i + 5] = S.v.z;
  if (acc) acc[6 * i] = S.al.x, acc[6 * i + 1] = S.al.y, acc[6 * i + 2] = S.al.z, acc[6 * i + 3] = S.a.x, acc[6 * i + 4] = S.a.y, acc[6 * i + 5] = S.a.z;
}

/// Pixel -> unit bearing in the sensor frame (radtan undistortion by fixed-point iteration; cam = [T_bs(7) | cx cy fx fy | k1 k2 p1 p2]).
HSD V3 pixel_to_bearing(const double* cam, double u, double v) {
  const double xd = (u - cam[7]) / cam[9], yd = (v - cam[8]) / cam[10];
  const double k1 = cam[11], k2 = cam[12], p1 = cam[13], p2 = cam[14];
  double x = xd, y = yd;
  for (int it = 0; it < 20; ++it) {
    const double r2 = x * x + y * y, rad = 1 + k1 * r2 + k2 * r2 * r2;
    const double dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x), dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
    x = (xd - dx) / rad, y = (yd - dy) / rad;
  }
  const double n = sqrt(x * x + y * y + 1);
  return V3{x / n, y / n, 1 / n};
}

/// AbstractOptimizer::process(VisualTracks) front half (abstract.cpp:197-223,250-255): one stereo track per lane.
template <int K>
__global__ void __launch_bounds__(kBlock) k_process_tracks(Tables T, double stamp, int n, const double* px0, const double* px1, double* b0o, double* b1o,
                                                           double* pwo) {
  HS_DYNAMIC_LDS(smem);
  double* cps = smem;
  stage_cps(T.cp, cps, 8 * T.sp.n_cp);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* c0 = T.cam, *c1 = T.cam + 16;
  const V3 b0 = pixel_to_bearing(c0, px0[2 * i], px0[2 * i + 1]), b1 = pixel_to_bearing(c1, px1[2 * i], px1[2 * i + 1]);
  if (b0o) b0o[3 * i] = b0.x, b0o[3 * i + 1] = b0.y, b0o[3 * i + 2] = b0.z;
  if (b1o) b1o[3 * i] = b1.x, b1o[3 * i + 1] = b1.y, b1o[3 * i + 2] = b1.z;
  if (!pwo) return;
  // T_wb(stamp), T_w0 = T_wb o T_b0, T_01 = T_b0^-1 o T_b1
  double u;
  const int first = segment_of(stamp, T.sp.t0, T.sp.dt, K, &u);
  double lam[K], dlam[K], ddlam[K];
  basis_weights<K>(T.basis, u, T.sp.inv_dt, lam, dlam, ddlam, 0);
  Quat q_wb;
  V3 p_wb;
  spline_pose<K>(cps + 8 * first, lam, &q_wb, &p_wb);
  const Quat q_b0 = load_quat(c0), q_b1 = load_quat(c1);
  const V3 t_b0 = V3{c0[4], c0[5], c0[6]}, t_b1 = V3{c1[4], c1[5], c1[6]};
  const M3 R_wb = qmat(q_wb), R_b0 = qmat(q_b0), R_b1 = qmat(q_b1);
  const M3 R_01 = mul_tn(R_b0, R_b1);
  const V3 o = mul_t(R_b0, t_b1 - t_b0);  // origin of camera 1 in frame 0
  const V3 d1 = mul(R_01, b1);
  const double a = dot(b0, b0), b = dot(b0, d1), c = dot(d1, d1), e = dot(b0, o), f = dot(d1, o);
  const double den = a * c - b * b;
  const double s0 = den > 1e-12 ? (c * e - b * f) / den : 1.0, s1 = den > 1e-12 ? (b * e - a * f) / den : 1.0;
  const V3 p0 = 0.5 * (s0 * b0 + o + s1 * d1);  // midpoint of the two rays, frame 0
  const V3 pb = mul(R_b0, p0) + t_b0;
  const V3 pw = mul(R_wb, pb) + p_wb;
  pwo[3 * i] = pw.x, pwo[3 * i + 1] = pw.y, pwo[3 * i + 2] = pw.z;
}

/// Fresh trust-region state (LevenbergMarquardtStrategy: initial radius 1e4, decrease factor 2).
__global__ void k_reset_state(DevState* st, int max_iterations, double radius, int spec) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st->radius = radius, st->decrease_factor = 2.0;
  st->cost = st->cand_cost = st->model_cost_change = 0.0;
  st->gmax_bits = 0ull, st->gmax_pose_bits = 0ull, st->gmax = 0.0, st->x_sqnorm = st->step_sqnorm = 0.0;
  st->g_dot_step_pose = st->d2_step2_pose = st->g_dot_step_far = st->d2_step2_far = 0.0;
  st->iteration = 0, st->done = 0, st->termination = HS_NO_CONVERGENCE, st->accepted = 0, st->step_valid = 0;
  st->invalid_streak = 0, st->num_successful = 0, st->num_iterations = 0, st->scaling_ready = 0;
  st->max_iterations = max_iterations, st->chol_failed = 0;
  st->spec = spec, st->rec_sel = 0, st->rec_pending = 0;
}

/// Batched Manifold::Plus / PlusJacobian of the variable classes on the path (hs_manifold_plus*): the same device
/// functions k_backsub_retract and the local-coordinate Jacobians use. One element per lane. kind: HS_MANIFOLD_* of the C ABI.
__global__ void __launch_bounds__(kBlock) k_manifold_plus(int kind, int ambient, int tangent, int n, const double* __restrict__ x,
                                                          const double* __restrict__ d, double* __restrict__ out, double* __restrict__ jac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* xi = x + size_t(i) * ambient;
  if (out) {
    const double* di = d + size_t(i) * tangent;
    double* o = out + size_t(i) * ambient;
    switch (kind) {
      case 1:
        for (int c = 0; c < ambient; ++c) o[c] = xi[c] + di[c];
        break;
      case 2:
      case 3: {
        const Quat q = quat_plus(Quat{xi[0], xi[1], xi[2], xi[3]}, V3{di[0], di[1], di[2]});
        o[0] = q.x, o[1] = q.y, o[2] = q.z, o[3] = q.w;
        o[4] = xi[4] + di[3], o[5] = xi[5] + di[4], o[6] = xi[6] + di[5];
        if (kind == 2) o[7] = xi[7];
        break;
      }
      case 4: sphere_plus(xi, di, o); break;
      case 5:
        o[0] = xi[0] + di[0], o[1] = xi[1] + di[1], o[2] = xi[2] + di[2], o[3] = xi[3];
        break;
      default:
        for (int c = 0; c < ambient; ++c) o[c] = xi[c];
    }
  }
  if (jac && tangent > 0) {
    double* J = jac + size_t(i) * ambient * tangent;
    for (int e = 0; e < ambient * tangent; ++e) J[e] = 0.0;
    switch (kind) {
      case 1:
        for (int c = 0; c < ambient; ++c) J[c * tangent + c] = 1.0;
        break;
      case 2:
      case 3: {
        const Quat q = Quat{xi[0], xi[1], xi[2], xi[3]};
        // column c = d/d delta_c of [delta ; 1] (x) q at delta = 0 = (e_c, 0) (x) q
        const Quat c0 = qmul(Quat{1, 0, 0, 0}, q), c1 = qmul(Quat{0, 1, 0, 0}, q), c2 = qmul(Quat{0, 0, 1, 0}, q);
        J[0 * 6 + 0] = c0.x, J[1 * 6 + 0] = c0.y, J[2 * 6 + 0] = c0.z, J[3 * 6 + 0] = c0.w;
        J[0 * 6 + 1] = c1.x, J[1 * 6 + 1] = c1.y, J[2 * 6 + 1] = c1.z, J[3 * 6 + 1] = c1.w;
        J[0 * 6 + 2] = c2.x, J[1 * 6 + 2] = c2.y, J[2 * 6 + 2] = c2.z, J[3 * 6 + 2] = c2.w;
        J[4 * 6 + 3] = 1.0, J[5 * 6 + 4] = 1.0, J[6 * 6 + 5] = 1.0;
        break;
      }
      case 4: sphere_plus_jacobian(xi, J); break;
      case 5: J[0 * 3 + 0] = 1.0, J[1 * 3 + 1] = 1.0, J[2 * 3 + 2] = 1.0; break;
      default: break;
    }
  }
}

/// Batched Manifold::Minus / MinusJacobian (hs_manifold_minus*, wrapper.hpp:44-50). One element per lane; jac is tangent x ambient.
__global__ void __launch_bounds__(kBlock) k_manifold_minus(int kind, int ambient, int tangent, int n, const double* __restrict__ y,
                                                           const double* __restrict__ x, double* __restrict__ out, double* __restrict__ jac) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* xi = x + size_t(i) * ambient;
  if (out) {
    const double* yi = y + size_t(i) * ambient;
    double* o = out + size_t(i) * tangent;
    switch (kind) {
      case 1:
        for (int c = 0; c < ambient; ++c) o[c] = yi[c] - xi[c];
        break;
      case 2:
      case 3: {
        const V3 d = quat_minus(Quat{yi[0], yi[1], yi[2], yi[3]}, Quat{xi[0], xi[1], xi[2], xi[3]});
        o[0] = d.x, o[1] = d.y, o[2] = d.z;
        o[3] = yi[4] - xi[4], o[4] = yi[5] - xi[5], o[5] = yi[6] - xi[6];
        break;
      }
      case 4: sphere_minus(yi, xi, o); break;
      case 5: o[0] = yi[0] - xi[0], o[1] = yi[1] - xi[1], o[2] = yi[2] - xi[2]; break;
      default: break;
    }
  }
  if (jac) {
    double* J = jac + size_t(i) * ambient * tangent;
    for (int e = 0; e < ambient * tangent; ++e) J[e] = 0.0;
    switch (kind) {
      case 1:
        for (int c = 0; c < ambient; ++c) J[c * ambient + c] = 1.0;
        break;
      case 2:
      case 3: {
        // row c = d/dy of entry c of the vector part of y (x) conj(x) (= the transpose of the quaternion's PlusJacobian)
        const double qx = xi[0], qy = xi[1], qz = xi[2], qw = xi[3];
        J[0 * ambient + 0] = qw, J[0 * ambient + 1] = -qz, J[0 * ambient + 2] = qy, J[0 * ambient + 3] = -qx;
        J[1 * ambient + 0] = qz, J[1 * ambient + 1] = qw, J[1 * ambient + 2] = -qx, J[1 * ambient + 3] = -qy;
        J[2 * ambient + 0] = -qy, J[2 * ambient + 1] = qx, J[2 * ambient + 2] = qw, J[2 * ambient + 3] = -qz;
        J[3 * ambient + 4] = 1.0, J[4 * ambient + 5] = 1.0, J[5 * ambient + 6] = 1.0;
        break;
      }
      case 4: sphere_minus_jacobian(xi, J); break;
      case 5: J[0 * 4 + 0] = 1.0, J[1 * 4 + 1] = 1.0, J[2 * 4 + 2] = 1.0; break;
      default: break;
    }
  }
}

}  // namespace hs
