// ORACLE — TEST INFRASTRUCTURE ONLY (see hs_math.hpp header). PARITY UNPINNED.
//
// hs_problem.hpp: window tables + CPU restatement of CeresOptimizer::optimize()
//   (internal/hyper/optimizers/ceres/optimizer.cpp:38-54 options, :276-280 optimize) with Ceres' trust-region
//   Levenberg-Marquardt semantics (third-party) — Jacobi scaling computed at iteration 0, LM diagonal
//   clamp(diag(J'J), 1e-6, 1e32)/radius, step quality, radius update, function/parameter/gradient tolerances,
//   max_num_iterations = 5.  The linear solve eliminates landmarks by Schur complement and factors the reduced
//   system densely (algebraically identical to SPARSE_NORMAL_CHOLESKY on the full J'J).
#pragma once
#include <chrono>
#include <algorithm>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "hs_factors.hpp"

namespace hso {

struct Problem {
  // Spline: control point j = [qx qy qz qw px py pz t], t = t0 + j*dt.
  int k = 4;
  double t0 = 0, dt = 0.1;
  int n_cp = 0;
  std::vector<double> cp;
  std::vector<uint8_t> cp_const;  // optimizer.cpp:323-328 (frozen control points)
  bool rot_const = false, trans_const = false;  // backend.cpp:52-55

  // Cameras (constant blocks by default, camera.hpp:18).
  int n_cam = 0;
  std::vector<double> cam_T_bs, cam_intr, cam_dist;

  // Plain sensors for pose priors.
  int n_sensor = 0;
  std::vector<double> sensor_T_bs;

  // Landmarks (optimizer.cpp:347-358).
  int n_lm = 0;
  std::vector<double> lm;
  std::vector<uint8_t> lm_const;

  // IMU (single; inertial.cpp:35-42) + bias splines (free, optimizer.cpp:62-63) + gravity.
  bool has_imu = false;
  double imu_T_bs[7] = {0, 0, 0, 1, 0, 0, 0}, imu_i_g[6] = {1, 1, 1, 0, 0, 0}, imu_i_a[6] = {1, 1, 1, 0, 0, 0};
  double imu_S_g[9] = {0}, imu_X_a[9] = {0};
  int kb = 4;
  double bias_t0 = 0, bias_dt = 1.0;
  int n_bias = 0;
  std::vector<double> bias_g, bias_a;  // 4 doubles each [x y z t]
  bool bias_const = false;
  double gravity[3] = {0, 0, -9.80665};
  bool gravity_const = true;  // abstract.cpp:57-61

  // Residual tables.
  std::vector<double> px_stamp, px_meas;
  std::vector<int32_t> px_lm, px_cam;
  std::vector<double> br_stamp, br_meas;
  std::vector<int32_t> br_lm, br_cam;
  std::vector<double> pr_stamp, pr_meas;
  std::vector<int32_t> pr_sensor;
  std::vector<double> in_stamp, in_meas;
  int inertial_mode = 0;  // HS_INERTIAL_AS_REFERENCE (inertial.cpp as written) | 1 HS_INERTIAL_EXACT
  std::vector<double> weights[4];  // CostConfiguration::weights per factor type (empty: none)

  int n_res(FactorType t) const {
    switch (t) {
      case kPixel: return int(px_stamp.size());
      case kBearing: return int(br_stamp.size());
      case kPrior: return int(pr_stamp.size());
      case kInertial: return int(in_stamp.size());
    }
    return 0;
  }
  // Reduced ("pose-side") unknown layout: [cp 6 each | gyro bias 3 each | accel bias 3 each | gravity 2].
  int dim_pose() const { return 6 * n_cp + (has_imu ? 6 * n_bias + 2 : 0); }
  int off_bias_g() const { return 6 * n_cp; }
  int off_bias_a() const { return 6 * n_cp + 3 * n_bias; }
  int off_gravity() const { return 6 * n_cp + 6 * n_bias; }
};

/// One linearised residual block in local coordinates.
struct Linearized {
  int n_res = 0;
  double r[6];
  double cost = 0;   // 0.5 * rho(|r|^2)
  int first_cp = 0;  // state columns: 6*k starting at 6*first_cp
  std::vector<double> J_state;  // n_res x 6k row-major (zero columns for constant control points)
  int lm = -1;
  double J_lm[6 * 3];
  // inertial extras
  int first_bias = 0;
  std::vector<double> J_bias_g, J_bias_a;  // n_res x 3kb
  double J_grav[6 * 2];
  // sensor parameter blocks (only with want_sensor; constant in the solver): local Jacobians, n_res x local size
  double J_ext[6 * 6];                         // T_bs  [d_rot d_trans]
  double J_intr[2 * 4], J_dist[2 * 4];         // pixel
  double J_ig[6 * 6], J_ia[6 * 6], J_Sg[6 * 9], J_Xa[6 * 9];  // inertial
};

struct Evaluator {
  const Problem& P;
  Basis basis, bias_basis;
  explicit Evaluator(const Problem& p) : P(p), basis(make_basis(p.k)), bias_basis(make_basis(p.kb)) {}

  /// Gathers parameter-block pointers exactly in ExteroceptiveCost::update order and evaluates residual idx of
  /// `type`. If `lin` is null only the (uncorrected) residual and cost are produced.
  /// raw_r (optional) receives the un-robustified residual.
  void evaluate(FactorType type, int idx, bool robustify, Linearized* lin, double* raw_r = nullptr, double* cost = nullptr,
                bool want_jac = true, bool want_sensor = false) const {
    const int k = P.k;
    const Layout L = make_layout(type, k, P.kb);
    std::vector<const double*> params(L.sizes.size());
    std::vector<ManifoldKind> kinds(L.sizes.size(), kManifoldConstant);
    double stamp = 0;
    const double* meas = nullptr;
    int lm = -1;
    switch (type) {
      case kPixel: stamp = P.px_stamp[idx], meas = &P.px_meas[2 * idx], lm = P.px_lm[idx]; break;
      case kBearing: stamp = P.br_stamp[idx], meas = &P.br_meas[3 * idx], lm = P.br_lm[idx]; break;
      case kPrior: stamp = P.pr_stamp[idx], meas = &P.pr_meas[7 * idx]; break;
      case kInertial: stamp = P.in_stamp[idx], meas = &P.in_meas[6 * idx]; break;
    }
    double u;
    const int first = segment_of(stamp, P.t0, P.dt, k, &u);
    for (int j = 0; j < k; ++j) {
      params[j] = &P.cp[8 * (first + j)];
      kinds[j] = P.cp_const[first + j] ? kManifoldConstant : kManifoldControlPoint;
    }
    int first_bias = 0;
    if (type == kPixel || type == kBearing) {
      const int cam = (type == kPixel) ? P.px_cam[idx] : P.br_cam[idx];
      params[k + 0] = &P.cam_T_bs[7 * cam], params[k + 1] = &P.cam_intr[4 * cam], params[k + 2] = &P.cam_dist[4 * cam];
      params[k + 3] = &P.lm[3 * lm];
      kinds[k + 3] = P.lm_const[lm] ? kManifoldConstant : kManifoldEuclidean;
    } else if (type == kPrior) {
      params[k + 0] = &P.sensor_T_bs[7 * P.pr_sensor[idx]];
    } else {
      params[k + 0] = P.imu_T_bs, params[k + 1] = P.imu_i_g, params[k + 2] = P.imu_i_a, params[k + 3] = P.imu_S_g, params[k + 4] = P.imu_X_a;
      double ub;
      first_bias = segment_of(stamp, P.bias_t0, P.bias_dt, P.kb, &ub);
      for (int j = 0; j < P.kb; ++j) {
        params[k + 5 + j] = &P.bias_g[4 * (first_bias + j)];
        params[k + 5 + P.kb + j] = &P.bias_a[4 * (first_bias + j)];
        kinds[k + 5 + j] = kinds[k + 5 + P.kb + j] = P.bias_const ? kManifoldConstant : kManifoldBiasPoint;
      }
      params[k + 5 + 2 * P.kb] = P.gravity;
      kinds[k + 5 + 2 * P.kb] = P.gravity_const ? kManifoldConstant : kManifoldSphere3;
    }
    if (want_sensor) {  // as if the sensor blocks were variable (sensors/sensor.cpp:26-29 manifolds)
      const int n_static = type == kInertial ? 5 : (type == kPrior ? 1 : 3);
      kinds[k] = kManifoldSE3;
      for (int b = 1; b < n_static; ++b) kinds[k + b] = kManifoldEuclidean;
    }
    const CostContext ctx = {type, &basis, &bias_basis, stamp, meas, P.inertial_mode == 0, P.weights[type].empty() ? nullptr : P.weights[type].data()};
    const Loss loss = loss_for(type);
    double r[6];
    if (!lin || !want_jac) {
      cost_evaluate(ctx, L, params.data(), r, nullptr);
      double s = 0;
      for (int i = 0; i < L.num_residuals; ++i) s += r[i] * r[i];
      double rho[3];
      loss_evaluate(loss, s, rho);
      if (raw_r)
        for (int i = 0; i < L.num_residuals; ++i) raw_r[i] = r[i];
      if (cost) *cost = 0.5 * rho[0];
      if (lin) {
        lin->n_res = L.num_residuals, lin->cost = 0.5 * rho[0];
        for (int i = 0; i < L.num_residuals; ++i) lin->r[i] = r[i];
      }
      return;
    }
    // Jacobian buffers: null for constant blocks (what Ceres passes for constant parameter blocks).
    std::vector<std::vector<double>> jbuf(L.sizes.size());
    std::vector<double*> jac(L.sizes.size(), nullptr);
    for (size_t i = 0; i < L.sizes.size(); ++i)
      if (kinds[i] != kManifoldConstant) {
        jbuf[i].assign(size_t(L.num_residuals) * L.sizes[i], 0.0);
        jac[i] = jbuf[i].data();
      }
    cost_evaluate(ctx, L, params.data(), r, jac.data());
    const int n = L.num_residuals;
    double s = 0;
    for (int i = 0; i < n; ++i) s += r[i] * r[i];
    double rho[3];
    loss_evaluate(loss, s, rho);
    const double scale = robustify ? corrector_scale(rho) : 1.0;
    lin->n_res = n, lin->cost = 0.5 * rho[0], lin->first_cp = first, lin->lm = lm, lin->first_bias = first_bias;
    if (raw_r)
      for (int i = 0; i < n; ++i) raw_r[i] = r[i];
    if (cost) *cost = lin->cost;
    for (int i = 0; i < n; ++i) lin->r[i] = scale * r[i];
    lin->J_state.assign(size_t(n) * 6 * k, 0.0);
    double Jl[6 * 6];
    for (int j = 0; j < k; ++j) {
      if (!jac[j]) continue;
      to_local(kManifoldControlPoint, 8, n, params[j], jac[j], Jl);
      for (int rr = 0; rr < n; ++rr)
        for (int c = 0; c < 6; ++c) {
          const bool frozen = (c < 3) ? P.rot_const : P.trans_const;
          lin->J_state[size_t(rr) * 6 * k + 6 * j + c] = frozen ? 0.0 : scale * Jl[rr * 6 + c];
        }
    }
    if (want_sensor) {
      auto local = [&](int b, ManifoldKind kind, int ambient, double* dst) {
        to_local(kind, ambient, n, params[b], jac[b], dst);
        for (int i = 0; i < n * manifold_local_size(kind, ambient); ++i) dst[i] *= scale;
      };
      local(k, kManifoldSE3, 7, lin->J_ext);
      if (type == kPixel) local(k + 1, kManifoldEuclidean, 4, lin->J_intr), local(k + 2, kManifoldEuclidean, 4, lin->J_dist);
      if (type == kInertial) {
        local(k + 1, kManifoldEuclidean, 6, lin->J_ig), local(k + 2, kManifoldEuclidean, 6, lin->J_ia);
        local(k + 3, kManifoldEuclidean, 9, lin->J_Sg), local(k + 4, kManifoldEuclidean, 9, lin->J_Xa);
      }
    }
    for (int i = 0; i < 18; ++i) lin->J_lm[i] = 0;
    if (lm >= 0 && jac[k + 3]) {
      to_local(kManifoldEuclidean, 3, n, params[k + 3], jac[k + 3], Jl);
      for (int i = 0; i < n * 3; ++i) lin->J_lm[i] = scale * Jl[i];
    }
    if (type == kInertial) {
      lin->J_bias_g.assign(size_t(n) * 3 * P.kb, 0.0);
      lin->J_bias_a.assign(size_t(n) * 3 * P.kb, 0.0);
      for (int j = 0; j < P.kb; ++j) {
        if (jac[k + 5 + j]) {
          to_local(kManifoldBiasPoint, 4, n, params[k + 5 + j], jac[k + 5 + j], Jl);
          for (int rr = 0; rr < n; ++rr)
            for (int c = 0; c < 3; ++c) lin->J_bias_g[size_t(rr) * 3 * P.kb + 3 * j + c] = scale * Jl[rr * 3 + c];
        }
        if (jac[k + 5 + P.kb + j]) {
          to_local(kManifoldBiasPoint, 4, n, params[k + 5 + P.kb + j], jac[k + 5 + P.kb + j], Jl);
          for (int rr = 0; rr < n; ++rr)
            for (int c = 0; c < 3; ++c) lin->J_bias_a[size_t(rr) * 3 * P.kb + 3 * j + c] = scale * Jl[rr * 3 + c];
        }
      }
      for (int i = 0; i < 12; ++i) lin->J_grav[i] = 0;
      const int og = k + 5 + 2 * P.kb;
      if (jac[og]) {
        to_local(kManifoldSphere3, 3, n, params[og], jac[og], Jl);
        for (int i = 0; i < n * 2; ++i) lin->J_grav[i] = scale * Jl[i];
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------------------
// Normal equations with landmark Schur complement.
// ---------------------------------------------------------------------------------------------------------
struct NormalEquations {
  int np = 0, nl = 0;
  std::vector<double> Hpp, gp;  // np x np (full symmetric), np          (unscaled J'J, J'r)
  std::vector<double> Hll, bl;  // nl x 9, nl x 3
  std::vector<int> w_first, w_count;  // per landmark: first cp and number of cps touched
  std::vector<std::vector<double>> W;  // per landmark: (6*count) x 3 row-major  (H_pl rows of its cp range)
  double cost = 0;
  // The linearised blocks themselves, in evaluation order (Ceres keeps the Jacobian of the current point and reuses it for
  // the model cost change of every candidate step, trust_region_minimizer.cc).
  std::vector<FactorType> block_type;
  std::vector<Linearized> blocks;
};

struct Solver {
  Problem& P;
  explicit Solver(Problem& p) : P(p) {}

  static const FactorType* types() {
    static const FactorType t[4] = {kPixel, kBearing, kPrior, kInertial};
    return t;
  }

  double total_cost() const {
    Evaluator ev(P);
    double c = 0;
    for (int ti = 0; ti < 4; ++ti)
      for (int i = 0; i < P.n_res(types()[ti]); ++i) {
        double ci;
        ev.evaluate(types()[ti], i, true, nullptr, nullptr, &ci, false);
        c += ci;
      }
    return c;
  }

  /// Scatter list of one linearised residual: pose-side column indices and values per residual row.
  void pose_columns(const Linearized& lin, FactorType type, std::vector<int>* cols, std::vector<double>* vals) const {
    const int k = P.k, n = lin.n_res;
    cols->clear();
    vals->clear();
    for (int c = 0; c < 6 * k; ++c) cols->push_back(6 * lin.first_cp + c);
    if (type == kInertial) {
      for (int c = 0; c < 3 * P.kb; ++c) cols->push_back(P.off_bias_g() + 3 * lin.first_bias + c);
      for (int c = 0; c < 3 * P.kb; ++c) cols->push_back(P.off_bias_a() + 3 * lin.first_bias + c);
      for (int c = 0; c < 2; ++c) cols->push_back(P.off_gravity() + c);
    }
    const int nc = int(cols->size());
    vals->assign(size_t(n) * nc, 0.0);
    for (int r = 0; r < n; ++r) {
      double* v = &(*vals)[size_t(r) * nc];
      for (int c = 0; c < 6 * k; ++c) v[c] = lin.J_state[size_t(r) * 6 * k + c];
      if (type == kInertial) {
        int o = 6 * k;
        for (int c = 0; c < 3 * P.kb; ++c) v[o + c] = lin.J_bias_g[size_t(r) * 3 * P.kb + c];
        o += 3 * P.kb;
        for (int c = 0; c < 3 * P.kb; ++c) v[o + c] = lin.J_bias_a[size_t(r) * 3 * P.kb + c];
        o += 3 * P.kb;
        for (int c = 0; c < 2; ++c) v[o + c] = lin.J_grav[r * 2 + c];
      }
    }
  }

  void build(NormalEquations* ne) const {
    Evaluator ev(P);
    const int np = P.dim_pose(), nl = P.n_lm;
    ne->np = np, ne->nl = nl, ne->cost = 0;
    ne->Hpp.assign(size_t(np) * np, 0.0);
    ne->gp.assign(np, 0.0);
    ne->Hll.assign(size_t(nl) * 9, 0.0);
    ne->bl.assign(size_t(nl) * 3, 0.0);
    // landmark cp ranges
    ne->w_first.assign(nl, 1 << 30);
    std::vector<int> w_last(nl, -1);
    for (int ti = 0; ti < 2; ++ti) {
      const FactorType t = types()[ti];
      for (int i = 0; i < P.n_res(t); ++i) {
        const double st = (t == kPixel) ? P.px_stamp[i] : P.br_stamp[i];
        const int l = (t == kPixel) ? P.px_lm[i] : P.br_lm[i];
        double u;
        const int f = segment_of(st, P.t0, P.dt, P.k, &u);
        ne->w_first[l] = std::min(ne->w_first[l], f);
        w_last[l] = std::max(w_last[l], f + P.k - 1);
      }
    }
    ne->w_count.assign(nl, 0);
    ne->W.assign(nl, {});
    for (int l = 0; l < nl; ++l)
      if (w_last[l] >= 0) {
        ne->w_count[l] = w_last[l] - ne->w_first[l] + 1;
        ne->W[l].assign(size_t(6) * ne->w_count[l] * 3, 0.0);
      }
    std::vector<int> cols;
    std::vector<double> vals;
    size_t n_blocks = 0;
    for (int ti = 0; ti < 4; ++ti) n_blocks += size_t(P.n_res(types()[ti]));
    ne->block_type.resize(n_blocks);
    ne->blocks.resize(n_blocks);
    size_t bi = 0;
    for (int ti = 0; ti < 4; ++ti) {
      const FactorType t = types()[ti];
      for (int i = 0; i < P.n_res(t); ++i, ++bi) {
        Linearized& lin = ne->blocks[bi];
        ne->block_type[bi] = t;
        ev.evaluate(t, i, true, &lin);
        ne->cost += lin.cost;
        pose_columns(lin, t, &cols, &vals);
        const int nc = int(cols.size()), n = lin.n_res;
        for (int r = 0; r < n; ++r) {
          const double* v = &vals[size_t(r) * nc];
          for (int a = 0; a < nc; ++a) {
            if (v[a] == 0.0) continue;
            ne->gp[cols[a]] += v[a] * lin.r[r];
            double* row = &ne->Hpp[size_t(cols[a]) * np];
            for (int b = 0; b < nc; ++b) row[cols[b]] += v[a] * v[b];
          }
        }
        if (lin.lm >= 0) {
          const int l = lin.lm;
          for (int r = 0; r < n; ++r) {
            const double* jl = &lin.J_lm[r * 3];
            for (int a = 0; a < 3; ++a) {
              ne->bl[3 * l + a] += jl[a] * lin.r[r];
              for (int b = 0; b < 3; ++b) ne->Hll[9 * l + 3 * a + b] += jl[a] * jl[b];
            }
            const int row0 = 6 * (lin.first_cp - ne->w_first[l]);
            for (int c = 0; c < 6 * P.k; ++c) {
              const double v = lin.J_state[size_t(r) * 6 * P.k + c];
              for (int b = 0; b < 3; ++b) ne->W[l][size_t(row0 + c) * 3 + b] += v * jl[b];
            }
          }
        }
      }
    }
  }
};

// Row envelope of a symmetric matrix stored dense: first[i] = first column of row i (lower triangle) holding a non-zero.
// The Cholesky factor fills only inside the envelope, so the products skipped below are exact zeros and the results are
// those of the plain dense algorithm; the reduced system is block-banded (plus border rows for bias splines / gravity),
// which is what the reference's sparse Cholesky exploits (optimizer.cpp:42-43).
inline std::vector<int> row_envelope(const std::vector<double>& A, int n) {
  std::vector<int> first(n);
  for (int i = 0; i < n; ++i) {
    int f = 0;
    while (f < i && A[size_t(i) * n + f] == 0.0) ++f;
    first[i] = f;
  }
  return first;
}
// Cholesky (lower) in place inside the envelope; returns false if not positive definite.
inline bool cholesky_lower(std::vector<double>& A, int n, const std::vector<int>& first) {
  for (int j = 0; j < n; ++j) {
    const double* rj = &A[size_t(j) * n];
    double d = rj[j];
    for (int k = first[j]; k < j; ++k) d -= rj[k] * rj[k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[size_t(j) * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      if (first[i] > j) continue;
      const double* ri = &A[size_t(i) * n];
      double s = ri[j];
      for (int k = std::max(first[i], first[j]); k < j; ++k) s -= ri[k] * rj[k];
      A[size_t(i) * n + j] = s / d;
    }
  }
  return true;
}
inline void cholesky_solve(const std::vector<double>& L, int n, const std::vector<int>& first, std::vector<double>& b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = first[i]; k < i; ++k) s -= L[size_t(i) * n + k] * b[k];
    b[i] = s / L[size_t(i) * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k)
      if (first[k] <= i) s -= L[size_t(k) * n + i] * b[k];
    b[i] = s / L[size_t(i) * n + i];
  }
}
/// Inverse of a symmetric positive definite 3x3 block through its Cholesky factor A = L L', A^-1 = L^-T L^-1: backward stable for
/// the landmark blocks of poorly observed points (two views, little parallax: cond(V) ~ 1e8). The cofactor / determinant formula
/// this replaced lost those digits to cancellation: with it the double build of this file stood 1.4e-6 away (landmarks, stereo
/// replay call 32) from its own long-double build (capi_ld.cpp) and from the HIP library alike, which agree with each other to
/// 4e-10 (profiles/r05_three_way_3_6_0_4.txt). The reference never forms this inverse: its linear solver is
/// SPARSE_NORMAL_CHOLESKY on the full system (optimizer.cpp:28-54), the landmark elimination is this path's own formulation.
inline bool inv3_spd(const double* A, double* inv) {
  if (!(A[0] > 0.0)) return false;
  const double l00 = std::sqrt(A[0]), l10 = A[1] / l00, l20 = A[2] / l00;
  const double d1 = A[4] - l10 * l10;
  if (!(d1 > 0.0)) return false;
  const double l11 = std::sqrt(d1), l21 = (A[5] - l20 * l10) / l11;
  const double d2 = A[8] - l20 * l20 - l21 * l21;
  if (!(d2 > 0.0)) return false;
  const double l22 = std::sqrt(d2);
  const double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22;  // M = L^-1 (lower triangular)
  const double m10 = -l10 * m00 * m11, m21 = -l21 * m11 * m22, m20 = -(l20 * m00 + l21 * m10) * m22;
  inv[0] = m00 * m00 + m10 * m10 + m20 * m20, inv[1] = m10 * m11 + m20 * m21, inv[2] = m20 * m22;
  inv[3] = inv[1], inv[4] = m11 * m11 + m21 * m21, inv[5] = m21 * m22;
  inv[6] = inv[2], inv[7] = inv[5], inv[8] = m22 * m22;
  return true;
}

struct IterationRecord {
  int iteration;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius;
  int step_is_valid, step_is_successful;
};
struct Summary {
  double initial_cost = 0, final_cost = 0;
  int num_iterations = 0;       // linear solves attempted (Ceres iterations beyond iteration 0)
  int num_successful_steps = 0;
  int termination = 0;          // 0 NO_CONVERGENCE (max iterations), 1 CONVERGENCE, 2 FAILURE
  std::vector<IterationRecord> iterations;
  // host clocks per stage, summed over the iterations (SURVEY.md §8d per-stage CPU times): linearise (residuals, Jacobians and the
  // accumulation of J'J / J'r, one pass as in Ceres' evaluator), Schur complement (reduced system), factor + solves, update
  // (retraction, cost at the candidate, step decision)
  double linearize_ms = 0, schur_ms = 0, solve_ms = 0, update_ms = 0;
};

/// Scaled + damped reduced system at the current point (what one LM iteration factors):
///   S = Dp (Hpp) Dp + Dlm_p - sum_l Ws_l V_l^-1 Ws_l',   g = Dp gp - sum_l Ws_l V_l^-1 Dl bl
struct ReducedSystem {
  int np = 0;
  std::vector<double> S, g;
  std::vector<double> Vinv;  // nl x 9
};

struct LM {
  Problem& P;
  Solver solver;
  std::vector<double> scale_p, scale_l;  // Jacobi scaling (TrustRegionMinimizer, jacobi_scaling = true)
  double radius = 1e4, decrease_factor = 2.0;
  static constexpr double kMinDiag = 1e-6, kMaxDiag = 1e32;
  // Residual-sharded operation: every additive quantity goes through this sum-all-reduce (no-op when unset).
  std::function<void(double*, int64_t)> allreduce;
  int rank = 0, world = 1;
  explicit LM(Problem& p) : P(p), solver(p) {}
  void sum(std::vector<double>& v) const {
    if (allreduce && !v.empty()) allreduce(v.data(), int64_t(v.size()));
  }
  /// Makes the pose-side normal equations and the cost global (landmark blocks stay local to their shard).
  void globalize(NormalEquations* ne) const {
    if (!allreduce) return;
    std::vector<double> buf(ne->Hpp);
    buf.insert(buf.end(), ne->gp.begin(), ne->gp.end());
    buf.push_back(ne->cost);
    sum(buf);
    const size_t n2 = ne->Hpp.size();
    std::copy(buf.begin(), buf.begin() + n2, ne->Hpp.begin());
    std::copy(buf.begin() + n2, buf.begin() + n2 + ne->gp.size(), ne->gp.begin());
    ne->cost = buf.back();
  }

  std::vector<uint8_t> active_pose_mask(const NormalEquations& ne) const {
    // A pose-side coordinate is active iff its column of J is not structurally zero (constant blocks / never observed).
    std::vector<uint8_t> m(ne.np, 0);
    for (int i = 0; i < ne.np; ++i) m[i] = ne.Hpp[size_t(i) * ne.np + i] > 0.0;
    return m;
  }

  void compute_scaling(const NormalEquations& ne) {
    scale_p.assign(ne.np, 1.0);
    scale_l.assign(size_t(ne.nl) * 3, 1.0);
    for (int i = 0; i < ne.np; ++i) scale_p[i] = 1.0 / (1.0 + std::sqrt(ne.Hpp[size_t(i) * ne.np + i]));
    for (int l = 0; l < ne.nl; ++l)
      for (int a = 0; a < 3; ++a) scale_l[3 * l + a] = 1.0 / (1.0 + std::sqrt(ne.Hll[9 * l + 4 * a]));
  }

  void reduce(const NormalEquations& ne, ReducedSystem* rs) const {
    const int np = ne.np, nl = ne.nl;
    rs->np = np;
    rs->S.assign(size_t(np) * np, 0.0);
    rs->g.assign(np, 0.0);
    rs->Vinv.assign(size_t(nl) * 9, 0.0);
    for (int i = 0; i < np; ++i) {
      for (int j = 0; j < np; ++j) rs->S[size_t(i) * np + j] = scale_p[i] * ne.Hpp[size_t(i) * np + j] * scale_p[j];
      rs->g[i] = scale_p[i] * ne.gp[i];
      const double diag = std::min(std::max(rs->S[size_t(i) * np + i], kMinDiag), kMaxDiag);
      rs->S[size_t(i) * np + i] += diag / radius;
    }
    std::vector<double> schur(size_t(np) * np + np, 0.0);  // [dS | dg]: this shard's landmark eliminations (additive)
    double* dS = schur.data();
    double* dg = schur.data() + size_t(np) * np;
    for (int l = 0; l < nl; ++l) {
      if (ne.w_count[l] == 0 || P.lm_const[l]) continue;
      double V[9];
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) V[3 * a + b] = scale_l[3 * l + a] * ne.Hll[9 * l + 3 * a + b] * scale_l[3 * l + b];
      for (int a = 0; a < 3; ++a) V[4 * a] += std::min(std::max(V[4 * a], kMinDiag), kMaxDiag) / radius;
      double* Vi = &rs->Vinv[9 * l];
      inv3_spd(V, Vi);
      const int rows = 6 * ne.w_count[l], r0 = 6 * ne.w_first[l];
      std::vector<double> Ws(size_t(rows) * 3), WV(size_t(rows) * 3);
      for (int r = 0; r < rows; ++r)
        for (int b = 0; b < 3; ++b) Ws[size_t(r) * 3 + b] = scale_p[r0 + r] * ne.W[l][size_t(r) * 3 + b] * scale_l[3 * l + b];
      for (int r = 0; r < rows; ++r)
        for (int b = 0; b < 3; ++b) {
          double s = 0;
          for (int c = 0; c < 3; ++c) s += Ws[size_t(r) * 3 + c] * Vi[3 * c + b];
          WV[size_t(r) * 3 + b] = s;
        }
      double sb[3];
      for (int a = 0; a < 3; ++a) sb[a] = scale_l[3 * l + a] * ne.bl[3 * l + a];
      for (int r = 0; r < rows; ++r) {
        dg[r0 + r] -= WV[size_t(r) * 3] * sb[0] + WV[size_t(r) * 3 + 1] * sb[1] + WV[size_t(r) * 3 + 2] * sb[2];
        double* Srow = &dS[size_t(r0 + r) * np + r0];
        for (int c = 0; c < rows; ++c)
          Srow[c] -= WV[size_t(r) * 3] * Ws[size_t(c) * 3] + WV[size_t(r) * 3 + 1] * Ws[size_t(c) * 3 + 1] + WV[size_t(r) * 3 + 2] * Ws[size_t(c) * 3 + 2];
      }
    }
    sum(schur);
    for (size_t e = 0; e < size_t(np) * np; ++e) rs->S[e] += dS[e];
    for (int i = 0; i < np; ++i) rs->g[i] += dg[i];
  }

  /// Solves the LM system; outputs the *scaled* step (trust_region_step) for pose-side and landmark unknowns.
  mutable double t_schur_ms = 0, t_solve_ms = 0;  // stage clocks of solve_step
  static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  bool solve_step(const NormalEquations& ne, std::vector<double>* step_p, std::vector<double>* step_l, ReducedSystem* rs_out = nullptr) const {
    const double ts0 = now_ms();
    struct StageClock {
      const LM* lm;
      double t0, t_mid = -1;
      ~StageClock() {
        const double t1 = now_ms();
        lm->t_schur_ms += (t_mid < 0 ? t1 : t_mid) - t0, lm->t_solve_ms += t_mid < 0 ? 0.0 : t1 - t_mid;
      }
    } clock{this, ts0};
    ReducedSystem rs;
    reduce(ne, &rs);
    const int np = ne.np;
    const std::vector<uint8_t> active = active_pose_mask(ne);
    // Inactive coordinates: unit diagonal, zero rhs (they are not part of Ceres' reduced program).
    std::vector<double> S = rs.S, y = rs.g;
    for (int i = 0; i < np; ++i)
      if (!active[i]) {
        for (int j = 0; j < np; ++j) S[size_t(i) * np + j] = S[size_t(j) * np + i] = 0.0;
        S[size_t(i) * np + i] = 1.0;
        y[i] = 0.0;
      }
    if (rs_out) {
      *rs_out = rs;
      rs_out->S = S, rs_out->g = y;
    }
    clock.t_mid = now_ms();
    const std::vector<int> env = row_envelope(S, np);
    if (!cholesky_lower(S, np, env)) return false;
    cholesky_solve(S, np, env, y);  // y = (J'J + D^2)^-1 J'r
    step_p->assign(np, 0.0);
    for (int i = 0; i < np; ++i) (*step_p)[i] = -y[i];
    step_l->assign(size_t(ne.nl) * 3, 0.0);
    for (int l = 0; l < ne.nl; ++l) {
      if (ne.w_count[l] == 0 || P.lm_const[l]) continue;
      // V y_l = Dl bl - Ws' y_p  ->  step_l = -y_l
      const int rows = 6 * ne.w_count[l], r0 = 6 * ne.w_first[l];
      double rhs[3];
      for (int a = 0; a < 3; ++a) rhs[a] = scale_l[3 * l + a] * ne.bl[3 * l + a];
      for (int r = 0; r < rows; ++r)
        for (int b = 0; b < 3; ++b) rhs[b] -= scale_p[r0 + r] * ne.W[l][size_t(r) * 3 + b] * scale_l[3 * l + b] * y[r0 + r];
      const double* Vi = &rs.Vinv[9 * l];
      for (int a = 0; a < 3; ++a) (*step_l)[3 * l + a] = -(Vi[3 * a] * rhs[0] + Vi[3 * a + 1] * rhs[1] + Vi[3 * a + 2] * rhs[2]);
    }
    for (double v : *step_p)
      if (!std::isfinite(v)) return false;
    for (double v : *step_l)
      if (!std::isfinite(v)) return false;
    return true;
  }

  /// model_cost_change = -(J step) . (r + J step / 2) on the scaled Jacobian (TrustRegionMinimizer::ComputeTrustRegionStep).
  double model_cost_change(const NormalEquations& ne, const std::vector<double>& delta_p, const std::vector<double>& delta_l) const {
    std::vector<int> cols;
    std::vector<double> vals;
    double acc = 0;
    for (size_t bi = 0; bi < ne.blocks.size(); ++bi) {
      const Linearized& lin = ne.blocks[bi];
      solver.pose_columns(lin, ne.block_type[bi], &cols, &vals);
      const int nc = int(cols.size());
      for (int r = 0; r < lin.n_res; ++r) {
        double m = 0;
        for (int a = 0; a < nc; ++a) m += vals[size_t(r) * nc + a] * delta_p[cols[a]];
        if (lin.lm >= 0)
          for (int b = 0; b < 3; ++b) m += lin.J_lm[r * 3 + b] * delta_l[3 * lin.lm + b];
        acc += m * (lin.r[r] + 0.5 * m);
      }
    }
    return -acc;
  }

  void apply(const std::vector<double>& delta_p, const std::vector<double>& delta_l, Problem* Q) const {
    *Q = P;
    for (int j = 0; j < P.n_cp; ++j) {
      if (P.cp_const[j]) continue;
      double d[6];
      for (int c = 0; c < 6; ++c) d[c] = delta_p[6 * j + c];
      if (P.rot_const) d[0] = d[1] = d[2] = 0;
      if (P.trans_const) d[3] = d[4] = d[5] = 0;
      manifold_plus(kManifoldControlPoint, 8, &P.cp[8 * j], d, &Q->cp[8 * j]);
    }
    for (int l = 0; l < P.n_lm; ++l)
      if (!P.lm_const[l])
        for (int a = 0; a < 3; ++a) Q->lm[3 * l + a] = P.lm[3 * l + a] + delta_l[3 * l + a];
    if (P.has_imu) {
      if (!P.bias_const)
        for (int j = 0; j < P.n_bias; ++j)
          for (int a = 0; a < 3; ++a) {
            Q->bias_g[4 * j + a] = P.bias_g[4 * j + a] + delta_p[P.off_bias_g() + 3 * j + a];
            Q->bias_a[4 * j + a] = P.bias_a[4 * j + a] + delta_p[P.off_bias_a() + 3 * j + a];
          }
      if (!P.gravity_const) manifold_plus(kManifoldSphere3, 3, P.gravity, &delta_p[P.off_gravity()], Q->gravity);
    }
  }

  /// Squared norm of the ambient parameter vector over the non-constant blocks (x_norm in Ceres).
  void x_squared_norm(const NormalEquations& ne, double* replicated, double* landmarks) const {
    double s = 0, sl = 0;
    const std::vector<uint8_t> active = active_pose_mask(ne);
    for (int j = 0; j < P.n_cp; ++j) {
      bool a = false;
      for (int c = 0; c < 6; ++c) a |= active[6 * j + c];
      if (a)
        for (int c = 0; c < 8; ++c) s += P.cp[8 * j + c] * P.cp[8 * j + c];
    }
    for (int l = 0; l < P.n_lm; ++l)
      if (!P.lm_const[l] && ne.w_count[l] > 0)
        for (int a = 0; a < 3; ++a) sl += P.lm[3 * l + a] * P.lm[3 * l + a];
    if (P.has_imu) {
      if (!P.bias_const)
        for (int j = 0; j < P.n_bias; ++j) {
          if (active[P.off_bias_g() + 3 * j])
            for (int a = 0; a < 4; ++a) s += P.bias_g[4 * j + a] * P.bias_g[4 * j + a];
          if (active[P.off_bias_a() + 3 * j])
            for (int a = 0; a < 4; ++a) s += P.bias_a[4 * j + a] * P.bias_a[4 * j + a];
        }
      if (!P.gravity_const && active[P.off_gravity()])
        for (int a = 0; a < 3; ++a) s += P.gravity[a] * P.gravity[a];
    }
    *replicated = s, *landmarks = sl;
  }

  double gradient_max_norm(const NormalEquations& ne, const Problem& P) const {
    double m = 0, ml = 0;
    for (double v : ne.gp) m = std::max(m, std::fabs(v));  // global after globalize()
    for (int l = 0; l < ne.nl; ++l)
      if (!P.lm_const[l])
        for (int a = 0; a < 3; ++a) ml = std::max(ml, std::fabs(ne.bl[3 * l + a]));
    std::vector<double> slots(world, 0.0);  // a SUM all-reduce of one slot per rank delivers every rank's maximum
    slots[rank] = ml;
    sum(slots);
    for (double v : slots) m = std::max(m, v);
    return m;
  }

  Summary run(int max_iterations) {
    constexpr double kFunctionTolerance = 1e-6, kGradientTolerance = 1e-10, kParameterTolerance = 1e-8;
    constexpr double kMinRelativeDecrease = 1e-3, kMaxRadius = 1e16, kMinRadius = 1e-32;
    Summary sum;
    NormalEquations ne;
    double tl = now_ms();
    solver.build(&ne);
    sum.linearize_ms += now_ms() - tl;
    t_schur_ms = t_solve_ms = 0;
    globalize(&ne);
    compute_scaling(ne);
    double cost = ne.cost;
    sum.initial_cost = cost;
    double gmax = gradient_max_norm(ne, P);
    sum.iterations.push_back({0, cost, 0, gmax, 0, 0, radius, 1, 1});
    int invalid_streak = 0;
    sum.termination = 0;
    for (int it = 1;; ++it) {
      // FinalizeIterationAndCheckIfMinimizerCanContinue
      if (it - 1 >= max_iterations) { sum.termination = 0; break; }
      if (gmax <= kGradientTolerance) { sum.termination = 1; break; }
      if (radius <= kMinRadius) { sum.termination = 1; break; }
      IterationRecord rec = {it, cost, 0, gmax, 0, 0, radius, 0, 0};
      sum.num_iterations = it;
      std::vector<double> step_p, step_l;
      bool valid = solve_step(ne, &step_p, &step_l);
      std::vector<double> delta_p, delta_l;
      double mcc = 0;
      if (valid) {
        delta_p = step_p, delta_l = step_l;
        for (size_t i = 0; i < delta_p.size(); ++i) delta_p[i] *= scale_p[i];
        for (size_t i = 0; i < delta_l.size(); ++i) delta_l[i] *= scale_l[i];
        std::vector<double> m1(1, model_cost_change(ne, delta_p, delta_l));  // per-residual sum: additive across shards
        this->sum(m1);
        mcc = m1[0];
        if (!(mcc > 0.0)) valid = false;  // TrustRegionMinimizer: step_is_valid = model_cost_change > 0
      }
      if (!valid) {  // HandleInvalidStep
        if (++invalid_streak >= 5) { sum.termination = 2; sum.iterations.push_back(rec); break; }
        radius *= 0.5;
        rec.radius = radius;
        sum.iterations.push_back(rec);
        continue;
      }
      invalid_streak = 0;
      rec.step_is_valid = 1;
      const double tu = now_ms();
      Problem cand;
      apply(delta_p, delta_l, &cand);
      // candidate cost and norms: replicated unknowns (control points, biases, gravity) are counted by rank 0 only
      double sn = 0, sn_rep = 0;
      for (size_t i = 0; i < P.cp.size(); ++i) sn_rep += (P.cp[i] - cand.cp[i]) * (P.cp[i] - cand.cp[i]);
      for (size_t i = 0; i < P.lm.size(); ++i) sn += (P.lm[i] - cand.lm[i]) * (P.lm[i] - cand.lm[i]);
      for (size_t i = 0; i < P.bias_g.size(); ++i) sn_rep += (P.bias_g[i] - cand.bias_g[i]) * (P.bias_g[i] - cand.bias_g[i]);
      for (size_t i = 0; i < P.bias_a.size(); ++i) sn_rep += (P.bias_a[i] - cand.bias_a[i]) * (P.bias_a[i] - cand.bias_a[i]);
      for (int i = 0; i < 3; ++i) sn_rep += (P.gravity[i] - cand.gravity[i]) * (P.gravity[i] - cand.gravity[i]);
      double xs_rep = 0, xs_lm = 0;
      x_squared_norm(ne, &xs_rep, &xs_lm);
      std::vector<double> dec = {Solver(cand).total_cost(), sn + (rank == 0 ? sn_rep : 0.0), xs_lm + (rank == 0 ? xs_rep : 0.0)};
      this->sum(dec);
      const double cand_cost = dec[0];
      sum.update_ms += now_ms() - tu;
      rec.step_norm = std::sqrt(dec[1]);
      const double x_norm = std::sqrt(dec[2]);
      if (rec.step_norm <= kParameterTolerance * (x_norm + kParameterTolerance)) {
        sum.termination = 1;
        sum.iterations.push_back(rec);
        break;
      }
      // FunctionToleranceReached
      rec.cost_change = cost - cand_cost;
      if (std::fabs(rec.cost_change) <= kFunctionTolerance * cost) {
        sum.termination = 1;
        sum.iterations.push_back(rec);
        break;
      }
      rec.relative_decrease = (cost - cand_cost) / mcc;
      if (rec.relative_decrease > kMinRelativeDecrease) {  // HandleSuccessfulStep
        rec.step_is_successful = 1;
        sum.num_successful_steps++;
        P = cand;
        tl = now_ms();
        solver.build(&ne);
        sum.linearize_ms += now_ms() - tl;
        globalize(&ne);
        cost = ne.cost;
        gmax = gradient_max_norm(ne, P);
        rec.cost = cost, rec.gradient_max_norm = gmax;
        radius = radius / std::max<double>(1.0 / 3.0, 1.0 - std::pow(2.0 * rec.relative_decrease - 1.0, 3));
        radius = std::min(kMaxRadius, radius);
        decrease_factor = 2.0;
      } else {
        rec.cost = cand_cost;  // TrustRegionMinimizer reports the candidate's cost for an unsuccessful step
        radius = radius / decrease_factor;
        decrease_factor *= 2.0;
      }
      rec.radius = radius;
      sum.iterations.push_back(rec);
    }
    sum.final_cost = cost;
    sum.schur_ms = t_schur_ms, sum.solve_ms = t_solve_ms;
    return sum;
  }
};

}  // namespace hso
