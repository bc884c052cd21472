// ORACLE — TEST INFRASTRUCTURE ONLY. Nothing under oracle/ is part of the shipped product; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it (as the checker / CPU denominator).
//
// PARITY UNPINNED: the reference's arithmetic for this path lives in un-vendored dependencies
// (HyperSensors@a24e9e1a -> HyperState -> HyperVariables, Ceres >= 2.1, Eigen; /root/reference/CMakeLists.txt:26-27)
// that are absent from the build image, and the reference's own tests hold no golden vectors
// (only analytic-vs-numeric Jacobian self-consistency, tests/include/tests/optimizers/evaluators/evaluator.hpp:38-65).
// This file restates the published algorithms (Sommer et al. 2020 cumulative B-splines, README.md:174; Sola 2018)
// under the call-site constraints of the in-tree evaluators; it is pinned against an independent 50-digit
// mpmath restatement (tests/golden/make_golden.py) and by the reference-style gradient probe.
//
// hs_math.hpp: dependency-free fixed-size linear algebra, quaternion / SO(3) primitives
// and the uniform cumulative B-spline on split SE3 = SU2 x R^3.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace hso {

// ---------------------------------------------------------------------------------------------------------
// Fixed-size dense matrix (row-major), value semantics.
// ---------------------------------------------------------------------------------------------------------
template <int R, int C>
struct Mat {
  double a[R * C];
  double& operator()(int i, int j) { return a[i * C + j]; }
  const double& operator()(int i, int j) const { return a[i * C + j]; }
  double& operator[](int i) { return a[i]; }
  const double& operator[](int i) const { return a[i]; }
  static Mat zero() {
    Mat m;
    for (int i = 0; i < R * C; ++i) m.a[i] = 0.0;
    return m;
  }
  static Mat identity() {
    Mat m = zero();
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0;
    return m;
  }
};
using V3 = Mat<3, 1>;
using M3 = Mat<3, 3>;

template <int R, int K, int C>
inline Mat<R, C> operator*(const Mat<R, K>& x, const Mat<K, C>& y) {
  Mat<R, C> z;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += x(i, k) * y(k, j);
      z(i, j) = s;
    }
  return z;
}
template <int R, int C>
inline Mat<R, C> operator+(const Mat<R, C>& x, const Mat<R, C>& y) {
  Mat<R, C> z;
  for (int i = 0; i < R * C; ++i) z.a[i] = x.a[i] + y.a[i];
  return z;
}
template <int R, int C>
inline Mat<R, C> operator-(const Mat<R, C>& x, const Mat<R, C>& y) {
  Mat<R, C> z;
  for (int i = 0; i < R * C; ++i) z.a[i] = x.a[i] - y.a[i];
  return z;
}
template <int R, int C>
inline Mat<R, C> operator*(double s, const Mat<R, C>& x) {
  Mat<R, C> z;
  for (int i = 0; i < R * C; ++i) z.a[i] = s * x.a[i];
  return z;
}
template <int R, int C>
inline Mat<R, C> operator-(const Mat<R, C>& x) {
  return -1.0 * x;
}
template <int R, int C>
inline Mat<C, R> T(const Mat<R, C>& x) {
  Mat<C, R> z;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) z(j, i) = x(i, j);
  return z;
}
inline V3 v3(double x, double y, double z) {
  V3 v;
  v[0] = x, v[1] = y, v[2] = z;
  return v;
}
inline double dot(const V3& a, const V3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }
inline V3 cross(const V3& a, const V3& b) {
  return v3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
/// hat(a) b = a x b (reference: `.hat()`, inertial.cpp:124).
inline M3 hat(const V3& a) {
  M3 m = M3::zero();
  m(0, 1) = -a[2], m(0, 2) = a[1];
  m(1, 0) = a[2], m(1, 2) = -a[0];
  m(2, 0) = -a[1], m(2, 1) = a[0];
  return m;
}

// ---------------------------------------------------------------------------------------------------------
// Quaternion, Eigen coefficient order (x, y, z, w) (reference su2.cpp:21 EigenQuaternionManifold,
// settings.yaml:34-36). Hamilton product.
// ---------------------------------------------------------------------------------------------------------
struct Quat {
  double x, y, z, w;
};
inline Quat qmul(const Quat& a, const Quat& b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Quat qconj(const Quat& a) { return {-a.x, -a.y, -a.z, a.w}; }
inline Quat qnormalized(const Quat& a) {
  const double n = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  return {a.x / n, a.y / n, a.z / n, a.w / n};
}
inline M3 qmat(const Quat& q) {
  M3 R;
  const double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
  const double xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
  const double wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  R(0, 0) = 1 - 2 * (yy + zz), R(0, 1) = 2 * (xy - wz), R(0, 2) = 2 * (xz + wy);
  R(1, 0) = 2 * (xy + wz), R(1, 1) = 1 - 2 * (xx + zz), R(1, 2) = 2 * (yz - wx);
  R(2, 0) = 2 * (xz - wy), R(2, 1) = 2 * (yz + wx), R(2, 2) = 1 - 2 * (xx + yy);
  return R;
}

// ---------------------------------------------------------------------------------------------------------
// SO(3) exponential / logarithm and their Jacobians (Sola 2018; README.md:170-178 literature).
// ---------------------------------------------------------------------------------------------------------
inline Quat so3_exp_q(const V3& phi) {
  const double t2 = dot(phi, phi);
  double s, c;
  if (t2 < 1e-16) {
    s = 0.5 - t2 / 48.0;
    c = 1.0 - t2 / 8.0;
  } else {
    const double t = std::sqrt(t2);
    s = std::sin(0.5 * t) / t;
    c = std::cos(0.5 * t);
  }
  return {s * phi[0], s * phi[1], s * phi[2], c};
}
inline M3 so3_exp(const V3& phi) {
  const double t2 = dot(phi, phi);
  double A, B;
  if (t2 < 1e-12) {
    A = 1.0 - t2 / 6.0;
    B = 0.5 - t2 / 24.0;
  } else {
    const double t = std::sqrt(t2);
    A = std::sin(t) / t;
    B = (1.0 - std::cos(t)) / t2;
  }
  const M3 H = hat(phi);
  return M3::identity() + A * H + B * (H * H);
}
/// Principal logarithm (angle in [0, pi]); q and -q map to the same tangent (shortest path).
inline V3 so3_log_q(Quat q) {
  if (q.w < 0) q = {-q.x, -q.y, -q.z, -q.w};
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z;
  double s;
  if (n2 < 1e-16) {
    s = 2.0 / q.w * (1.0 - n2 / (3.0 * q.w * q.w));
  } else {
    const double n = std::sqrt(n2);
    s = 2.0 * std::atan2(n, q.w) / n;
  }
  return v3(s * q.x, s * q.y, s * q.z);
}
/// Right Jacobian J_r(phi): Exp(phi + d) ~ Exp(phi) Exp(J_r d).
inline M3 so3_Jr(const V3& phi) {
  const double t2 = dot(phi, phi);
  double B, C;
  if (t2 < 1e-8) {
    B = 0.5 - t2 / 24.0;
    C = 1.0 / 6.0 - t2 / 120.0;
  } else {
    const double t = std::sqrt(t2);
    B = (1.0 - std::cos(t)) / t2;
    C = (t - std::sin(t)) / (t2 * t);
  }
  const M3 H = hat(phi);
  return M3::identity() - B * H + C * (H * H);
}
/// Inverse right Jacobian; J_l^-1(phi) = J_r^-1(-phi) = J_r^-1(phi)^T.
inline M3 so3_Jr_inv(const V3& phi) {
  const double t2 = dot(phi, phi);
  double D;
  if (t2 < 1e-8) {
    D = 1.0 / 12.0 + t2 / 720.0;
  } else {
    const double t = std::sqrt(t2);
    D = 1.0 / t2 - (1.0 + std::cos(t)) / (2.0 * t * std::sin(t));
  }
  const M3 H = hat(phi);
  return M3::identity() + 0.5 * H + D * (H * H);
}

// ---------------------------------------------------------------------------------------------------------
// Uniform cumulative B-spline basis of order k (EXTERNAL BasisInterpolator(degree, uniform),
// call sites abstract.cpp:79, tests/.../pixel.cpp:50).
// ---------------------------------------------------------------------------------------------------------
constexpr int kMaxOrder = 8;

struct Basis {
  int k;
  double Ct[kMaxOrder][kMaxOrder];  // cumulative blending matrix, Ct[j][n]
};

inline double binom(int n, int r) {
  if (r < 0 || r > n) return 0.0;
  double v = 1.0;
  for (int i = 1; i <= r; ++i) v = v * (n - r + i) / i;
  return v;
}

inline Basis make_basis(int k) {
  Basis b;
  b.k = k;
  double M[kMaxOrder][kMaxOrder];
  double fact = 1.0;
  for (int i = 2; i <= k - 1; ++i) fact *= i;
  for (int s = 0; s < k; ++s)
    for (int n = 0; n < k; ++n) {
      double sum = 0.0;
      for (int l = s; l <= k - 1; ++l) {
        const double sign = ((l - s) % 2 == 0) ? 1.0 : -1.0;
        sum += sign * binom(k, l - s) * std::pow(double(k - 1 - l), double(k - 1 - n));
      }
      M[s][n] = binom(k - 1, n) / fact * sum;
    }
  for (int j = 0; j < k; ++j)
    for (int n = 0; n < k; ++n) {
      double sum = 0.0;
      for (int s = j; s < k; ++s) sum += M[s][n];
      b.Ct[j][n] = sum;
    }
  return b;
}

/// Cumulative weights and their first two time derivatives at normalised time u in [0,1); inv_dt = 1/separation.
inline void basis_weights(const Basis& b, double u, double inv_dt, double* lam, double* dlam, double* ddlam) {
  const int k = b.k;
  double pw[kMaxOrder];
  pw[0] = 1.0;
  for (int n = 1; n < k; ++n) pw[n] = pw[n - 1] * u;
  for (int j = 0; j < k; ++j) {
    double l0 = 0, l1 = 0, l2 = 0;
    for (int n = 0; n < k; ++n) {
      l0 += b.Ct[j][n] * pw[n];
      if (n >= 1) l1 += b.Ct[j][n] * n * pw[n - 1];
      if (n >= 2) l2 += b.Ct[j][n] * n * (n - 1) * pw[n - 2];
    }
    lam[j] = l0;
    dlam[j] = l1 * inv_dt;
    ddlam[j] = l2 * inv_dt * inv_dt;
  }
}

/// Segment lookup for uniform knots: control point j carries stamp t0 + j*dt. Returns index of the first of the
/// k control points used at stamp t and writes u. (abstract.cpp:89 bootstrap stamps; optimizer.cpp:288-290 padding.)
inline int segment_of(double t, double t0, double dt, int k, double* u) {
  const double x = (t - t0) / dt;
  const double fl = std::floor(x);
  *u = x - fl;
  return int(fl) - (k - 1) / 2;
}

// ---------------------------------------------------------------------------------------------------------
// Split SE3 cumulative spline: value / velocity / acceleration with Jacobians w.r.t. the k control points in
// *tangent* coordinates (phi_j: world-frame/left rotation perturbation R_j <- Exp(phi_j) R_j; dp_j additive).
// SURVEY.md Appendix A.2 / A.2b.  EXTERNAL AbstractState::evaluate (call sites bearing.cpp:59-60, inertial.cpp:93-94).
// ---------------------------------------------------------------------------------------------------------
struct SplineValue {
  Quat q;    // rotation R_wb
  M3 R;      // same as matrix
  V3 p;      // position (world)
  V3 w;      // body-frame angular velocity
  V3 al;     // body-frame angular acceleration d(w)/dt
  V3 v;      // world-frame linear velocity
  V3 a;      // world-frame linear acceleration
  // Jacobians (filled iff requested). Rotation blocks are w.r.t. phi_j (left), 3x3 each.
  M3 dth[kMaxOrder];  // d theta / d phi_j, theta = left perturbation of R
  M3 dw[kMaxOrder];   // d w / d phi_j
  M3 dal[kMaxOrder];  // d al / d phi_j
  double B[kMaxOrder], Bd[kMaxOrder], Bdd[kMaxOrder];  // p = sum B_j p_j etc. (non-cumulative weights)
};

/// cps: k pointers to 8-double blocks [qx qy qz qw px py pz t] (Stamped<SE3>).
inline void spline_evaluate(const Basis& basis, const double* const* cps, double u, double inv_dt, int derivative, bool jac,
                            SplineValue* out) {
  const int k = basis.k;
  double lam[kMaxOrder], dlam[kMaxOrder], ddlam[kMaxOrder];
  basis_weights(basis, u, inv_dt, lam, dlam, ddlam);

  Quat qs[kMaxOrder] = {};
  M3 Rs[kMaxOrder];
  for (int j = 0; j < k; ++j) {
    qs[j] = {cps[j][0], cps[j][1], cps[j][2], cps[j][3]};
    if (jac) Rs[j] = qmat(qs[j]);
  }

  // Translation part.
  V3 p = v3(0, 0, 0), v = p, a = p;
  for (int j = 0; j < k; ++j) {
    const double Bj = lam[j] - (j + 1 < k ? lam[j + 1] : 0.0);
    const double Bdj = dlam[j] - (j + 1 < k ? dlam[j + 1] : 0.0);
    const double Bddj = ddlam[j] - (j + 1 < k ? ddlam[j + 1] : 0.0);
    out->B[j] = Bj, out->Bd[j] = Bdj, out->Bdd[j] = Bddj;
    const V3 pj = v3(cps[j][4], cps[j][5], cps[j][6]);
    p = p + Bj * pj;
    v = v + Bdj * pj;
    a = a + Bddj * pj;
  }
  out->p = p, out->v = v, out->a = a;

  // Rotation part (right-perturbation recursion, A.2b).
  Quat q = qs[0];
  V3 w = v3(0, 0, 0), al = w;
  M3 E[kMaxOrder], W[kMaxOrder], Qm[kMaxOrder];
  if (jac) {
    for (int m = 0; m < k; ++m) E[m] = W[m] = Qm[m] = M3::zero();
    E[0] = M3::identity();
  }
  for (int j = 1; j < k; ++j) {
    const V3 d = so3_log_q(qmul(qconj(qs[j - 1]), qs[j]));
    const V3 ld = lam[j] * d;
    const Quat Aq = so3_exp_q(ld);
    q = qmul(q, Aq);
    if (derivative < 1 && !jac) continue;
    const M3 At = T(so3_exp(ld));
    const V3 w_rot = At * w;
    const V3 al_rot = At * al;
    const V3 w_new = w_rot + dlam[j] * d;
    const V3 al_new = al_rot + dlam[j] * cross(w_new, d) + ddlam[j] * d;
    if (jac) {
      const M3 Jri = so3_Jr_inv(d);
      const M3 Jli = T(Jri);
      const M3 JrL = so3_Jr(ld);
      for (int m = 0; m < k; ++m) {
        E[m] = At * E[m];
        W[m] = At * W[m];
        Qm[m] = At * Qm[m];
      }
      const M3 hw_rot = hat(w_rot), hal_rot = hat(al_rot), hd = hat(d), hw_new = hat(w_new);
      // delta d_j = Jri eps_j - Jli eps_{j-1}
      const int blk[2] = {j - 1, j};
      const M3 dd[2] = {-Jli, Jri};
      M3 Wadd[2];
      for (int c = 0; c < 2; ++c) {
        const M3 eta = lam[j] * (JrL * dd[c]);
        E[blk[c]] = E[blk[c]] + eta;
        Wadd[c] = hw_rot * eta + dlam[j] * dd[c];
        W[blk[c]] = W[blk[c]] + Wadd[c];
        Qm[blk[c]] = Qm[blk[c]] + hal_rot * eta + dlam[j] * (hw_new * dd[c]) + ddlam[j] * dd[c];
      }
      for (int m = 0; m < k; ++m) Qm[m] = Qm[m] - dlam[j] * (hd * W[m]);
    }
    w = w_new, al = al_new;
  }
  out->q = qnormalized(q);
  out->R = qmat(out->q);
  out->w = w, out->al = al;
  if (jac) {
    for (int m = 0; m < k; ++m) {
      const M3 Rmt = T(Rs[m]);
      out->dth[m] = out->R * E[m] * Rmt;
      out->dw[m] = W[m] * Rmt;
      out->dal[m] = Qm[m] * Rmt;
    }
  }
}

/// R^3 uniform B-spline (IMU bias splines, Stamped<Cartesian3> = [x y z t]; imu.cpp:64-66, inertial.cpp:58-60).
inline V3 r3_spline_evaluate(const Basis& basis, const double* const* cps, double u, double* Bout) {
  const int k = basis.k;
  double lam[kMaxOrder], dlam[kMaxOrder], ddlam[kMaxOrder];
  basis_weights(basis, u, 1.0, lam, dlam, ddlam);
  V3 b = v3(0, 0, 0);
  for (int j = 0; j < k; ++j) {
    const double Bj = lam[j] - (j + 1 < k ? lam[j + 1] : 0.0);
    Bout[j] = Bj;
    b = b + Bj * v3(cps[j][0], cps[j][1], cps[j][2]);
  }
  return b;
}

// ---------------------------------------------------------------------------------------------------------
// SplitMix64 (shared bit-for-bit with the Python generator).
// ---------------------------------------------------------------------------------------------------------
struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
};

}  // namespace hso
