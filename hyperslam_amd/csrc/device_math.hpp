// device_math.hpp — gfx950 device-side primitives of the continuous-time NLLS hot path.
//
// Register-resident fixed-size fp64 algebra for one residual per lane: quaternion / SO(3) maps, the uniform
// cumulative B-spline on split SE3 = SU2 x R^3 (EXTERNAL AbstractState::evaluate; call sites
// /root/reference/internal/hyper/optimizers/evaluators/bearing.cpp:59-60, inertial.cpp:93-94)
// and Ceres' manifold retractions. Unlike the reference (ambient 8k-column Jacobians later projected
// by Ceres), everything here is differentiated directly in Ceres-local coordinates (rotation: delta with
// R <- Exp(2 delta) R; translation additive), so no adapter / PlusJacobian products exist on the device.
#pragma once
#include <hip/hip_runtime.h>

namespace hsd {

#define HSD __device__ __forceinline__

constexpr int kMaxOrder = 8;

struct V3 {
  double x, y, z;
};
struct M3 {
  double m[9];  // row-major
};
struct Quat {
  double x, y, z, w;
};

HSD V3 mk(double x, double y, double z) { return V3{x, y, z}; }
HSD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
HSD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
HSD V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
HSD double dot(V3 a, V3 b) { return fma(a.x, b.x, fma(a.y, b.y, a.z * b.z)); }
HSD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

HSD M3 eye() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }
HSD M3 zero3() { return M3{{0, 0, 0, 0, 0, 0, 0, 0, 0}}; }
HSD M3 mul(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[3 * i + j] = fma(a.m[3 * i], b.m[j], fma(a.m[3 * i + 1], b.m[3 + j], a.m[3 * i + 2] * b.m[6 + j]));
  return c;
}
/// a * b^T
HSD M3 mul_nt(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[3 * i + j] = fma(a.m[3 * i], b.m[3 * j], fma(a.m[3 * i + 1], b.m[3 * j + 1], a.m[3 * i + 2] * b.m[3 * j + 2]));
  return c;
}
/// a^T * b
HSD M3 mul_tn(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c.m[3 * i + j] = fma(a.m[i], b.m[j], fma(a.m[3 + i], b.m[3 + j], a.m[6 + i] * b.m[6 + j]));
  return c;
}
HSD V3 mul(const M3& a, V3 v) {
  return V3{fma(a.m[0], v.x, fma(a.m[1], v.y, a.m[2] * v.z)), fma(a.m[3], v.x, fma(a.m[4], v.y, a.m[5] * v.z)),
            fma(a.m[6], v.x, fma(a.m[7], v.y, a.m[8] * v.z))};
}
/// a^T v
HSD V3 mul_t(const M3& a, V3 v) {
  return V3{fma(a.m[0], v.x, fma(a.m[3], v.y, a.m[6] * v.z)), fma(a.m[1], v.x, fma(a.m[4], v.y, a.m[7] * v.z)),
            fma(a.m[2], v.x, fma(a.m[5], v.y, a.m[8] * v.z))};
}
HSD M3 add(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] + b.m[i];
  return c;
}
HSD M3 sub(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = a.m[i] - b.m[i];
  return c;
}
HSD M3 scale(double s, const M3& a) {
  M3 c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.m[i] = s * a.m[i];
  return c;
}
HSD M3 transpose(const M3& a) { return M3{{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}}; }
HSD M3 hat(V3 a) { return M3{{0, -a.z, a.y, a.z, 0, -a.x, -a.y, a.x, 0}}; }
/// I + a*hat(v) + b*hat(v)^2
HSD M3 rodrigues_poly(V3 v, double a, double b) {
  const double xx = v.x * v.x, yy = v.y * v.y, zz = v.z * v.z;
  const double xy = v.x * v.y, xz = v.x * v.z, yz = v.y * v.z;
  M3 r;
  r.m[0] = 1.0 - b * (yy + zz), r.m[1] = fma(b, xy, -a * v.z), r.m[2] = fma(b, xz, a * v.y);
  r.m[3] = fma(b, xy, a * v.z), r.m[4] = 1.0 - b * (xx + zz), r.m[5] = fma(b, yz, -a * v.x);
  r.m[6] = fma(b, xz, -a * v.y), r.m[7] = fma(b, yz, a * v.x), r.m[8] = 1.0 - b * (xx + yy);
  return r;
}

// ---- quaternion (x, y, z, w), Hamilton --------------------------------------------------------------------------
HSD Quat qmul(Quat a, Quat b) {
  return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
HSD Quat qconj(Quat a) { return Quat{-a.x, -a.y, -a.z, a.w}; }
HSD M3 qmat(Quat q) {
  const double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z;
  const double xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
  const double wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
  return M3{{1 - 2 * (yy + zz), 2 * (xy - wz), 2 * (xz + wy), 2 * (xy + wz), 1 - 2 * (xx + zz), 2 * (yz - wx), 2 * (xz - wy), 2 * (yz + wx),
             1 - 2 * (xx + yy)}};
}
HSD Quat qnormalized(Quat a) {
  const double n = rsqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  return Quat{a.x * n, a.y * n, a.z * n, a.w * n};
}

// ---- SO(3) ----------------------------------------------------------------------------------------------------------
/// Principal logarithm of a unit quaternion (angle in [0, pi]).
HSD V3 so3_log(Quat q) {
  if (q.w < 0) q = Quat{-q.x, -q.y, -q.z, -q.w};
  const double n2 = q.x * q.x + q.y * q.y + q.z * q.z;
  double s;
  if (n2 < 1e-16) {
    s = 2.0 / q.w * (1.0 - n2 / (3.0 * q.w * q.w));
  } else {
    const double n = sqrt(n2);
    s = 2.0 * atan2(n, q.w) / n;
  }
  return V3{s * q.x, s * q.y, s * q.z};
}

/// Coefficients of the SO(3) maps of a rotation vector with squared norm t2:
///   Exp = I + A H + B H^2;  J_r = I - B H + C H^2;  J_r^-1 = I + H/2 + D H^2   (H = hat(phi)).
struct So3Coef {
  double A, B, C, D;
};
HSD So3Coef so3_coef(double t2, bool need_inv) {
  So3Coef c;
  if (t2 < 1e-8) {
    c.A = 1.0 - t2 / 6.0 * (1.0 - t2 / 20.0);
    c.B = 0.5 - t2 / 24.0 * (1.0 - t2 / 30.0);
    c.C = 1.0 / 6.0 - t2 / 120.0 * (1.0 - t2 / 42.0);
    c.D = 1.0 / 12.0 + t2 / 720.0;
  } else {
    const double t = sqrt(t2);
    double s, co;
    sincos(t, &s, &co);
    const double it2 = 1.0 / t2;
    c.A = s / t;
    c.B = (1.0 - co) * it2;
    c.C = (t - s) * it2 / t;
    c.D = need_inv ? (it2 - (1.0 + co) / (2.0 * t * s)) : 0.0;
  }
  return c;
}

// ---- B-spline basis -------------------------------------------------------------------------------------------------
/// Cumulative blending matrix of order K (row j, power n), computed on the host (host_basis.hpp) and passed by value.
struct BasisCoef {
  double c[kMaxOrder * kMaxOrder];
};

template <int K>
HSD void basis_weights(const BasisCoef& bc, double u, double inv_dt, double* lam, double* dlam, double* ddlam, int derivative) {
  double pw[K];
  pw[0] = 1.0;
#pragma unroll
  for (int n = 1; n < K; ++n) pw[n] = pw[n - 1] * u;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    double l0 = 0, l1 = 0, l2 = 0;
#pragma unroll
    for (int n = 0; n < K; ++n) {
      const double c = bc.c[j * kMaxOrder + n];
      l0 = fma(c, pw[n], l0);
      if (n >= 1) l1 = fma(c * n, pw[n - 1], l1);
      if (n >= 2) l2 = fma(c * (n * (n - 1)), pw[n - 2], l2);
    }
    lam[j] = l0;
    if (derivative >= 1) dlam[j] = l1 * inv_dt;
    if (derivative >= 2) ddlam[j] = l2 * inv_dt * inv_dt;
  }
}

/// Segment lookup (uniform knots): first control point index and normalised time.
HSD int segment_of(double t, double t0, double dt, int k, double* u) {
  const double x = (t - t0) / dt;  // IEEE division: identical to the host-side structure builder
  const double fl = floor(x);
  *u = x - fl;
  return int(fl) - (k - 1) / 2;
}

// ---- Ceres retractions ---------------------------------------------------------------------------
/// EigenQuaternionManifold::Plus: x+ = [sin|d|/|d| d ; cos|d|] (x) x   (su2.cpp:21).
HSD Quat quat_plus(Quat x, V3 d) {
  const double n2 = dot(d, d);
  if (n2 == 0.0) return x;
  const double n = sqrt(n2);
  double s, c;
  sincos(n, &s, &c);
  s /= n;
  return qmul(Quat{s * d.x, s * d.y, s * d.z, c}, x);
}
/// EigenQuaternionManifold::Minus: [v ; w] = y (x) conj(x), delta = atan2(|v|, w) / |v| * v (zero for v = 0).
HSD V3 quat_minus(Quat y, Quat x) {
  const Quat r = qmul(y, qconj(x));
  const double n = sqrt(r.x * r.x + r.y * r.y + r.z * r.z);
  if (n == 0.0) return V3{0.0, 0.0, 0.0};
  const double s = atan2(n, r.w) / n;
  return V3{s * r.x, s * r.y, s * r.z};
}
/// Householder vector of Ceres' SphereManifold<3> (manifolds/variables/bearing.cpp:15, gravity.hpp:11-17).
HSD void sphere_householder(const double* x, double* v, double* beta) {
  const double sigma = x[0] * x[0] + x[1] * x[1];
  v[0] = x[0], v[1] = x[1], v[2] = 1.0;
  *beta = 0.0;
  const double x_pn = x[2];
  if (sigma <= 2.220446049250313e-16) {
    if (x_pn < 0) *beta = 2.0;
    return;
  }
  const double mu = sqrt(x_pn * x_pn + sigma);
  const double v_pivot = (x_pn <= 0.0) ? (x_pn - mu) : (-sigma / (x_pn + mu));
  *beta = 2.0 * v_pivot * v_pivot / (sigma + v_pivot * v_pivot);
  v[0] /= v_pivot, v[1] /= v_pivot;
}
HSD void sphere_plus(const double* x, const double* d, double* out) {
  const double nd = sqrt(d[0] * d[0] + d[1] * d[1]);
  if (nd == 0.0) {
    out[0] = x[0], out[1] = x[1], out[2] = x[2];
    return;
  }
  double v[3], beta;
  sphere_householder(x, v, &beta);
  double s, c;
  sincos(nd, &s, &c);
  s /= nd;
  const double y[3] = {s * d[0], s * d[1], c};
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  const double vy = beta * (v[0] * y[0] + v[1] * y[1] + v[2] * y[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) out[i] = nx * (y[i] - v[i] * vy);
}
/// 3x2 PlusJacobian of SphereManifold<3> (row-major).
HSD void sphere_plus_jacobian(const double* x, double* J) {
  double v[3], beta;
  sphere_householder(x, v, &beta);
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 3; ++r) J[r * 2 + i] = nx * ((r == i ? 1.0 : 0.0) - beta * v[r] * v[i]);
}
/// SphereManifold<3>::Minus: h = H y / |x|, delta = atan2(|h_t|, h_last) / |h_t| * h_t over the two tangent entries of h.
HSD void sphere_minus(const double* y, const double* x, double* out) {
  double v[3], beta;
  sphere_householder(x, v, &beta);
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  const double vy = beta * (v[0] * y[0] + v[1] * y[1] + v[2] * y[2]);
  const double h0 = (y[0] - v[0] * vy) / nx, h1 = (y[1] - v[1] * vy) / nx, h2 = (y[2] - v[2] * vy) / nx;
  const double n = sqrt(h0 * h0 + h1 * h1);
  if (n == 0.0) {
    out[0] = out[1] = 0.0;
    return;
  }
  const double s = atan2(n, h2) / n;
  out[0] = s * h0, out[1] = s * h1;
}
/// 2x3 MinusJacobian of SphereManifold<3> (row-major): the first two rows of H / |x|.
HSD void sphere_minus_jacobian(const double* x, double* J) {
  double v[3], beta;
  sphere_householder(x, v, &beta);
  const double nx = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) J[i * 3 + c] = ((c == i ? 1.0 : 0.0) - beta * v[i] * v[c]) / nx;
}

// ---- losses (Ceres semantics; constants optimizer.cpp:204,226,250,267-268) -----------------------
/// Returns rho(s) and writes sqrt(rho'(s)), the corrector scale (rho'' <= 0 for all in-tree losses => alpha = 0).
HSD double loss_huber(double s, double a, double* sqrt_rho1) {
  const double b = a * a;
  if (s > b) {
    const double r = sqrt(s);
    *sqrt_rho1 = sqrt(a / r);
    return 2.0 * a * r - b;
  }
  *sqrt_rho1 = 1.0;
  return s;
}

}  // namespace hsd
