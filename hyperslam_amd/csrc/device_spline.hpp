// device_spline.hpp — split-SE3 cumulative B-spline evaluation on the device.
//
// EXTERNAL AbstractState::evaluate(StateQuery{stamp, derivative, jacobian}, raw) is replaced by three fixed-K
// device routines that work on one residual per lane, reading the k control points of the segment from LDS:
//   spline_pose<K>       value only (cost re-evaluation, bearing.cpp:42-51 branch)
//   spline_pose_jac<K>   value + d theta / d phi_j (visual + prior factors; derivative 0)
//   spline_full<K>       value, body angular velocity / acceleration, world linear velocity / acceleration and all
//                        rotation Jacobians (inertial factor; derivative 2, inertial.cpp:93-94)
//
// Rotation Jacobians w.r.t. a world-frame perturbation phi_j of control point j telescope (derivation in DESIGN.md):
//   with d_j = Log(R_{j-1}^T R_j), Q_j = R_0 Exp(l_1 d_1) ... Exp(l_j d_j), X_j = l_j J_r(l_j d_j) J_r^-1(d_j),
//   T_j = Q_j X_j R_j^T:      d theta / d phi_0 = I - T_1,   d theta / d phi_j = T_j - T_{j+1},   d theta / d phi_{K-1} = T_{K-1}.
// X_j is a polynomial in hat(d_j) and is formed from scalars (no 3x3 products); all trigonometry comes from one atan2
// and one sincos per j (the half-angle of d_j is read off the relative quaternion).
#pragma once
#include "device_math.hpp"

namespace hsd {

struct RelRot {
  V3 d;        // Log(R_{j-1}^T R_j)
  double t2;   // |d|^2
  double sh, ch;  // sin, cos of half the angle (read off the relative quaternion)
};

/// Log of the relative rotation between two control-point quaternions + the trig of its angle for free.
HSD RelRot rel_log(Quat a, Quat b) {
  Quat r = qmul(qconj(a), b);
  if (r.w < 0) r = Quat{-r.x, -r.y, -r.z, -r.w};
  const double n2 = r.x * r.x + r.y * r.y + r.z * r.z;
  RelRot o;
  double s;
  if (n2 < 1e-16) {
    s = 2.0 / r.w * (1.0 - n2 / (3.0 * r.w * r.w));
  } else {
    const double n = sqrt(n2);
    s = 2.0 * atan2(n, r.w) / n;
  }
  o.d = V3{s * r.x, s * r.y, s * r.z};
  o.t2 = s * s * n2;
  const double inv = rsqrt(n2 + r.w * r.w);  // guards slightly non-unit control points
  o.sh = sqrt(n2) * inv;
  o.ch = r.w * inv;
  return o;
}

/// Exp(lam * d) as a quaternion; also returns the J_r(lam d) polynomial coefficients (scaled for H = hat(d)):
///   J_r(lam d) = I - b H + c H^2.
HSD Quat exp_scaled(const RelRot& rr, double lam, double* b, double* c) {
  const double a2 = lam * lam * rr.t2;  // (lam*theta)^2
  double sh, ch;                        // sin, cos of half angle
  double sq;                            // sin(a/2)/theta  (so that q.v = sq * d ... scaled by lam below)
  if (a2 < 1e-10) {
    sq = lam * (0.5 - a2 / 48.0);
    ch = 1.0 - a2 / 8.0 + a2 * a2 / 384.0;
    *b = lam * (0.5 - a2 / 24.0);
    *c = lam * lam * (1.0 / 6.0 - a2 / 120.0);
  } else {
    const double theta = sqrt(rr.t2);
    const double a = lam * theta;
    sincos(0.5 * a, &sh, &ch);
    sq = sh / theta;
    const double one_minus_cos = 2.0 * sh * sh, sin_a = 2.0 * sh * ch;
    const double ia2 = 1.0 / a2;
    *b = lam * one_minus_cos * ia2;
    *c = lam * lam * (a - sin_a) * ia2 / a;
  }
  return Quat{sq * rr.d.x, sq * rr.d.y, sq * rr.d.z, ch};
}

/// Coefficient D of J_r^-1(d) = I + H/2 + D H^2.
HSD double jr_inv_coef(const RelRot& rr) {
  if (rr.t2 < 1e-8) return 1.0 / 12.0 + rr.t2 / 720.0;
  // (1 + cos t) / (2 t sin t) = cot(t/2) / (2 t)
  const double theta = sqrt(rr.t2);
  return 1.0 / rr.t2 - rr.ch / (2.0 * theta * rr.sh);
}

HSD Quat load_quat(const double* cp) { return Quat{cp[0], cp[1], cp[2], cp[3]}; }

/// Value only. cp points at the first of K consecutive 8-double control points.
template <int K>
HSD void spline_pose(const double* cp, const double* lam, Quat* q_out, V3* p_out) {
  Quat qprev = load_quat(cp);
  Quat q = qprev;
  V3 p = V3{cp[4], cp[5], cp[6]};
  V3 pprev = p;
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const double* c = cp + 8 * j;
    const Quat qj = load_quat(c);
    const RelRot rr = rel_log(qprev, qj);
    double b, cc;
    q = qmul(q, exp_scaled(rr, lam[j], &b, &cc));
    const V3 pj = V3{c[4], c[5], c[6]};
    p = p + lam[j] * (pj - pprev);
    qprev = qj, pprev = pj;
  }
  *q_out = qnormalized(q);
  *p_out = p;
}

/// Value + rotation Jacobian blocks G[j] = d theta / d phi_j (world-frame perturbations, theta = left perturbation of R).
/// The translation Jacobian is B_j * I with B_j = lam_j - lam_{j+1}.
template <int K>
HSD void spline_pose_jac(const double* cp, const double* lam, Quat* q_out, V3* p_out, M3* G) {
  Quat qprev = load_quat(cp);
  Quat q = qprev;
  V3 p = V3{cp[4], cp[5], cp[6]};
  V3 pprev = p;
  G[0] = eye();
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const double* c = cp + 8 * j;
    const Quat qj = load_quat(c);
    const RelRot rr = rel_log(qprev, qj);
    double b, cc;
    q = qmul(q, exp_scaled(rr, lam[j], &b, &cc));
    const double D = jr_inv_coef(rr);
    // X = lam (I - b H + c H^2)(I + H/2 + D H^2) = lam (I + alpha H + beta H^2)   (H^3 = -t2 H)
    const double alpha = 0.5 - b + rr.t2 * (b * D - 0.5 * cc);
    const double beta = D - 0.5 * b + cc - rr.t2 * cc * D;
    M3 X = rodrigues_poly(rr.d, alpha, beta);
    X = scale(lam[j], X);
    const M3 Tj = mul_nt(mul(qmat(q), X), qmat(qj));
    G[j - 1] = sub(G[j - 1], Tj);
    G[j] = Tj;
    const V3 pj = V3{c[4], c[5], c[6]};
    p = p + lam[j] * (pj - pprev);
    qprev = qj, pprev = pj;
  }
  *q_out = qnormalized(q);
  *p_out = p;
}

/// The part of spline_pose_jac that depends on the control points alone: relative rotation j -> j + 1 of two consecutive control points
/// (its Log with the trigonometry of the angle, the J_r^-1 coefficient). A workgroup whose residuals share a window of control points
/// computes these once per pair instead of once per residual and pair (one atan2, one quaternion product, three divisions each).
struct RelPre {
  RelRot rr;
  double D;
  double pad;  // (8 doubles: 16-byte rows in LDS)
};
HSD RelPre rel_precompute(const double* cp_prev, const double* cp_next) {
  RelPre p;
  p.rr = rel_log(load_quat(cp_prev), load_quat(cp_next));
  p.D = jr_inv_coef(p.rr);
  p.pad = 0.0;
  return p;
}

/// spline_pose with the relative rotations handed in (rel[j - 1] = rel_precompute(cp + 8 (j - 1), cp + 8 j)): same arithmetic, same result —
/// the logarithm of a pair (an atan2, two square roots) once per pair of the window instead of once per residual and pair.
template <int K>
HSD void spline_pose_pre(const double* cp, const RelPre* rel, const double* lam, Quat* q_out, V3* p_out) {
  Quat q = load_quat(cp);
  V3 p = V3{cp[4], cp[5], cp[6]};
  V3 pprev = p;
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const double* c = cp + 8 * j;
    double b, cc;
    q = qmul(q, exp_scaled(rel[j - 1].rr, lam[j], &b, &cc));
    const V3 pj = V3{c[4], c[5], c[6]};
    p = p + lam[j] * (pj - pprev);
    pprev = pj;
  }
  *q_out = qnormalized(q);
  *p_out = p;
}

/// spline_pose_jac with the relative rotations handed in: rel[j - 1] = rel_precompute(cp + 8 (j - 1), cp + 8 j). Same arithmetic, same result.
template <int K>
HSD void spline_pose_jac_pre(const double* cp, const RelPre* rel, const double* lam, Quat* q_out, V3* p_out, M3* G) {
  Quat q = load_quat(cp);
  V3 p = V3{cp[4], cp[5], cp[6]};
  V3 pprev = p;
  G[0] = eye();
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const double* c = cp + 8 * j;
    const RelRot rr = rel[j - 1].rr;
    double b, cc;
    q = qmul(q, exp_scaled(rr, lam[j], &b, &cc));
    const double D = rel[j - 1].D;
    const double alpha = 0.5 - b + rr.t2 * (b * D - 0.5 * cc);
    const double beta = D - 0.5 * b + cc - rr.t2 * cc * D;
    M3 X = rodrigues_poly(rr.d, alpha, beta);
    X = scale(lam[j], X);
    const M3 Tj = mul_nt(mul(qmat(q), X), qmat(load_quat(c)));
    G[j - 1] = sub(G[j - 1], Tj);
    G[j] = Tj;
    const V3 pj = V3{c[4], c[5], c[6]};
    p = p + lam[j] * (pj - pprev);
    pprev = pj;
  }
  *q_out = qnormalized(q);
  *p_out = p;
}

/// Full evaluation for the inertial factor (right-perturbation recursion of SURVEY.md A.2b, converted to world-frame
/// perturbations at the end): outputs body-frame w, alpha, world-frame v, a and the 3x3 Jacobian blocks of theta, w, alpha.
template <int K>
struct SplineFull {
  Quat q;
  V3 p, w, al, v, a;
  M3 dth[K], dw[K], dal[K];
  double B[K], Bd[K], Bdd[K];
};

template <int K, bool JAC>
HSD void spline_full(const double* cp, const double* lam, const double* dlam, const double* ddlam, SplineFull<K>* o) {
  V3 p = V3{0, 0, 0}, v = p, a = p;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const double Bj = lam[j] - (j + 1 < K ? lam[j + 1] : 0.0);
    const double Bdj = dlam[j] - (j + 1 < K ? dlam[j + 1] : 0.0);
    const double Bddj = ddlam[j] - (j + 1 < K ? ddlam[j + 1] : 0.0);
    o->B[j] = Bj, o->Bd[j] = Bdj, o->Bdd[j] = Bddj;
    const V3 pj = V3{cp[8 * j + 4], cp[8 * j + 5], cp[8 * j + 6]};
    p = p + Bj * pj, v = v + Bdj * pj, a = a + Bddj * pj;
  }
  o->p = p, o->v = v, o->a = a;

  Quat qprev = load_quat(cp);
  Quat q = qprev;
  V3 w = V3{0, 0, 0}, al = w;
  M3 E[K], W[K], Qm[K];
  if (JAC) {
#pragma unroll
    for (int m = 0; m < K; ++m) E[m] = zero3(), W[m] = zero3(), Qm[m] = zero3();
    E[0] = eye();
  }
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const Quat qj = load_quat(cp + 8 * j);
    const RelRot rr = rel_log(qprev, qj);
    double b, cc;
    const Quat eq = exp_scaled(rr, lam[j], &b, &cc);
    q = qmul(q, eq);
    const M3 A = qmat(eq);
    const V3 w_rot = mul_t(A, w), al_rot = mul_t(A, al);
    const V3 w_new = w_rot + dlam[j] * rr.d;
    const V3 al_new = al_rot + dlam[j] * cross(w_new, rr.d) + ddlam[j] * rr.d;
    if (JAC) {
      const double D = jr_inv_coef(rr);
      const M3 Jri = rodrigues_poly(rr.d, 0.5, D);
      const M3 Jli = transpose(Jri);
      const M3 JrL = rodrigues_poly(rr.d, -b, cc);
#pragma unroll
      for (int m = 0; m < K; ++m) {
        if (m <= j) {  // blocks beyond j are still zero
          E[m] = mul_tn(A, E[m]), W[m] = mul_tn(A, W[m]), Qm[m] = mul_tn(A, Qm[m]);
        }
      }
      const M3 hw_rot = hat(w_rot), hal_rot = hat(al_rot), hd = hat(rr.d), hw_new = hat(w_new);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int blk = j - 1 + c;
        const M3 dd = c == 0 ? scale(-1.0, Jli) : Jri;
        const M3 eta = scale(lam[j], mul(JrL, dd));
        E[blk] = add(E[blk], eta);
        W[blk] = add(W[blk], add(mul(hw_rot, eta), scale(dlam[j], dd)));
        Qm[blk] = add(Qm[blk], add(mul(hal_rot, eta), add(scale(dlam[j], mul(hw_new, dd)), scale(ddlam[j], dd))));
      }
#pragma unroll
      for (int m = 0; m < K; ++m)
        if (m <= j) Qm[m] = sub(Qm[m], scale(dlam[j], mul(hd, W[m])));
    }
    w = w_new, al = al_new, qprev = qj;
  }
  o->q = qnormalized(q);
  o->w = w, o->al = al;
  if (JAC) {
    const M3 R = qmat(o->q);
#pragma unroll
    for (int m = 0; m < K; ++m) {
      const M3 Rm = qmat(load_quat(cp + 8 * m));
      o->dth[m] = mul_nt(mul(R, E[m]), Rm);
      o->dw[m] = mul_nt(W[m], Rm);
      o->dal[m] = mul_nt(Qm[m], Rm);
    }
  }
}

/// spline_full for ONE control point m of the segment (the Jacobian blocks of the other control points are not formed): the recursion
/// over the K - 1 relative rotations is the same, but only E[m], W[m], Qm[m] are carried — 3 instead of 3 K 3 x 3 matrices, so that one
/// residual can be spread over K lanes (one per control point) at a quarter of the registers. Same arithmetic per block as spline_full.
template <int K>
struct SplineCol {
  Quat q;
  V3 p, w, al, v, a;
  M3 dth, dw, dal;  // blocks of control point m
  double Bdd_m;     // second derivative of the (non-cumulative) basis weight of control point m
};

template <int K>
HSD void spline_full_col(const double* cp, const double* lam, const double* dlam, const double* ddlam, const int m, SplineCol<K>* o) {
  V3 p = V3{0, 0, 0}, v = p, a = p;
#pragma unroll
  for (int j = 0; j < K; ++j) {
    const double Bj = lam[j] - (j + 1 < K ? lam[j + 1] : 0.0);
    const double Bdj = dlam[j] - (j + 1 < K ? dlam[j + 1] : 0.0);
    const double Bddj = ddlam[j] - (j + 1 < K ? ddlam[j + 1] : 0.0);
    if (j == m) o->Bdd_m = Bddj;  // (a select per j: indexing a register array with m would put it into scratch memory)
    const V3 pj = V3{cp[8 * j + 4], cp[8 * j + 5], cp[8 * j + 6]};
    p = p + Bj * pj, v = v + Bdj * pj, a = a + Bddj * pj;
  }
  o->p = p, o->v = v, o->a = a;

  Quat qprev = load_quat(cp);
  Quat q = qprev;
  V3 w = V3{0, 0, 0}, al = w;
  M3 E = m == 0 ? eye() : zero3(), W = zero3(), Qm = zero3();
#pragma unroll
  for (int j = 1; j < K; ++j) {
    const Quat qj = load_quat(cp + 8 * j);
    const RelRot rr = rel_log(qprev, qj);
    double b, cc;
    const Quat eq = exp_scaled(rr, lam[j], &b, &cc);
    q = qmul(q, eq);
    const M3 A = qmat(eq);
    const V3 w_rot = mul_t(A, w), al_rot = mul_t(A, al);
    const V3 w_new = w_rot + dlam[j] * rr.d;
    const V3 al_new = al_rot + dlam[j] * cross(w_new, rr.d) + ddlam[j] * rr.d;
    if (m <= j) {  // (blocks beyond j are still zero)
      E = mul_tn(A, E), W = mul_tn(A, W), Qm = mul_tn(A, Qm);
      if (m >= j - 1) {  // the two blocks this relative rotation depends on: m = j - 1 (through -J_l^-1) and m = j (through J_r^-1)
        const double D = jr_inv_coef(rr);
        const M3 Jri = rodrigues_poly(rr.d, 0.5, D);
        const M3 JrL = rodrigues_poly(rr.d, -b, cc);
        const M3 dd = m == j - 1 ? scale(-1.0, transpose(Jri)) : Jri;
        const M3 eta = scale(lam[j], mul(JrL, dd));
        E = add(E, eta);
        W = add(W, add(mul(hat(w_rot), eta), scale(dlam[j], dd)));
        Qm = add(Qm, add(mul(hat(al_rot), eta), add(scale(dlam[j], mul(hat(w_new), dd)), scale(ddlam[j], dd))));
      }
      Qm = sub(Qm, scale(dlam[j], mul(hat(rr.d), W)));
    }
    w = w_new, al = al_new, qprev = qj;
  }
  o->q = qnormalized(q);
  o->w = w, o->al = al;
  const M3 R = qmat(o->q);
  const M3 Rm = qmat(load_quat(cp + 8 * m));
  o->dth = mul_nt(mul(R, E), Rm);
  o->dw = mul_nt(W, Rm);
  o->dal = mul_nt(Qm, Rm);
}

}  // namespace hsd
