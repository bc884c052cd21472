// ORACLE — TEST INFRASTRUCTURE ONLY. Reference-style gradient probe for the CPU restatement: mirrors
// tests/internal/tests/optimizers/evaluators/{bearing,pixel,manifold,inertial}.cpp (random state at 5 Hz over [0,10] s,
// degree-3 uniform basis, every sensor block non-constant) and tests/include/tests/optimizers/evaluators/evaluator.hpp:38-65
// (analytic local Jacobian vs numeric at 1e-5). Numeric side: central differences through Manifold::Plus.
#include <cstdio>
#include <cstdlib>

#include "hs_problem.hpp"

using namespace hso;

static SplitMix64 rng(0x48595045ull);
static double urand(double a, double b) { return a + (b - a) * rng.uniform(); }
static void rand_quat(double* q) {
  double n = 0;
  for (int i = 0; i < 4; ++i) q[i] = urand(-1, 1), n += q[i] * q[i];
  n = std::sqrt(n);
  for (int i = 0; i < 4; ++i) q[i] /= n;
}

static int probe(FactorType type, int k, int kb, bool smooth, bool literal = false) {
  const Basis basis = make_basis(k), bias_basis = make_basis(kb);
  const Layout L = make_layout(type, k, kb);
  const int nb = int(L.sizes.size());
  std::vector<std::vector<double>> blocks(nb);
  std::vector<ManifoldKind> kinds(nb);
  const double dt = 0.2, t_first = 1.0;
  const int i_mid = (k - 1) / 2;
  const double stamp = t_first + dt * (i_mid + urand(0.02, 0.98));
  double qprev[4];
  rand_quat(qprev);
  for (int j = 0; j < k; ++j) {
    blocks[j].resize(8);
    if (smooth) {
      // small relative rotations (realistic trajectory)
      const V3 d = v3(urand(-0.3, 0.3), urand(-0.3, 0.3), urand(-0.3, 0.3));
      const Quat qn = qmul(Quat{qprev[0], qprev[1], qprev[2], qprev[3]}, so3_exp_q(d));
      qprev[0] = qn.x, qprev[1] = qn.y, qprev[2] = qn.z, qprev[3] = qn.w;
      for (int i = 0; i < 4; ++i) blocks[j][i] = qprev[i];
    } else {
      rand_quat(blocks[j].data());  // Mock<SE3>::Random (tests/include/tests/random.hpp:51-66)
    }
    for (int i = 0; i < 3; ++i) blocks[j][4 + i] = urand(-1, 1);
    blocks[j][7] = t_first + dt * j;
    kinds[j] = kManifoldControlPoint;
  }
  std::vector<double> meas(7);
  int b = k;
  auto add = [&](int n, ManifoldKind m) {
    blocks[b].resize(n);
    kinds[b] = m;
    return blocks[b++].data();
  };
  if (type == kPixel || type == kBearing) {
    double* T = add(7, kManifoldSE3);
    rand_quat(T);
    for (int i = 0; i < 3; ++i) T[4 + i] = urand(-1, 1);
    double* in = add(4, kManifoldEuclidean);
    in[0] = 367.215, in[1] = 248.375, in[2] = 458.654, in[3] = 457.296;
    double* di = add(4, kManifoldEuclidean);
    di[0] = -0.28340811 + urand(-0.05, 0.05), di[1] = 0.07395907 + urand(-0.05, 0.05);
    di[2] = 1.76187114e-05 + urand(-0.05, 0.05), di[3] = 0.00019359 + urand(-0.05, 0.05);
    double* lm = add(3, kManifoldEuclidean);
    // Landmark in front of the camera (so the projection is well defined): place via the value-only evaluator.
    std::vector<const double*> ps(nb);
    for (int i = 0; i < nb; ++i) ps[i] = blocks[i].data();
    lm[0] = lm[1] = lm[2] = 0;
    StateResult S;
    state_evaluate(basis, ps.data(), stamp, 0, false, &S);
    const Pose T_ws = group_plus(Pose{S.s.q, S.s.R, S.s.p}, make_pose(T), nullptr, nullptr);
    const V3 ps_cam = v3(urand(-1, 1), urand(-1, 1), urand(3, 8));
    const V3 pw = T_ws.R * ps_cam + T_ws.p;
    lm[0] = pw[0], lm[1] = pw[1], lm[2] = pw[2];
    if (type == kPixel) {
      meas[0] = urand(0, 752), meas[1] = urand(0, 480);
    } else {
      double q[4];
      rand_quat(q);
      const M3 R = qmat(Quat{q[0], q[1], q[2], q[3]});
      meas[0] = R(0, 0), meas[1] = R(1, 0), meas[2] = R(2, 0);
    }
  } else if (type == kPrior) {
    double* T = add(7, kManifoldSE3);
    rand_quat(T);
    for (int i = 0; i < 3; ++i) T[4 + i] = urand(-1, 1);
    rand_quat(meas.data());
    for (int i = 0; i < 3; ++i) meas[4 + i] = urand(-1, 1);
  } else {
    double* T = add(7, kManifoldSE3);
    rand_quat(T);
    for (int i = 0; i < 3; ++i) T[4 + i] = urand(-1, 1);
    double* ig = add(6, kManifoldEuclidean);
    double* ia = add(6, kManifoldEuclidean);
    // literal: the in-tree Jacobian (inertial.cpp:131-198) at the reference's own test point — Mock<IMU>::Create(), tests/include/tests/
    // sensors/imu.hpp:20-26: random T_bs, identity intrinsics, S_g = X_a = 0 — where it is exact; otherwise the exact form at random values
    const double spread = literal ? 0.0 : 1.0;
    for (int i = 0; i < 6; ++i) ig[i] = (i < 3 ? 1.0 : 0.0) + spread * urand(-0.1, 0.1), ia[i] = (i < 3 ? 1.0 : 0.0) + spread * urand(-0.1, 0.1);
    double* sg = add(9, kManifoldEuclidean);
    double* xa = add(9, kManifoldEuclidean);
    for (int i = 0; i < 9; ++i) sg[i] = spread * urand(-0.01, 0.01), xa[i] = spread * urand(-0.05, 0.05);
    const double bdt = 10.0, bt0 = stamp - bdt * ((kb - 1) / 2) - urand(0.1, 9.9);
    for (int j = 0; j < 2 * kb; ++j) {
      double* bc = add(4, kManifoldBiasPoint);
      for (int i = 0; i < 3; ++i) bc[i] = urand(-1, 1);
      bc[3] = bt0 + bdt * (j % kb);
    }
    double* g = add(3, kManifoldSphere3);
    double q[4];
    rand_quat(q);
    const M3 R = qmat(Quat{q[0], q[1], q[2], q[3]});
    for (int i = 0; i < 3; ++i) g[i] = 9.80665 * R(i, 0);
    for (int i = 0; i < 6; ++i) meas[i] = urand(-1, 1);
  }
  std::vector<const double*> ps(nb);
  std::vector<std::vector<double>> jb(nb);
  std::vector<double*> jac(nb);
  for (int i = 0; i < nb; ++i) {
    ps[i] = blocks[i].data();
    jb[i].assign(size_t(L.num_residuals) * L.sizes[i], 0.0);
    jac[i] = jb[i].data();
  }
  const CostContext ctx = {type, &basis, &bias_basis, stamp, meas.data(), literal};
  double r0[6];
  cost_evaluate(ctx, L, ps.data(), r0, jac.data());
  int fails = 0;
  double worst = 0;
  for (int i = 0; i < nb; ++i) {
    const int amb = L.sizes[i], loc = manifold_local_size(kinds[i], amb);
    std::vector<double> Jl(size_t(L.num_residuals) * loc);
    to_local(kinds[i], amb, L.num_residuals, ps[i], jac[i], Jl.data());
    for (int c = 0; c < loc; ++c) {
      const double h = 1e-6;
      double d[6] = {0, 0, 0, 0, 0, 0}, xp[9], xm[9], rp[6], rm[6];
      // manifold_plus of the Euclidean kinds uses `ambient` entries of d
      std::vector<double> dv(amb > 6 ? amb : 6, 0.0);
      (void)d;
      dv[c] = h;
      manifold_plus(kinds[i], amb, ps[i], dv.data(), xp);
      dv[c] = -h;
      manifold_plus(kinds[i], amb, ps[i], dv.data(), xm);
      const double* keep = ps[i];
      ps[i] = xp;
      cost_evaluate(ctx, L, ps.data(), rp, nullptr);
      ps[i] = xm;
      cost_evaluate(ctx, L, ps.data(), rm, nullptr);
      ps[i] = keep;
      for (int r = 0; r < L.num_residuals; ++r) {
        const double num = (rp[r] - rm[r]) / (2 * h), ana = Jl[size_t(r) * loc + c];
        const double err = std::fabs(num - ana) / std::max(1.0, std::max(std::fabs(num), std::fabs(ana)));
        worst = std::max(worst, err);
        if (err > 1e-5) {
          if (fails < 8) std::printf("  type %d k %d block %d col %d row %d: analytic %.10g numeric %.10g\n", type, k, i, c, r, ana, num);
          ++fails;
        }
      }
    }
  }
  std::printf("probe type=%d k=%d smooth=%d: worst rel err %.3g  %s\n", type, k, int(smooth), worst, fails ? "FAIL" : "ok");
  return fails;
}

int main() {
  int fails = 0;
  for (int rep = 0; rep < 8; ++rep)
    for (int k : {4, 6})
      for (int t = 0; t < 4; ++t) {
        fails += probe(FactorType(t), k, 4, true);
        fails += probe(FactorType(t), k, 4, false);
        if (t == kInertial) fails += probe(FactorType(t), k, 4, true, /*literal=*/true);
      }
  // basis sanity (SURVEY.md A.1 verified values)
  const Basis b4 = make_basis(4);
  const double ref4[4][4] = {{6, 0, 0, 0}, {5, 3, -3, 1}, {1, 3, 3, -2}, {0, 0, 0, 1}};
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j)
      if (std::fabs(6 * b4.Ct[i][j] - ref4[i][j]) > 1e-12) std::printf("basis4 mismatch %d %d\n", i, j), ++fails;
  const Basis b6 = make_basis(6);
  const double ref6[6][6] = {{120, 0, 0, 0, 0, 0}, {119, 5, -10, 10, -5, 1}, {93, 55, -30, -10, 15, -4}, {27, 55, 30, -10, -15, 6}, {1, 5, 10, 10, 5, -4}, {0, 0, 0, 0, 0, 1}};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j)
      if (std::fabs(120 * b6.Ct[i][j] - ref6[i][j]) > 1e-10) std::printf("basis6 mismatch %d %d: %g\n", i, j, 120 * b6.Ct[i][j]), ++fails;
  std::printf(fails ? "SELFTEST FAILED (%d)\n" : "SELFTEST OK\n", fails);
  return fails ? 1 : 0;
}
