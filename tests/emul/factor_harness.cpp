// factor_harness.cpp — TEST INFRASTRUCTURE (tests/test_emulated_factor.py): the look-ahead band Cholesky of the reduced system, compiled FROM
// THE PRODUCT'S KERNEL SOURCE for the host (tests/emul/hip/hip_runtime.h: one std::thread per lane) and run on a band system a Python test
// hands over.
//   k_band_factor_la<1, NCW>     (hyperslam_amd/csrc/kernels_factor.hpp), one-ended (grid 1) and from both ends (grid 2),
//   k_band_backward_sb           (kernels_backward_sb.hpp): the sweeps in super-blocks of four block rows + their inverse builders,
// with the jobs and launch shapes launch_factor (host_launch.hpp) uses. Output: the factor rows, the inverted diagonal blocks and the
// forward-solved right-hand side of each job — compared by the test with numpy's Cholesky factor of the same matrix (near end: its leading
// rows; far end: the leading rows of the reversed matrix; middle rows: the factor of the Schur complement both ends leave on them) — and
// the solution, the step, the scaled step and the two sums of the model cost change, compared with numpy's solve.
// Usage: factor_harness <system.bin> <out.bin>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "hip/hip_runtime.h"

thread_local dim3 threadIdx;
thread_local unsigned hs_emul::exchange_count = 0;
dim3 blockIdx, blockDim, gridDim;

#include "../../hyperslam_amd/csrc/kernels_common.hpp"
#include "../../hyperslam_amd/csrc/kernels_linearize.hpp"
#include "../../hyperslam_amd/csrc/kernels_sensor.hpp"
#include "../../hyperslam_amd/csrc/kernels_schur.hpp"
#include "../../hyperslam_amd/csrc/kernels_border.hpp"
#include "../../hyperslam_amd/csrc/kernels_factor.hpp"
#include "../../hyperslam_amd/csrc/kernels_factor_mx.hpp"
#include "../../hyperslam_amd/csrc/kernels_dense_mx.hpp"
#include "../../hyperslam_amd/csrc/kernels_backward_sb.hpp"

namespace hs {
HSD void begin_iteration(const Tables&, double, double, bool) {}  // (Tables::bookkeep = 0 in the harness: never reached)
}  // namespace hs

using namespace hs;

template <class T>
static std::vector<T> read_vec(FILE* f, size_t n) {
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
  return v;
}
static void write_vec(FILE* f, const std::vector<double>& v) { fwrite(v.data(), sizeof(double), v.size(), f); }

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  FILE* in = fopen(argv[1], "rb");
  if (!in) return 1;
  const std::vector<int> hdr = read_vec<int>(in, 6);  // np, bw, ends | f0 << 8, kernel variant, nb, 0
  // hdr[2]: bit 0 = from both ends; bits 8.. = f0, the leading block rows of constant control points (decoupled: k_factor_decoupled_rows, the
  // chain of the one-ended kernels and the sweeps start behind them — the sliding window's frozen prefix, launch_factor)
  const int np = hdr[0], bw = hdr[1], two_ended = hdr[2] & 1, f0 = hdr[2] >> 8, ncb = 6 * bw, n_blk = np / 6, w_mid = bw - 1;
  const std::vector<double> Sb = read_vec<double>(in, size_t(np) * ncb), g = read_vec<double>(in, np);
  const std::vector<double> Sb2 = read_vec<double>(in, size_t(np) * ncb), g2 = read_vec<double>(in, np);  // the reversed system
  const std::vector<double> scale_p = read_vec<double>(in, np), g_full = read_vec<double>(in, np), D2p = read_vec<double>(in, np);
  // border unknowns (bias splines, gravity): S_pb (np x nb), S_bb (nb x nb), g_b, their scaling and damping, and per group of kBorderCols
  // columns the first block row with a non-zero entry (Tables::bfwd_start)
  const int nb = hdr[4], n_groups = (nb + kBorderCols - 1) / kBorderCols;
  const std::vector<double> Spb = read_vec<double>(in, size_t(np) * nb), Sbb = read_vec<double>(in, size_t(nb) * nb), gb = read_vec<double>(in, nb);
  const std::vector<double> scale_b = read_vec<double>(in, nb), D2b = read_vec<double>(in, nb);
  const std::vector<int> bfwd_start = read_vec<int>(in, n_groups);
  fclose(in);
  if (hdr[3] == 7) {  // k_dense_solve_mx in border mode (Tables::dense_border): the border Schur complement C | h of a two-ended bordered system -> x_b.
                      // Input: S_bb stands for C, g_b for h; the dense copy is filled the way launch_factor does it (k_dense_border_init once, then
                      // k_border_schur's stores: both triangles, the right-hand side as row and column nb).
    if (nb < 1 || nb + 1 > 16 * kDxTiles) {
      fprintf(stderr, "border outside the kernel's range\n");
      return 3;
    }
    std::vector<double> ut(size_t(256) * 256, 0.0), xb(nb + 1, 7.0), xpart(64, 0.0), dense(size_t(kDenseLd) * kDenseLd, 7.0);
    hs_emul::launch(dim3(4), dim3(kBlock), 0, [&] { k_dense_border_init(dense.data(), nb); });
    for (int b = 0; b < nb; ++b) {
      for (int c = 0; c < nb; ++c) dense[size_t(b) * kDenseLd + c] = Sbb[size_t(b) * nb + c];
      dense[size_t(b) * kDenseLd + nb] = dense[size_t(nb) * kDenseLd + b] = gb[b];
    }
    DevState st{};
    Tables T{};
    T.np = np, T.bw = bw, T.nb = nb, T.st = &st, T.xpart = xpart.data(), T.xb = xb.data();
    T.dense = dense.data(), T.dense_f0 = np / 6, T.dense_border = 1;
    hs_emul::launch(dim3(1), dim3(kDxThreads), size_t(kDxLdsDoubles) * sizeof(double), [&] { k_dense_solve_mx(T, np / 6, ut.data()); });
    FILE* out = fopen(argv[2], "wb");
    const int res[4] = {-1, 0, st.chol_failed, 0};
    fwrite(res, sizeof(int), 4, out);
    xb.resize(nb);
    write_vec(out, xb);
    fclose(out);
    return 0;
  }
  if (hdr[3] == 6) {  // k_dense_solve_mx (kernels_dense_mx.hpp): factorisation, border and both sweeps of a small system in one launch
    if (two_ended || !dense_mx_fits(n_blk - f0, nb)) {
      fprintf(stderr, "system outside the kernel's range\n");
      return 3;
    }
    std::vector<double> ut(size_t(256) * 256, 0.0), step_p(np, 7.0), delta_p(np, 7.0), xb(nb + 1, 0.0), delta_b(nb + 1, 0.0), xpart(64, 0.0);
    DevState st{};
    Tables T{};
    T.np = np, T.bw = bw, T.nb = nb, T.st = &st, T.xpart = xpart.data();
    T.Sb = const_cast<double*>(Sb.data()), T.g_s = const_cast<double*>(g.data());
    T.Spb = const_cast<double*>(Spb.data()), T.Sbb = const_cast<double*>(Sbb.data()), T.gb_s = const_cast<double*>(gb.data());
    T.scale_p = const_cast<double*>(scale_p.data()), T.g_full = const_cast<double*>(g_full.data()), T.D2p = const_cast<double*>(D2p.data());
    T.scale_b = const_cast<double*>(scale_b.data()), T.D2b = const_cast<double*>(D2b.data());
    T.step_p = step_p.data(), T.delta_p = delta_p.data(), T.xb = xb.data(), T.delta_b = delta_b.data();
    // Tables::dense: the dense copy of the scaled, damped system the finalisation kernels write for this kernel (k_finalize_reduced,
    // finalize_border_body: row-major, leading dimension 256, both triangles, identity on the padding; the GPU suite covers those writers)
    std::vector<double> dense(size_t(kDenseLd) * kDenseLd, 7.0);  // (what the writers do not touch is never used)
    {
      const int n_pose = np - 6 * f0, n_dense = n_pose + nb, n_pad = 16 * ((n_dense + 1 + 15) / 16);
      auto entry = [&](int i, int j) -> double {  // i <= j
        if (j == n_dense) return i == j ? 1e300 : (i < n_pose ? g[6 * f0 + i] : gb[i - n_pose]);  // the right-hand side as column n_dense
        if (j > n_dense) return i == j ? 1.0 : 0.0;
        if (j < n_pose) {
          const int ri = 6 * f0 + i, c = 6 * f0 + j - 6 * (ri / 6);
          return c < ncb ? Sb[size_t(ri) * ncb + c] : 0.0;
        }
        if (i < n_pose) return Spb[size_t(6 * f0 + i) * nb + (j - n_pose)];
        return Sbb[size_t(i - n_pose) * nb + (j - n_pose)];
      };
      for (int i = 0; i < n_pad; ++i)
        for (int j = i; j < n_pad; ++j) dense[size_t(i) * kDenseLd + j] = dense[size_t(j) * kDenseLd + i] = entry(i, j);
    }
    T.dense = dense.data(), T.dense_f0 = f0;
    hs_emul::launch(dim3(1), dim3(kDxThreads), size_t(kDxLdsDoubles) * sizeof(double), [&] { k_dense_solve_mx(T, f0, ut.data()); });
    FILE* out = fopen(argv[2], "wb");
    const int res[4] = {-1, 0, st.chol_failed, 0};
    fwrite(res, sizeof(int), 4, out);
    const std::vector<double> zUb(size_t(np) * ncb, 0.0), zUbk(size_t(24) * n_blk, 0.0), zy(np, 0.0);
    write_vec(out, zUb), write_vec(out, zUbk), write_vec(out, zy), write_vec(out, zUb), write_vec(out, zUbk), write_vec(out, zy);
    std::vector<double> xsol(np);
    for (int i = 0; i < np; ++i) xsol[i] = -step_p[i];
    write_vec(out, xsol), write_vec(out, step_p), write_vec(out, delta_p);
    write_vec(out, {st.g_dot_step_pose, st.d2_step2_pose, st.g_dot_step_far, st.d2_step2_far});
    xb.resize(nb), delta_b.resize(nb);
    write_vec(out, xb), write_vec(out, delta_b);
    fclose(out);
    return 0;
  }
  const int ncw = la_compute_waves(bw);
  // hdr[3]: which one-ended kernel (launch_factor's rules pick one by band width and length; the test asks for each where it applies)
  //   0 look-ahead (k_band_factor_la), 1 k_band_factor<1> (bw^2 <= 256 lanes), 2 k_band_factor<2> (bw <= 21), 3 k_band_factor_wide, 4 k_dense_factor,
  //   5 k_band_factor_mx (trailing window in the accumulators of the f64 matrix cores; one-ended and from both ends)
  const int variant = hdr[3];
  if (variant == 5 && (!mx_fits(bw) || (two_ended && n_blk < 4 * bw) || f0 > 0)) {
    fprintf(stderr, "band width / length outside the kernel's range\n");
    return 3;
  }
  if ((variant == 0 && (ncw == 0 || (two_ended && n_blk < 4 * bw))) || (variant != 0 && variant != 5 && two_ended) || (variant == 1 && bw * bw > kCholThreads) ||
      (variant == 2 && bw > 21) || (variant == 4 && !dense_factor_fits(n_blk - f0, std::min(bw, n_blk - f0)))) {
    fprintf(stderr, "band width / length outside the kernel's range\n");
    return 3;
  }
  std::vector<double> Ub(size_t(np) * ncb, 0.0), Ubk(size_t(24) * n_blk, 0.0), yb(np, 0.0), Ub2 = Ub, Ubk2 = Ubk, yb2 = yb;
  std::vector<double> win(size_t(6 * w_mid) * (ncb + 1), 0.0), xpart(8 * 1024, 0.0);
  DevState st{};
  std::vector<unsigned> join_flag(kBfFlagBase + 512 + 4 * kProgressStride, 0u);  // (prepare(): junction word, super-block flags of the sweeps, one flag per column group of k_border_forward2)
  Tables T{};
  T.np = np, T.bw = bw, T.st = &st, T.join_flag = join_flag.data(), T.join_epoch = 1, T.xpart = xpart.data();
  const size_t la_lds = (size_t(42) * (ncb + 2) + size_t(np) + 48) * sizeof(double);  // launch_factor
  int m = -1, mB = 0;
  static const double zero = 0.0;
  if (two_ended) {  // launch_factor: the near end takes three block rows more than the far end
    m = std::min((n_blk - w_mid) / 2 + two_ended_lead(variant == 5), n_blk - w_mid - w_mid), mB = n_blk - w_mid - m;
    T.fj[0] = FactorJob{Sb.data(), g.data(), Ub.data(), Ubk.data(), yb.data(), win.data(), m + w_mid, m};
    T.fj[1] = FactorJob{Sb2.data(), g2.data(), Ub2.data(), Ubk2.data(), yb2.data(), win.data(), mB, -1};
    T.mj[0] = MfmaJob{Sb2.data(), g.data(), Ub.data(), Ubk.data(), yb.data(), win.data(), m + w_mid, m, m + w_mid, INT_MAX, 0, &zero};
    T.mj[1] = MfmaJob{Sb.data(), g2.data(), Ub2.data(), Ubk2.data(), yb2.data(), win.data(), mB, -1, mB + w_mid, mB, 1, &zero};
    if (variant == 5 && nb > 0) {  // launch_factor: a bordered system on k_band_factor_mx publishes its progress for the forward sweep of the border columns
      T.mj[0].progress = join_flag.data() + kBfFlagBase + 512, T.mj[1].progress = T.mj[0].progress + 2 * kProgressStride;
      T.mj[0].progress_base = T.mj[1].progress_base = 7u << 12;
    }
  } else {
    T.fj[0] = FactorJob{Sb.data(), g.data(), Ub.data(), Ubk.data(), yb.data(), nullptr, n_blk, -1};
    T.mj[0] = MfmaJob{Sb2.data(), g.data(), Ub.data(), Ubk.data(), yb.data(), nullptr, n_blk, -1, n_blk, INT_MAX, 0, &zero};
    T.mj[1] = T.mj[0];
  }
  const dim3 grid(two_ended ? 2 : 1);
  const std::vector<unsigned> far_first = {1, 0};  // (workgroup 0 waits at the junction for workgroup 1's window)
  T.Sb = const_cast<double*>(Sb.data()), T.g_s = const_cast<double*>(g.data()), T.Ub = Ub.data(), T.Ubk = Ubk.data(), T.ybuf = yb.data();
  const Tables Tfull = T;
  if (f0 > 0) {  // launch_factor: the decoupled rows one wave each, the kernels below on the trailing sub-matrix (the band storage is row relative)
    if (two_ended || f0 >= n_blk) return 3;
    if (variant != 4) hs_emul::launch(dim3(f0), dim3(64), 0, [&] { k_factor_decoupled_rows(Tfull, f0); });
    T.Sb += size_t(6 * f0) * ncb, T.g_s += 6 * f0, T.Ub += size_t(6 * f0) * ncb, T.Ubk += size_t(24) * f0, T.ybuf += 6 * f0, T.np -= 6 * f0;
    T.fj[0] = FactorJob{T.Sb, T.g_s, T.Ub, T.Ubk, T.ybuf, nullptr, T.np / 6, -1};
  }
  const size_t chol_lds = (size_t(24) * (ncb + 2) + size_t(np)) * sizeof(double);  // launch_factor
  if (variant == 1)
    hs_emul::launch(dim3(1), dim3(kCholThreads + kCholIo), chol_lds, [&] { k_band_factor<1>(T); });
  else if (variant == 2)
    hs_emul::launch(dim3(1), dim3(kCholThreads + kCholIo), chol_lds, [&] { k_band_factor<2>(T); });
  else if (variant == 3)
    hs_emul::launch(dim3(1), dim3(kWideThreads), size_t(12) * (ncb + 2) * sizeof(double), [&] { k_band_factor_wide(T); });
  else if (variant == 4)  // (the dense kernel writes the decoupled rows with extra workgroups of its own launch)
    hs_emul::launch(dim3(1 + f0), dim3(kDenseThreads), (size_t(12) * (ncb + 8) + size_t(32) * (n_blk - f0)) * sizeof(double), [&] { k_dense_factor(T, f0); });
  else if (variant == 5)
    hs_emul::launch(grid, dim3(kMxThreads), size_t(kMxLds) * sizeof(double), [&] {
      if (T.mj[0].progress)  // (launch_factor: the instance that publishes its progress)
        mx_wide(bw) ? k_band_factor_mx<true, true>(T) : k_band_factor_mx<false, true>(T);
      else
        mx_wide(bw) ? k_band_factor_mx<true>(T) : k_band_factor_mx<false>(T);
    },
                    two_ended ? far_first : std::vector<unsigned>{});
  else if (ncw == 3)
    hs_emul::launch(grid, dim3(la_threads(3)), la_lds, [&] { k_band_factor_la<1, 3>(T); }, two_ended ? far_first : std::vector<unsigned>{});
  else
    hs_emul::launch(grid, dim3(la_threads(4)), la_lds, [&] { k_band_factor_la<1, 4>(T); }, two_ended ? far_first : std::vector<unsigned>{});
  // ---- bordered system (launch_factor): Z = U^-T S_pb in the elimination order of the factorisation, C = S_bb - Z'Z, h = g_b - Z'y,
  //      dense Cholesky of the border, y' = y - Z x_b ----
  std::vector<double> Zb(size_t(np) * std::max(nb, 1), 0.0), Cb(size_t(nb) * nb + 1, 0.0), hb(nb + 1, 0.0), xb(nb + 1, 0.0), delta_b(nb + 1, 0.0);
  std::vector<double> handover(size_t(std::max(n_groups, 1)) * 6 * w_mid * kBorderCols + 1, 0.0);
  if (nb > 0) {
    if (f0 > 0 || (variant != 0 && variant != 5)) return 3;
    Tables Tb = T;
    Tb.nb = nb, Tb.Spb = const_cast<double*>(Spb.data()), Tb.Sbb = const_cast<double*>(Sbb.data()), Tb.gb_s = const_cast<double*>(gb.data());
    Tb.Zb = Zb.data(), Tb.Cb = Cb.data(), Tb.hb = hb.data(), Tb.xb = xb.data(), Tb.bfwd_start = bfwd_start.data();
    Tb.ybuf = yb.data(), Tb.ybuf2 = nullptr, Tb.y_split = np;
    const int fwd_threads = std::max(128, 64 * ((6 * w_mid + 63) / 64));  // one lane per pending row
    const int n_tiles = (nb + kSchurTile - 1) / kSchurTile;
    if (two_ended) {
      Tb.ybuf2 = yb2.data(), Tb.y_split = 6 * (m + w_mid);
      Tb.join_epoch = 3;
      // (k_band_factor_mx: the sweep follows the factorisation's progress words — here the factorisation has finished; + the polling wave)
      const unsigned* prog = variant == 5 ? T.mj[0].progress : nullptr;
      hs_emul::launch(dim3(n_groups, 2), dim3(fwd_threads + (prog ? 64 : 0)), size_t(np) * kBorderLd * sizeof(double), [&] {
        k_border_forward2(Tb, BfJob{Ub.data(), Ubk.data(), m + w_mid, 0, prog, 7u << 12}, BfJob{Ub2.data(), Ubk2.data(), mB, 1, prog ? prog + 2 * kProgressStride : nullptr, 7u << 12}, m, 0, 1,
                          handover.data());
      });
      if (prog && (prog[0] != (7u << 12) + unsigned(m + w_mid) || prog[kProgressStride] != prog[0] || prog[2 * kProgressStride] != (7u << 12) + unsigned(mB) || prog[3 * kProgressStride] != prog[2 * kProgressStride])) return 9;
      hs_emul::launch(dim3(n_tiles, n_tiles), dim3(kBlock), 0, [&] { k_border_schur(Tb, 0, 1, m); });
    } else {
      hs_emul::launch(dim3(n_groups), dim3(fwd_threads), size_t(np) * kBorderLd * sizeof(double), [&] { k_border_forward(Tb, 0, 1); });
      hs_emul::launch(dim3(n_tiles, n_tiles), dim3(kBlock), 0, [&] { k_border_schur(Tb, 0, 1, n_blk); });
    }
    {  // launch_border_solve
      const int R = std::max(3, (nb + 1 + 15) / 16), N = 16 * R;
      const size_t lds = (size_t(4) * N + size_t(nb) * (N + 1) + nb) * sizeof(double);
      if (nb + 1 > 128)
        hs_emul::launch(dim3(1), dim3(kBlock), (size_t(nb + 1) * (nb + 1) + nb) * sizeof(double), [&] { k_border_solve(Tb); });
      else if (R == 3)
        hs_emul::launch(dim3(1), dim3(kBlock), lds, [&] { k_border_solve_reg<3>(Tb); });
      else if (R == 4)
        hs_emul::launch(dim3(1), dim3(kBlock), lds, [&] { k_border_solve_reg<4>(Tb); });
      else if (R == 5)
        hs_emul::launch(dim3(1), dim3(kBlock), lds, [&] { k_border_solve_reg<5>(Tb); });
      else if (R == 6)
        hs_emul::launch(dim3(1), dim3(kBlock), lds, [&] { k_border_solve_reg<6>(Tb); });
      else if (R == 7)
        hs_emul::launch(dim3(1), dim3(kBlock), lds, [&] { k_border_solve_reg<7>(Tb); });
      else
        hs_emul::launch(dim3(1), dim3(kBlock), lds, [&] { k_border_solve_reg<8>(Tb); });
    }
    hs_emul::launch(dim3((np + kBlock / 64 - 1) / (kBlock / 64)), dim3(kBlock), 0, [&] { k_border_apply(Tb); });
    T.nb = nb, T.xb = xb.data(), T.delta_b = delta_b.data(), T.scale_b = const_cast<double*>(scale_b.data()), T.gb_s = const_cast<double*>(gb.data());
    T.D2b = const_cast<double*>(D2b.data());
  }
  // ---- the sweeps (launch_factor: the inverses of the diagonal super-blocks come from extra workgroups of the same launch) ----
  std::vector<double> Vb(size_t(sb_count(n_blk) + 1) * kSbN * kSbN, 0.0), Vb2 = Vb, xsol(np, 0.0), step_p(np, 0.0), delta_p(np, 0.0);
  if (f0 > 0) T = Tfull;  // (the sweeps run on the whole factor and stop above block row f0)
  T.join_epoch = 2;
  T.scale_p = const_cast<double*>(scale_p.data()), T.g_full = const_cast<double*>(g_full.data()), T.D2p = const_cast<double*>(D2p.data());
  T.xsol = xsol.data(), T.step_p = step_p.data(), T.delta_p = delta_p.data();
  if (6 * (bw - 1) > 96) {  // wide bands (long feature tracks): one block row per step, one lane per pending row (launch_factor)
    hs_emul::launch(dim3(1), dim3(kCholThreads), 2 * size_t(np) * sizeof(double), [&] { k_band_backward(T, f0); });
    for (int i = 0; i < np; ++i) xsol[i] = -step_p[i];
  } else if (two_ended) {
    const BackJob j0{Ub.data(), Ubk.data(), yb.data(), Vb.data(), nullptr, m + w_mid, 0, 0};
    const BackJob j1{Ub2.data(), Ubk2.data(), yb2.data(), Vb2.data(), nullptr, mB, w_mid, 1};
    const size_t g_lds = size_t(6 * (bw - 1)) * (6 * (bw - 1) | 1) * sizeof(double);
    const size_t lds = std::max((2 * size_t(np) + 32) * sizeof(double) + g_lds + sb_phase_a_doubles(bw) * sizeof(double), size_t(3 * kSbN * (kSbN + 1)) * sizeof(double));
    const unsigned n_wg = 2 + sb_count(m + w_mid) + sb_count(mB);
    std::vector<unsigned> order;  // builders, then the near sweep (publishes the middle solution when the far one does not redo it), then the far sweep
    for (unsigned w = 2; w < n_wg; ++w) order.push_back(w);
    order.push_back(0), order.push_back(1);
    hs_emul::launch(dim3(n_wg), dim3(kCholThreads), lds, [&] { k_band_backward_sb(T, j0, j1, m, 2, 0); }, order);
  } else {
    const BackJob j0{Ub.data(), Ubk.data(), yb.data(), Vb.data(), nullptr, n_blk, 0, 0};
    const size_t lds = std::max((2 * size_t(np) + 32) * sizeof(double), size_t(3 * kSbN * (kSbN + 1)) * sizeof(double));
    const unsigned n_wg = 1 + sb_count(n_blk);
    std::vector<unsigned> order;
    for (unsigned w = 1; w < n_wg; ++w) order.push_back(w);
    order.push_back(0);
    hs_emul::launch(dim3(n_wg), dim3(kCholThreads), lds, [&] { k_band_backward_sb(T, j0, j0, -1, 1, f0); }, order);
  }
  FILE* out = fopen(argv[2], "wb");
  const int res[4] = {m, mB, st.chol_failed, 0};
  fwrite(res, sizeof(int), 4, out);
  write_vec(out, Ub), write_vec(out, Ubk), write_vec(out, yb), write_vec(out, Ub2), write_vec(out, Ubk2), write_vec(out, yb2);
  write_vec(out, xsol), write_vec(out, step_p), write_vec(out, delta_p);
  write_vec(out, {st.g_dot_step_pose, st.d2_step2_pose, st.g_dot_step_far, st.d2_step2_far});
  xb.resize(nb), delta_b.resize(nb);
  write_vec(out, xb), write_vec(out, delta_b);
  if (getenv("HS_EMUL_DEBUG")) {
    FILE* d = fopen(getenv("HS_EMUL_DEBUG"), "wb");
    write_vec(d, Zb), write_vec(d, Cb), write_vec(d, hb);
    fclose(d);
  }
  fclose(out);
  return 0;
}
