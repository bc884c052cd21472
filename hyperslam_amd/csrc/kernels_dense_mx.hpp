// kernels_dense_mx.hpp — the whole solve of a SMALL reduced system in one launch, trailing matrix in the accumulators of the f64 matrix cores
// (part of kernels.hpp; included once by capi.hip through it).
//
// What it is for. The sliding window HyperSLAM actually runs (max_window 3.0 s at 0.1 s separation,
// /root/reference/internal/hyper/optimizers/abstract.cpp:26-28, settings.yaml:145-148) leaves ~33 free control points whose landmark tracks
// are as long as the window: the "band" of the reduced system IS the matrix (~200 x 200), plus, with an IMU, a border of 6 n_bias + 2
// unknowns (~45). Up to round 5 such a window paid, per LM iteration, k_dense_factor (2.1 us per block row, 72 us), then — bordered — the
// forward sweep of the border columns, the border Schur complement, its dense Cholesky and y' = y - Z x_b in four more launches, then
// k_band_backward: six dependent launches, ~150 us, a third of the iteration. Replaces what CHOLMOD does behind
// /root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:46-48 (SPARSE_NORMAL_CHOLESKY) for these windows.
//
// Here the bordered system [S_pp S_pb; S_pb' S_bb] x = [g_p; g_b] of N = 6 n_free + n_b <= 256 unknowns is ONE dense Cholesky solve in ONE
// workgroup of eight waves: the border is simply the last columns of the matrix — Z = U^-T S_pb, the border Schur complement and its
// factor are what the elimination produces on the way (nothing in the algebra distinguishes them).
//
//   data      16 x 16 tiles (I, J), I <= J, of the padded matrix (identity on the padding) live in MFMA accumulators for the whole
//             factorisation: register r of lane l = entry (4 r + (l >> 4), l & 15) of the tile. Wave w = (a, b) owns the tiles with
//             I mod 2 = a, J in {b, 7 - b, 8 + b, 15 - b} (2-D cyclic, the columns dealt in a zigzag: 18 or 16 tiles per wave; every step's
//             tile row and trailing matrix are spread over all waves; a wave loads 8 + 4 operand columns per step instead of two per tile).
//   step k    (1) the owners of tile row k write it to LDS                                                                --- barrier ---
//             (2) panel, one lane per column of the row (the 16 columns of the diagonal tile and the right-hand side redundantly in every
//                 wave, like the six extra lanes of k_band_factor_mx's panel): right-looking Cholesky of the 16 x 16 diagonal block fused with
//                 the forward substitution of every column, pivots and multipliers broadcast with v_readlane (SGPR operands) — plain
//                 substitution, no explicit inverse on the forward path. Sixteen extra columns start as the identity and come out as the rows
//                 of W_k = U_kk^-1 (for the backward sweep only). The panel also updates the right-hand side (g_c -= x_c . y_k), writes
//                 X(k, :) back to LDS as the operand of the update and to memory as the factor rows                        --- barrier ---
//             (3) tile (I, J) -= X(k, I)' X(k, J) for k < I <= J: four v_mfma_f64_16x16x4_f64 per tile, A operand (negated) shared by the
//                 tiles of a row, B operands loaded once per step.
//   sweep     U x = y in 16-row blocks from the last one: x_K = W_K pend_K, then every pending row above subtracts U(:, K) x_K — one barrier
//             per block; the columns of U come back from memory (L2), requested one block ahead.
//   outputs   step_p / delta_p (zero on the decoupled rows of the leading constant control points), x_b / delta_b, the two sums of the
//             model cost change: what k_band_backward wrote (kernels_factor.hpp), so that the update kernels do not change.
// Bring-up: tests/emul/factor_harness.cpp variant 6 (the kernel source on the CPU against numpy), then tests/test_gpu_edge_cases.py.
#pragma once
#include "kernels_factor.hpp"

namespace hs {

constexpr int kDxThreads = 512;   // eight waves, two per SIMD
constexpr int kDxTiles = 16;      // N <= 256
constexpr int kDxLd = 272;        // LDS row stride of a panel row (doubles)
constexpr int kDxColsPerWave = 47;  // lanes 17 .. 63 of a panel wave: one column each (lanes 0 .. 15: diagonal tile, 16: right-hand side)
// LDS (doubles): panel rows, double buffered | W_k rows (16 x 256) | g | y | x | pend_K (2 x 16) | block sums
constexpr int kDxOffW = 2 * 16 * kDxLd, kDxOffG = kDxOffW + 16 * 256, kDxOffY = kDxOffG + kDxLd, kDxOffX = kDxOffY + kDxLd, kDxOffP = kDxOffX + kDxLd,
              kDxOffRed = kDxOffP + 32, kDxLdsDoubles = kDxOffRed + 32;
typedef double dx_f64x4 __attribute__((vector_size(32)));

/// Systems the kernel holds: n_free free block rows + nb border unknowns within 16 tiles of 16.
__host__ __device__ constexpr bool dense_mx_fits(int n_free, int nb) { return n_free >= 1 && 6 * n_free + nb <= 16 * kDxTiles; }

/// Tiles of wave (A, B): (I, J) = (A + 2 iq, col(jq)), I <= J < 16, with the tile columns dealt to the four classes B in a zigzag —
/// {B, 7 - B, 8 + B, 15 - B} — so that every wave holds 18 or 16 tiles (J mod 4 = B: 12 .. 20); slot of a tile in the wave's accumulator array.
template <int A, int B>
struct DxTiles {
  static constexpr int col(int jq) { return 8 * (jq >> 1) + ((jq & 1) ? 7 - B : B); }
  static constexpr int rows_of(int jq) { return (col(jq) - A) >= 0 ? (col(jq) - A) / 2 + 1 : 0; }  // tiles of tile column jq
  // (no recursion: a recursive constexpr function is not inlined on the device, and a run-time call here turns the accumulator array into scratch memory)
  static constexpr int first(int jq) { return (jq > 0 ? rows_of(0) : 0) + (jq > 1 ? rows_of(1) : 0) + (jq > 2 ? rows_of(2) : 0) + (jq > 3 ? rows_of(3) : 0); }
  static constexpr int count = first(4);
  static constexpr int slot(int iq, int jq) { return first(jq) + iq; }
};

HSD double dx_readlane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
HSD double dx_rsqrt(double d) {  // 1 / sqrt(d): hardware estimate + one Newton-Halley step (as in the 6 x 6 panels and k_border_solve_reg)
  const double dd = d > 0.0 ? d : 1.0, y0 = __builtin_amdgcn_rsq(dd), e = fma(-dd * y0, y0, 1.0);
  return fma(y0 * e, fma(0.375, e, 0.5), y0);
}

/// Where the scaled, damped system lives (copies of the Tables fields: selecting between the pointers of the kernel argument itself made
/// the compiler copy the whole 2.3 KB structure to scratch memory).
struct DxSys {
  const double *Sb, *Spb, *Sbb;
  int ncb, nb, f0, n_pose, n_dense;
};
/// Entry (i, j), i <= j, of the padded dense system: pose rows 6 f0 .. (band storage), border columns, identity on the padding. Branch free (one
/// load from a selected address): the tile loads of a wave are 72 of these, unrolled, and a version with a branch per region was 60 000 lines of ISA.
HSD double dx_entry(const DxSys& Y, int i, int j) {
  const double *sb = Y.Sb, *spb = Y.Spb, *sbb = Y.Sbb;  // (values, not fields: a select between FIELDS becomes an indexed load from a stack copy of Y)
  const int ncb = Y.ncb, nb = Y.nb, f0 = Y.f0, n_pose = Y.n_pose, n_dense = Y.n_dense;
  const int ri = 6 * f0 + i, c = 6 * f0 + j - 6 * (ri / 6), jb = j - n_pose, ib = i - n_pose;
  const bool pose_col = j < n_pose, pose_row = i < n_pose, inside = j < n_dense && (!pose_col || c < ncb);
  const size_t idx_pp = size_t(ri) * ncb + c, idx_pb = size_t(ri) * nb + jb, idx_bb = size_t(ib) * nb + jb;
  const double* src = pose_col ? sb + idx_pp : (pose_row ? spb + idx_pb : sbb + idx_bb);
  const double v = *(inside ? src : sb);
  return inside ? v : ((j >= n_dense && i == j) ? 1.0 : 0.0);
}

constexpr int kDxMaxTilesPerWave = 18;

/// The three phases that depend on which tiles a wave owns (compile time: the accumulators are registers). Everything else — panel, sweep,
/// outputs — is the same code for every wave and is written once in the kernel below.
template <int A, int B>
struct DxWave {
  using TL = DxTiles<A, B>;
  static_assert(TL::count <= kDxMaxTilesPerWave, "accumulator array too short");
  /// load: the wave's tiles from the scaled, damped system (a diagonal tile whole: the entries below its diagonal are the mirror images).
  /// What depends on the row of an entry only is formed once per row, what depends on its column once per column.
  static HSD void load(const DxSys& Y, int nt, int i16, int g4, dx_f64x4 (&acc)[kDxMaxTilesPerWave]) {
    int cj[4];
    bool pose_col[4], in_dense[4];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      const int j = 16 * TL::col(jq) + i16;
      cj[jq] = 6 * Y.f0 + j, pose_col[jq] = j < Y.n_pose, in_dense[jq] = j < Y.n_dense;
    }
#pragma unroll
    for (int iq = 0; iq < 8; ++iq) {
      const int I = A + 2 * iq;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + 4 * r + g4, ri = 6 * Y.f0 + i, six = 6 * (ri / 6);
        const bool pose_row = i < Y.n_pose;
        const double* row_pp = Y.Sb + (size_t(ri) * Y.ncb - six);                                                     // + cj: S[ri][.] in the band row
        const double* row_b = (pose_row ? Y.Spb + size_t(ri) * Y.nb : Y.Sbb + size_t(i - Y.n_pose) * Y.nb) - (6 * Y.f0 + Y.n_pose);  // + cj: border column
#pragma unroll
        for (int jq = 0; jq < 4; ++jq) {
          if (iq >= TL::rows_of(jq)) continue;
          const int J = TL::col(jq), j = 16 * J + i16;
          double v;
          if (I == J) {
            v = dx_entry(Y, i > j ? j : i, i > j ? i : j);
          } else {
            const bool inside = in_dense[jq] && (!pose_col[jq] || cj[jq] - six < Y.ncb);
            const double* src = pose_col[jq] ? row_pp : row_b;
            const double got = src[inside ? cj[jq] : 6 * Y.f0 + Y.n_pose];  // (outside: any address inside the tables)
            v = inside ? got : 0.0;  // (off-diagonal tile: the padding's diagonal is not in it)
          }
          acc[TL::slot(iq, jq)][r] = J < nt ? v : 0.0;
        }
      }
    }
  }
  /// (1) tile row k -> LDS
  static HSD void extract(int k, int nt, int i16, int g4, double* xb, const dx_f64x4 (&acc)[kDxMaxTilesPerWave]) {
#pragma unroll
    for (int iq = 0; iq < 8; ++iq) {
      if (A + 2 * iq != k) continue;
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        if (iq >= TL::rows_of(jq)) continue;
        const int J = TL::col(jq);
        if (J >= nt) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) xb[(4 * r + g4) * kDxLd + 16 * J + i16] = acc[TL::slot(iq, jq)][r];
      }
    }
  }
  /// (3) trailing update: tile (I, J) -= X(k, I)' X(k, J), k < I <= J
  static HSD void update(int k, int nt, int i16, int g4, const double* xb, dx_f64x4 (&acc)[kDxMaxTilesPerWave]) {
    double bop[4][4];
#pragma unroll
    for (int jq = 0; jq < 4; ++jq) {
      const int J = TL::col(jq);
#pragma unroll
      for (int s = 0; s < 4; ++s) bop[jq][s] = (J > k && J < nt) ? xb[(4 * s + g4) * kDxLd + 16 * J + i16] : 0.0;
    }
#pragma unroll
    for (int iq = 0; iq < 8; ++iq) {
      const int I = A + 2 * iq;
      if (I <= k || I >= nt) continue;
      double aop[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) aop[s] = -xb[(4 * s + g4) * kDxLd + 16 * I + i16];
#pragma unroll
      for (int jq = 0; jq < 4; ++jq) {
        if (iq >= TL::rows_of(jq)) continue;
        if (TL::col(jq) >= nt) continue;
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[TL::slot(iq, jq)] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[s], bop[jq][s], acc[TL::slot(iq, jq)], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // (the operand loads of the eight tile rows hoisted to the top of the phase cost 64 registers next to 144 of tiles)
    }
  }
};

/// wave (a, b): SIMD w mod 4 holds (0, b) and (1, 3 - b)
#define HS_DX_DISPATCH(wave, CALL)      \
  switch (wave) {                       \
    case 0: DxWave<0, 0>::CALL; break;  \
    case 1: DxWave<0, 1>::CALL; break;  \
    case 2: DxWave<0, 2>::CALL; break;  \
    case 3: DxWave<0, 3>::CALL; break;  \
    case 4: DxWave<1, 3>::CALL; break;  \
    case 5: DxWave<1, 2>::CALL; break;  \
    case 6: DxWave<1, 1>::CALL; break;  \
    default: DxWave<1, 0>::CALL; break; \
  }

/// One workgroup. f0: leading block rows of constant control points (decoupled, solution zero: the chain starts behind them).
/// ut: 256 x 256 doubles of scratch (the factor by columns, for the sweep).
__global__ void __launch_bounds__(kDxThreads) k_dense_solve_mx(Tables T, int f0, double* ut) {
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, i16 = l & 15, g4 = l >> 4;
  const int n_pose = T.np - 6 * f0, n_dense = n_pose + T.nb, nt = (n_dense + 15) / 16, n_pad = 16 * nt;
  double* gv = smem + kDxOffG;
  double* yv = smem + kDxOffY;
  double* xv = smem + kDxOffX;
  double* wk = smem + kDxOffW;
  dx_f64x4 acc[kDxMaxTilesPerWave];
  {
    const DxSys Y{T.Sb, T.Spb, T.Sbb, 6 * T.bw, T.nb, f0, n_pose, n_dense};
    HS_DX_DISPATCH(w, load(Y, nt, i16, g4, acc))
  }
  for (int c = tid; c < kDxLd; c += kDxThreads) gv[c] = c < n_pose ? T.g_s[6 * f0 + c] : (c < n_dense ? T.gb_s[c - n_pose] : 0.0);
  bool fail = false;
  for (int k = 0; k < nt; ++k) {
    double* xb = smem + (k & 1) * 16 * kDxLd;
    HS_DX_DISPATCH(w, extract(k, nt, i16, g4, xb, acc))
    lds_barrier();
    // ---- (2) panel ----
    const int n_tr = n_pad - 16 * (k + 1);  // trailing columns of the row
    if (w * kDxColsPerWave < n_tr + 16) {    // (waves without a column of their own skip the panel: their SIMDs are the other waves')
      const int idx = w * kDxColsPerWave + (l - 17);
      const bool trailing = l >= 17 && idx < n_tr, inverse = l >= 17 && idx >= n_tr && idx < n_tr + 16;
      const int col = l < 16 ? 16 * k + l : 16 * (k + 1) + (trailing ? idx : 0);
      double a[16];
      {
        const double* src = l == 16 ? gv + 16 * k : xb + col;  // (idle and identity lanes read a column of the buffer they do not use)
        const int stride = l == 16 ? 1 : kDxLd;
        const bool keep = l <= 16 || trailing;
        const int unit = inverse ? idx - n_tr : -1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const double v = src[r * stride];
          a[r] = keep ? v : (r == unit ? 1.0 : 0.0);
        }
      }
#pragma unroll
      for (int p = 0; p < 16; ++p) {
        const double d = dx_readlane(a[p], p);
        fail |= !(d > 0.0);
        a[p] *= dx_rsqrt(d);
#pragma unroll
        for (int r = p + 1; r < 16; ++r) a[r] = fma(-dx_readlane(a[p], r), a[p], a[r]);
      }
      // right-hand side of the trailing columns, X(k, :) as the operand of the update and as factor rows, W_k, y_k
      double gc = trailing ? gv[col] : 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) gc = fma(-a[r], dx_readlane(a[r], 16), gc);
      if (trailing) {
        gv[col] = gc;
#pragma unroll
        for (int r = 0; r < 16; ++r) xb[r * kDxLd + col] = a[r];
        double* dst = ut + size_t(col) * n_pad + 16 * k;  // column `col` of U, rows of block k: 16 contiguous doubles
#pragma unroll
        for (int r = 0; r < 16; r += 2) *reinterpret_cast<double2*>(dst + r) = make_double2(a[r], a[r + 1]);
      }
      if (inverse) {
#pragma unroll
        for (int r = 0; r < 16; ++r) wk[(16 * k + (idx - n_tr)) * 16 + r] = a[r];
      }
      if (w == 0 && l == 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) yv[16 * k + r] = a[r];
      }
    }
    lds_barrier();
    HS_DX_DISPATCH(w, update(k, nt, i16, g4, xb, acc))
  }
  // ---- backward sweep U x = y, lane = row (the first four waves hold the rows; everybody keeps the barriers) ----
  wait_vmem();  // this wave's factor rows have left
  __threadfence();
  lds_barrier();
  double* pk = smem + kDxOffP;
  const int rho = tid;  // row of the padded system (tid < 256)
  double pend = rho < n_pad ? yv[rho] : 0.0;
  double un[16];
  auto fetch = [&](int K) {  // column block K of U for this row
#pragma unroll
    for (int c = 0; c < 16; ++c) un[c] = (K >= 0 && rho < 16 * K) ? ut[size_t(16 * K + c) * n_pad + rho] : 0.0;
  };
  fetch(nt - 1);
  for (int K = nt - 1; K >= 0; --K) {
    if (rho >= 16 * K && rho < 16 * K + 16) pk[(K & 1) * 16 + (rho - 16 * K)] = pend;
    double u[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) u[c] = un[c];
    fetch(K - 1);
    lds_barrier();
    // x_K = W_K pend_K in lanes 0 .. 15 of every wave (W_K upper triangular: the entries left of the diagonal came out as exact zeros)
    double xr = 0.0;
#pragma unroll
    for (int c = 0; c < 16; ++c) xr = fma(wk[(16 * K + i16) * 16 + c], pk[(K & 1) * 16 + c], xr);
    if (w == 0 && l < 16) xv[16 * K + l] = xr;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int c = 0; c < 16; c += 2) s0 = fma(u[c], dx_readlane(xr, c), s0), s1 = fma(u[c + 1], dx_readlane(xr, c + 1), s1);
    pend -= s0 + s1;
  }
  lds_barrier();
  // ---- outputs: step = -x, delta = scale o step, the two sums of the model cost change (k_band_backward's epilogue) ----
  double gd = 0.0, dd = 0.0;
  for (int r = tid; r < 6 * f0; r += kDxThreads) T.step_p[r] = 0.0, T.delta_p[r] = 0.0;
  if (rho < n_dense) {
    const double step = -xv[rho];
    if (rho < n_pose) {
      const int r = 6 * f0 + rho;
      T.step_p[r] = step, T.delta_p[r] = T.scale_p[r] * step;
      gd = T.g_full[r] * step, dd = T.D2p[r] * step * step;
    } else {
      const int b = rho - n_pose;
      T.xb[b] = -step, T.delta_b[b] = T.scale_b[b] * step;
      gd = T.gb_s[b] * step, dd = T.D2b[b] * step * step;
    }
  }
  double* red = smem + kDxOffRed;
  gd = block_sum(gd, red);
  dd = block_sum(dd, red + 8);
  if (tid == 0) {
    st->g_dot_step_pose = gd, st->d2_step2_pose = dd;
    if (fail) st->chol_failed = 1;  // (consumed and cleared by decide_step, kernels_update.hpp)
  }
}

#undef HS_DX_DISPATCH

}  // namespace hs
