// kernels_factor_mfma.hpp — block-banded Cholesky with the trailing window in f64 MFMA accumulator tiles (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include <climits>
#include <utility>

#include "kernels_factor.hpp"

namespace hs {

// ---------------------------------------------------------------------------------------------------------------------
// S = U'U for the block-banded reduced system (6 x 6 blocks, bw band blocks), fused forward solve — same inputs / outputs as
// k_band_factor_la (Ub, U_ii^-1, y = U^-T g; optional two-ended operation), but the rank-6 trailing update runs on the matrix
// cores: v_mfma_f64_16x16x4_f64 has the FMA rate of the vector ALU on gfx950 (64 clk per 16x16x4, tools/microbench/mfma_probe),
// what it removes is the operand traffic — two ds_read_b64 per lane feed 1024 FMAs, where the register-tile VALU update needed 72
// LDS operands per 216 FMAs and was LDS-issue bound (2200 clk per tile, p2_probe).
//
// Data layout. The trailing window is the symmetric W x W diagonal window (W = 16 NT >= 6 bw) of the partially eliminated
// matrix, kept in RING coordinates: matrix index rho lives at ring position rho mod W, so the window slides by six positions per
// block row without moving data. The ring is cut into NT x NT tiles of 16 x 16; only tiles I <= J are stored (the entry of the
// unordered pair {rho, sigma}); tile (I, J) is the accumulator of one wave, C/D layout of the instruction (lane l, register r:
// row (l >> 4) + 4 r, column l & 15). An extra tile column J = NT carries the right-hand side in its column 0, so the forward
// solve rides on the same instruction stream (X value of that column = y_i).
//   step i :  C(I, J) -= X_i[:, I]' X_i[:, J]   two MFMAs per tile (k = 0..3, k = 4..5 + two zero rows); X_i = row i of the factor
//             in ring layout (zeros at every position outside its trailing band), read from LDS: 2 (NT + 1) ds_read_b64 per lane.
// Block rows ENTER the window as column strips (all pairs (rho, sigma) with sigma in the entering block and rho resident): they
// are exactly the rows of the LOWER band, i.e. the rows of the upper band of the reversed system, which k_finalize_reduced
// writes anyway for the two-ended factorisation — each job reads the other job's array. Block row e enters during step
// e - bw - 1; the window keeps two spare block rows (W >= 6 bw + 12), so the ring positions it takes belong to a block row that
// was eliminated before (never to a row that is still needed: the junction re-reads rows m, m + 1 from the accumulators).
//
// Waves:  0 .. NC-1  compute (tiles dealt round robin; update, enter, extract the row two steps ahead into LDS)
//         NC         panel   (row i + 1: -= X_i,1' X_i, 6 x 6 Cholesky redundantly in registers, column solves -> X_(i+1)), as in
//                            k_band_factor_la: one lane per band column, branch free
//         NC + 1     loader  (lower-band rows of the entering block rows HBM -> LDS stage, two steps ahead)
//         NC + 2     storer  (factor row, y, U_ii^-1 -> HBM)
// One LDS-only barrier per block row. Two-ended operation (grid = 2) as in k_band_factor_la: job 1 eliminates the far end of the
// reversed system and hands over the Schur correction of the middle block rows (it loads zeros for the middle-middle pairs, so its
// accumulators hold the pure correction); job 0 adds it at the junction and restarts its pipeline at the first middle row.
// ---------------------------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));

/// Compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N - 1>) — tile indices must be constants, a
/// run-time index into the operand / accumulator arrays would put them into scratch memory.
template <class F, int... Is>
HSD void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
HSD void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <int W>
HSD int ring_rel(int x, int p) {  // (x - p) mod W for x, p in [0, W)
  const int d = x - p;
  return d < 0 ? d + W : d;
}
template <int W>
HSD int ring_add(int p, int c) {  // (p + c) mod W for p in [0, W), c in [0, W]
  const int x = p + c;
  return x >= W ? x - W : x;
}

/// Tile q of the enumeration: first the NT (NT + 1) / 2 tiles I <= J of the matrix window (row major), then the right-hand-side column.
template <int NT>
__host__ __device__ constexpr int mfma_tile_I(int q) {
  constexpr int NTRI = NT * (NT + 1) / 2;
  if (q >= NTRI) return q - NTRI;
  int I = 0;
  while (q >= NT - I) q -= NT - I, ++I;
  return I;
}
template <int NT>
__host__ __device__ constexpr int mfma_tile_J(int q) {
  constexpr int NTRI = NT * (NT + 1) / 2;
  if (q >= NTRI) return NT;
  int I = 0;
  while (q >= NT - I) q -= NT - I, ++I;
  return I + q;
}

template <int NT>
struct MfmaGeom {
  static constexpr int W = 16 * NT;
  static constexpr int LDX = W + ((W % 32 == 16) ? 0 : 16);  // row stride of X in LDS: rows k and k + 1 fall into different bank halves
  static constexpr int LDR = W + 2;                           // row stride of the ring-ordered hand-over buffers [ring columns | rhs | pad]
  static constexpr int NTILE = NT * (NT + 1) / 2 + NT;
  static constexpr int PC = (W + 1 + 63) / 64;                // panel columns per lane
  // LDS (doubles): rowbuf 2 x 6 x LDR | xring 2 x 8 x LDX | stage 2 x 6 x LDR | ublk 2 x 36 | ycol 2 x 8 | dscr 36 | dinv 2 x 6 | wbuf 24
  static constexpr int kRow = 0, kX = kRow + 12 * LDR, kStage = kX + 16 * LDX, kU = kStage + 12 * LDR, kY = kU + 72, kD = kY + 16, kInv = kD + 36,
                       kWb = kInv + 12, kB = kWb + 24, kTotal = kB + 72;  // kB: 2 x 36, block 1 of X (the columns of the next block row)
};

/// S(rho, sigma) of the job's system for a pair that enters the window (own order), with the job's zero rules.
HSD double mfma_job_value(const MfmaJob& J, int np, int ncb, int rho, int sigma) {
  if (rho > sigma) {
    const int t = rho;
    rho = sigma, sigma = t;
  }
  const int e = sigma / 6, off = 6 * e + 5 - rho;  // band offset in the lower row of sigma
  const bool ok = rho >= 0 && off < ncb && e < J.enter_limit && !(rho / 6 >= J.zero_from && e >= J.zero_from);
  const double v = J.L[ok ? size_t(np - 1 - sigma) * ncb + off : 0];
  return ok ? v : 0.0;
}

template <int NT, int NC, int WV>
HSD void mfma_compute_wave(const Tables& T, const MfmaJob& J, double* smem, int l) {
  using G = MfmaGeom<NT>;
  constexpr int W = G::W, LDX = G::LDX, LDR = G::LDR, TW = (G::NTILE - WV + NC - 1) / NC;
  const int bw = T.bw, ncb = 6 * bw, np = T.np, n_steps = J.n_steps, m_at = J.merge_at;
  double* rowbuf = smem + G::kRow;
  const double* xring = smem + G::kX;
  const double* stage = smem + G::kStage;
  const double* ycol = smem + G::kY;
  const int l15 = l & 15, g4 = l >> 4;
  f64x4 acc[TW];

  // ---- initial window: block rows 0 .. bw - 1 at ring position = matrix index ----
  static_for<TW>([&](auto mc) {
    constexpr int m = decltype(mc)::value, q = WV + NC * m, I = mfma_tile_I<NT>(q), Jt = mfma_tile_J<NT>(q);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int a = 16 * I + g4 + 4 * rr;
      double v = 0.0;
      if (Jt < NT) {
        const int b = 16 * Jt + l15;
        if (a < ncb && b < ncb) v = mfma_job_value(J, np, ncb, a, b);
      } else if (l15 == 0 && a < ncb) {
        v = (a / 6 < J.enter_limit && a / 6 < J.zero_from) ? J.g[a] : 0.0;
      }
      acc[m][rr] = v;
    }
  });
  // Both LDS hand-over buffers (rowbuf: row block -> panel, stage: loader -> entering block row) are indexed by RING column (the
  // right-hand side at column W), so that an accumulator element only needs its row test: no band arithmetic per element.
  auto block_row = [&](int a, int p) {  // index of ring row a inside the block row at ring position p (>= 6: not in it)
    const int d = a - p;
    return d < 0 ? d + W : d;
  };
  // dispatch on a wave-uniform tile index: one jump instead of a test per tile
  auto for_tile_index = [&](int t, auto&& f) {
    switch (t) {
#define HS_TILE_CASE(k)                                   \
  case k:                                                 \
    if constexpr (NT > k) f(std::integral_constant<int, k>{}); \
    break;
      HS_TILE_CASE(0) HS_TILE_CASE(1) HS_TILE_CASE(2) HS_TILE_CASE(3) HS_TILE_CASE(4) HS_TILE_CASE(5) HS_TILE_CASE(6) HS_TILE_CASE(7)
      HS_TILE_CASE(8) HS_TILE_CASE(9) HS_TILE_CASE(10) HS_TILE_CASE(11) HS_TILE_CASE(12) HS_TILE_CASE(13) HS_TILE_CASE(14) HS_TILE_CASE(15)
#undef HS_TILE_CASE
      default: break;
    }
  };
  // MOVE = 0: accumulators -> buf (row block r handed to the panel); MOVE = 1: buf -> accumulators (block row entering).
  auto move_rows = [&](auto ic, auto mv, double* buf, int p) {  // this wave's tiles in tile row IC: rows of the block x all columns
    constexpr int IC = decltype(ic)::value;
    int da[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) da[rr] = block_row(16 * IC + g4 + 4 * rr, p);
    static_for<TW>([&](auto mc) {
      constexpr int m = decltype(mc)::value, q = WV + NC * m, I = mfma_tile_I<NT>(q), Jt = mfma_tile_J<NT>(q);
      if constexpr (I == IC) {
        const int col = Jt < NT ? 16 * Jt + l15 : W;
        const bool col_ok = Jt < NT || l15 == 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          if (da[rr] < 6 && col_ok) {
            if constexpr (decltype(mv)::value == 0)
              buf[da[rr] * LDR + col] = acc[m][rr];
            else
              acc[m][rr] = buf[da[rr] * LDR + col];
          }
        }
      }
    });
  };
  // this wave's tiles (I, JC): columns of the block x all rows. The diagonal tile (JC, JC) keeps both orders of a pair, so it is part
  // of this rule too (rows of OTHER block rows that share the tile x columns of the block).
  auto move_cols = [&](auto jc, auto mv, double* buf, int p) {
    constexpr int JC = decltype(jc)::value;
    const int db = block_row(16 * JC + l15, p);
    static_for<TW>([&](auto mc) {
      constexpr int m = decltype(mc)::value, q = WV + NC * m, I = mfma_tile_I<NT>(q), Jt = mfma_tile_J<NT>(q);
      if constexpr (Jt == JC) {
        if (db < 6) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            if constexpr (decltype(mv)::value == 0)
              buf[db * LDR + 16 * I + g4 + 4 * rr] = acc[m][rr];
            else
              acc[m][rr] = buf[db * LDR + 16 * I + g4 + 4 * rr];
          }
        }
      }
    });
  };
  auto move_block = [&](auto mv, double* buf, int p) {
    const int T1 = p >> 4, T2 = ring_add<W>(p, 5) >> 4;
    for_tile_index(T1, [&](auto ic) { move_rows(ic, mv, buf, p), move_cols(ic, mv, buf, p); });
    if (T2 != T1) for_tile_index(T2, [&](auto ic) { move_rows(ic, mv, buf, p), move_cols(ic, mv, buf, p); });
  };
  auto extract = [&](int r, int pr) { move_block(std::integral_constant<int, 0>{}, rowbuf + (r & 1) * 6 * LDR, pr); };
  // Block row e enters from stage[e & 1]: stage[q][ring position of rho] = S(rho, sigma = 6 e + q) over the band of sigma, [q][W] = g.
  // (Ring columns outside that band carry stale values into out-of-band pairs of the window: never read, and overwritten when the
  // block row at that position enters.)
  auto enter = [&](int e, int pe) { move_block(std::integral_constant<int, 1>{}, const_cast<double*>(stage) + (e & 1) * 6 * LDR, pe); };
  auto update = [&](int i) {
    const double* x = xring + (i & 1) * 8 * LDX;
    double xf[NT + 1][2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 2; ++g) xf[t][g] = x[(g4 + 4 * g) * LDX + 16 * t + l15];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const double y = ycol[(i & 1) * 8 + g4 + 4 * g];
      xf[NT][g] = l15 == 0 ? y : 0.0;
    }
    static_for<TW>([&](auto mc) {
      constexpr int m = decltype(mc)::value, q = WV + NC * m, I = mfma_tile_I<NT>(q), Jt = mfma_tile_J<NT>(q);
#pragma unroll
      for (int g = 0; g < 2; ++g) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(-xf[I][g], xf[Jt][g], acc[m], 0, 0, 0);
    });
  };

  // ---- prologue ----
  extract(0, 0);
  lds_barrier();  // B_a: stage holds block row bw (loader)
  {
    int pe = 6 * bw;
    pe = pe >= W ? pe - W : pe;
    enter(bw, pe);
  }
  extract(1, 6);
  lds_barrier();  // B_b: rows 0, 1 handed over
  lds_barrier();  // B_c: X_0 published
  int p_i = 0;    // ring position of block row i
  const bool prof = (T.debug_flags & 16) && l == 0 && WV == 0 && blockIdx.x == 0;  // HS_DEBUG_FLAGS: phase timestamps -> hs_debug_read
  long long* tlog = reinterpret_cast<long long*>(T.xpart);
  for (int i = 0; i < n_steps; ++i) {
    if (prof) tlog[8 * i + 0] = wall_clock64();
    update(i);
    if (prof) tlog[8 * i + 1] = wall_clock64();
    {
      const int pe = ring_add<W>(p_i, (6 * (bw + 1)) % W);
      enter(i + bw + 1, pe);
    }
    if (prof) tlog[8 * i + 2] = wall_clock64();
    extract(i + 2, ring_add<W>(p_i, 12));
    if (prof) tlog[8 * i + 3] = wall_clock64();
    p_i = ring_add<W>(p_i, 6);
    lds_barrier();  // B_i
    if (m_at >= 0 && i + 1 == m_at) {  // ---- junction: add the other end's Schur correction of the middle block rows, restart at row m ----
      while (__hip_atomic_load(T.join_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < T.join_epoch) __builtin_amdgcn_s_sleep(8);
      const int dm = 6 * (bw - 1);
      const double* D = J.win;
      static_for<TW>([&](auto mc) {
        constexpr int m = decltype(mc)::value, q = WV + NC * m, I = mfma_tile_I<NT>(q), Jt = mfma_tile_J<NT>(q);
        const int db = Jt < NT ? ring_rel<W>(16 * Jt + l15, p_i) : dm;
        const bool col_ok = Jt < NT ? db < dm : l15 == 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int da = ring_rel<W>(16 * I + g4 + 4 * rr, p_i);
          if (da < dm && col_ok) acc[m][rr] += D[size_t(da) * (dm + 1) + db];
        }
      });
      extract(m_at, p_i);
      extract(m_at + 1, ring_add<W>(p_i, 6));
      lds_barrier();  // B_j1: rows m, m + 1 handed over again (now with the correction, updated through X_(m-1))
      lds_barrier();  // B_j2: X_m published
    }
  }
  lds_barrier();  // B_end
  if (J.dump) {  // job 1: the pure correction of the middle block rows, in job 0's coordinates (index reversal inside the middle block)
    const int dm = 6 * (bw - 1);
    double* D = J.win;
    static_for<TW>([&](auto mc) {
      constexpr int m = decltype(mc)::value, q = WV + NC * m, I = mfma_tile_I<NT>(q), Jt = mfma_tile_J<NT>(q);
      const int db = Jt < NT ? ring_rel<W>(16 * Jt + l15, p_i) : dm;
      const bool col_ok = Jt < NT ? db < dm : l15 == 0;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int da = ring_rel<W>(16 * I + g4 + 4 * rr, p_i);
        if (da < dm && col_ok) {
          const int ra = dm - 1 - da;
          if (Jt < NT) {
            const int rb = dm - 1 - db;
            D[size_t(ra) * (dm + 1) + rb] = acc[m][rr];
            D[size_t(rb) * (dm + 1) + ra] = acc[m][rr];
          } else {
            D[size_t(ra) * (dm + 1) + dm] = acc[m][rr];
          }
        }
      }
    });
    __threadfence();
    lds_barrier();  // B_dump
  }
}

template <int NT, int NC>
__global__ void __launch_bounds__(64 * (NC + 3)) k_band_factor_mfma(Tables T) {
  using G = MfmaGeom<NT>;
  constexpr int W = G::W, LDX = G::LDX, LDR = G::LDR, PC = G::PC;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  DevState* st = T.st;
  if (st->done) return;
  const MfmaJob J = T.mj[blockIdx.x];
  // Hardware wave 3 is the panel wave: waves are placed round robin on the four SIMDs, so with at most seven waves it has SIMD 3 to
  // itself (sharing a SIMD with a compute wave doubled the latency of its dependency chain). Logical roles: compute 0 .. NC - 1, panel
  // NC, loader NC + 1, storer NC + 2.
  const int tid = threadIdx.x, hw = tid >> 6, l = tid & 63;
  const int wave = hw == 3 ? NC : (hw < 3 ? hw : (hw <= NC ? hw - 1 : hw));
  const int bw = T.bw, ncb = 6 * bw, np = T.np, n_blk = np / 6, n_steps = J.n_steps, m_at = J.merge_at;
  double* rowbuf = smem + G::kRow;
  double* xring = smem + G::kX;
  double* stage = smem + G::kStage;
  double* ublk = smem + G::kU;
  double* ycol = smem + G::kY;
  double* dscr = smem + G::kD;
  double* dinv = smem + G::kInv;
  double* wbuf = smem + G::kWb;
  double* bblk = smem + G::kB;
  __shared__ int fail;
  if (tid == 0) fail = 0;
  // rows 6, 7 of both X buffers and entries 6, 7 of y stay zero (k = 4 .. 7 of the second MFMA)
  for (int e = tid; e < 2 * 2 * LDX; e += blockDim.x) xring[(e / (2 * LDX)) * 8 * LDX + 6 * LDX + e % (2 * LDX)] = 0.0;
  if (tid < 4) ycol[(tid >> 1) * 8 + 6 + (tid & 1)] = 0.0;
  __syncthreads();
  auto junction_wait = [&]() {
    while (__hip_atomic_load(T.join_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < T.join_epoch) __builtin_amdgcn_s_sleep(8);
  };

  if (wave < NC) {  // ================================ compute waves ================================
    if constexpr (NC >= 1) if (wave == 0) mfma_compute_wave<NT, NC, 0>(T, J, smem, l);
    if constexpr (NC >= 2) if (wave == 1) mfma_compute_wave<NT, NC, 1>(T, J, smem, l);
    if constexpr (NC >= 3) if (wave == 2) mfma_compute_wave<NT, NC, 2>(T, J, smem, l);
    if constexpr (NC >= 4) if (wave == 3) mfma_compute_wave<NT, NC, 3>(T, J, smem, l);
    if constexpr (NC >= 5) if (wave == 4) mfma_compute_wave<NT, NC, 4>(T, J, smem, l);
    if constexpr (NC >= 6) if (wave == 5) mfma_compute_wave<NT, NC, 5>(T, J, smem, l);
    if constexpr (NC >= 7) if (wave == 6) mfma_compute_wave<NT, NC, 6>(T, J, smem, l);
    if constexpr (NC >= 8) if (wave == 7) mfma_compute_wave<NT, NC, 7>(T, J, smem, l);
    return;
  }

  if (wave == NC + 1) {  // ================================ loader ================================
    // lane l owns columns t = l + 64 m of a staged block row (t < ncb: band, t == ncb: right-hand side), all six rows
    double va[PC][6], vb[PC][6];
    auto fetch = [&](double (*v)[6], int e) {  // block row e: S(rho = 6 (e - bw + 1) + t, sigma = 6 e + q) with the job's zero rules
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const int t = l + 64 * m;
        const int rho = 6 * (e - bw + 1) + t;
        const bool band = t < ncb, rhs = t == ncb;
        const bool ok_e = e < J.enter_limit;
        const bool ok = band && ok_e && rho >= 0 && !(rho / 6 >= J.zero_from && e >= J.zero_from);
        const bool ok_g = rhs && ok_e && e < J.zero_from;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int sigma = ok_e ? 6 * e + q : 0;
          const double* src = ok ? J.L + size_t(np - 1 - sigma) * ncb + (ncb - 1 - t) : (ok_g ? J.g + sigma : J.L);
          const double x = *src;
          v[m][q] = (ok || ok_g) ? x : 0.0;
        }
      }
    };
    int p_lo = 6;  // ring position of the first band column of the block row being staged (block row bw first: rho_lo = 6)
    auto put = [&](const double (*v)[6], int e) {  // blocks are staged in order: p_lo advances by six per call
      double* dst = stage + (e & 1) * 6 * LDR;
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const int t = l + 64 * m;
        if (t <= ncb) {
          const int col = t < ncb ? ring_add<W>(p_lo, t) : W;
#pragma unroll
          for (int q = 0; q < 6; ++q) dst[q * LDR + col] = v[m][q];
        }
      }
      p_lo = ring_add<W>(p_lo, 6);
    };
    auto junction_io = [&](int i_done) {
      if (m_at >= 0 && i_done + 1 == m_at) {
        junction_wait();
        lds_barrier();  // B_j1
        lds_barrier();  // B_j2
      }
    };
    fetch(va, bw), fetch(vb, bw + 1);
    put(va, bw);
    fetch(va, bw + 2);
    lds_barrier();  // B_a
    put(vb, bw + 1);
    fetch(vb, bw + 3);
    lds_barrier();  // B_b
    lds_barrier();  // B_c
    for (int i = 0; i < n_steps; i += 2) {
      put(va, i + bw + 2);
      fetch(va, i + bw + 4);
      if ((T.debug_flags & 16) && l == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(T.xpart)[8 * i + 7] = wall_clock64();
      lds_barrier();  // B_i
      junction_io(i);
      if (i + 1 < n_steps) {
        put(vb, i + bw + 3);
        fetch(vb, i + bw + 5);
        lds_barrier();
        junction_io(i + 1);
      }
    }
    lds_barrier();  // B_end
    if (J.dump) lds_barrier();
    return;
  }

  if (wave == NC + 2) {  // ================================ storer ================================
    lds_barrier();  // B_a
    lds_barrier();  // B_b
    lds_barrier();  // B_c: X_0 complete
    int p_i = 0;
    for (int i = 0; i < n_steps; ++i) {
      const double* x = xring + (i & 1) * 8 * LDX;
      const double* ub = ublk + (i & 1) * 36;
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        const int c = l + 64 * m;
        if (c < ncb) {
          const int pos = ring_add<W>(p_i, c);
          double* dst = J.Ub + size_t(6 * i) * ncb + c;
#pragma unroll
          for (int a = 0; a < 6; ++a) dst[a * ncb] = c < 6 ? ub[a * 6 + c] : x[a * LDX + pos];
        }
      }
      if (l < 6) J.ybuf[6 * i + l] = ycol[(i & 1) * 8 + l];
      {  // W = U_ii^-1 (upper triangular, packed): lane c < 6 solves U w = e_c; 1 / u_aa from the panel wave
        const double* di = dinv + (i & 1) * 6;
        const int c = l < 6 ? l : 0;
        double w[6];
#pragma unroll
        for (int a = 5; a >= 0; --a) {
          double t = a == c ? 1.0 : 0.0;
#pragma unroll
          for (int k = a + 1; k < 6; ++k) t = fma(-ub[a * 6 + k], w[k], t);
          w[a] = t * di[a];
        }
        if (l < 6) {
#pragma unroll
          for (int a = 0; a < 6; ++a)
            if (a <= c) J.Ubk[size_t(i) * 24 + (a * 6 - a * (a - 1) / 2 + (c - a))] = w[a];
        }
      }
      p_i = ring_add<W>(p_i, 6);
      if ((T.debug_flags & 16) && l == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(T.xpart)[8 * i + 6] = wall_clock64();
      lds_barrier();  // B_i
      if (m_at >= 0 && i + 1 == m_at) {
        junction_wait();
        lds_barrier();  // B_j1
        lds_barrier();  // B_j2
      }
    }
    lds_barrier();  // B_end
    if (l == 0 && fail) st->chol_failed = 1;
    if (J.dump) {
      lds_barrier();  // B_dump: the compute waves have written the window
      if (l == 0) {
        __threadfence();
        __hip_atomic_store(T.join_flag, T.join_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    (void)wbuf;
    return;
  }

  // ================================ panel wave ================================
#define UIDX(a, c) ((a) * 6 - (a) * ((a)-1) / 2 + ((c) - (a)))
  // Column c = l + 64 m of the pivot row in band order (c < ncb), ring position (p_r + c) mod W; the last slot of lane 63 carries the
  // right-hand side. Slots with ncb <= c < W only zero their ring position (X is zero outside the trailing band).
  const bool pp = (T.debug_flags & 16) && l == 0 && blockIdx.x == 0;
  long long* plog = reinterpret_cast<long long*>(T.xpart) + 8 * 1024;
  auto panel = [&](int r, int pr, bool update) {
    if (pp) plog[8 * r + 0] = wall_clock64();
    const double* row = rowbuf + (r & 1) * 6 * LDR;
    const double* xp = xring + ((r - 1) & 1) * 8 * LDX;
    const double* yp = ycol + ((r - 1) & 1) * 8;
    double* xo = xring + (r & 1) * 8 * LDX;
    double v[PC][6];
    int pos[PC];
    bool is_rhs[PC], is_band[PC];
#pragma unroll
    for (int m = 0; m < PC; ++m) {
      const int c = l + 64 * m;
      is_rhs[m] = m == PC - 1 && l == 63;
      is_band[m] = c < ncb && !is_rhs[m];
      pos[m] = c < W ? ring_add<W>(pr, c) : 0;
      const int cr = is_rhs[m] ? W : pos[m];  // (slots outside the band read a stale value that is never used)
#pragma unroll
      for (int a = 0; a < 6; ++a) v[m][a] = row[a * LDR + cr];
    }
    if (update) {
      double B[6][6];  // block 1 of X_(r-1): the columns of block row r (kept contiguous by the previous panel: 18 broadcast reads)
      const double* bp = bblk + ((r - 1) & 1) * 36;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int a = 0; a < 6; a += 2) {
          const double2 t = *reinterpret_cast<const double2*>(bp + 6 * k + a);
          B[k][a] = t.x, B[k][a + 1] = t.y;
        }
#pragma unroll
      for (int m = 0; m < PC; ++m) {
        double xc[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const double t = is_rhs[m] ? yp[k] : xp[k * LDX + pos[m]];
          xc[k] = (is_band[m] || is_rhs[m]) ? t : 0.0;
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int k = 0; k < 6; ++k) v[m][a] = fma(-B[k][a], xc[k], v[m][a]);
      }
    }
    if (pp) plog[8 * r + 1] = wall_clock64();
    if (l < 6) {
#pragma unroll
      for (int a = 0; a < 6; ++a) dscr[6 * a + l] = v[0][a];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave: LDS is in order, only the compiler must not reorder
    double U[21], inv[6], dmin;
    {
      int pidx = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = a; c < 6; ++c) U[pidx++] = dscr[6 * a + c];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      double d = U[UIDX(a, a)];
#pragma unroll
      for (int k = 0; k < a; ++k) d = fma(-U[UIDX(k, a)], U[UIDX(k, a)], d);
      dmin = a == 0 ? d : fmin(dmin, d);
      const double y = __builtin_amdgcn_rsq(d);
      const double e = fma(-d * y, y, 1.0);
      const double rs = fma(y * e, fma(0.375, e, 0.5), y);
      inv[a] = rs;
      const double nrs = -rs;  // off-diagonal entries are kept negated (products of two of them are unchanged)
#pragma unroll
      for (int c = a + 1; c < 6; ++c) {
        double t = U[UIDX(a, c)];
#pragma unroll
        for (int k = 0; k < a; ++k) t = fma(-U[UIDX(k, a)], U[UIDX(k, c)], t);
        U[UIDX(a, c)] = t * nrs;
      }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) dinv[(r & 1) * 6 + a] = inv[a];  // every lane, same value
    if (!(dmin > 0.0) && l == 0) fail = 1;
    if (pp) plog[8 * r + 2] = wall_clock64();
#pragma unroll
    for (int m = 0; m < PC; ++m) {
      double x[6];
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        double t = v[m][a];
#pragma unroll
        for (int k = 0; k < a; ++k) t = fma(U[UIDX(k, a)], x[k], t);  // U holds -u_ka
        x[a] = t * inv[a];
      }
      const int c = l + 64 * m;
      if (is_rhs[m]) {
#pragma unroll
        for (int a = 0; a < 6; ++a) ycol[(r & 1) * 8 + a] = x[a];
      } else if (c < W) {
        const bool live = c >= 6 && c < ncb;
#pragma unroll
        for (int a = 0; a < 6; ++a) xo[a * LDX + pos[m]] = live ? x[a] : 0.0;
        if (c < 6) {  // column c of U_rr (upper part; the rest of the column is never read)
#pragma unroll
          for (int a = 0; a < 6; ++a) ublk[(r & 1) * 36 + a * 6 + c] = x[a];
        } else if (c < 12) {  // block 1: what the next panel subtracts from its row
#pragma unroll
          for (int a = 0; a < 6; ++a) bblk[(r & 1) * 36 + a * 6 + (c - 6)] = x[a];
        }
      }
    }
    if (pp) plog[8 * r + 3] = wall_clock64();
  };
  lds_barrier();  // B_a
  lds_barrier();  // B_b: rows 0, 1 in rowbuf
  panel(0, 0, false);
  lds_barrier();  // B_c
  int p_next = 6;  // ring position of block row i + 1
  for (int i = 0; i < n_steps; ++i) {
    const bool junction = m_at >= 0 && i + 1 == m_at;  // no look-ahead across the junction: row m changes there
    const bool pprof = (T.debug_flags & 16) && l == 0 && blockIdx.x == 0;
    if (pprof) reinterpret_cast<long long*>(T.xpart)[8 * i + 4] = wall_clock64();
    if (i + 1 < n_steps && !junction) panel(i + 1, p_next, true);
    if (pprof) reinterpret_cast<long long*>(T.xpart)[8 * i + 5] = wall_clock64();
    lds_barrier();  // B_i
    if (junction) {
      junction_wait();
      lds_barrier();  // B_j1
      panel(m_at, p_next, false);
      lds_barrier();  // B_j2
    }
    p_next = ring_add<W>(p_next, 6);
  }
  lds_barrier();  // B_end
  if (J.dump) lds_barrier();
#undef UIDX
}

}  // namespace hs
