// kernels_schur.hpp — landmark elimination and the owner-computes reduced system: k_landmark, k_seg_gram, k_group_gram, k_assemble, packing / finalisation (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include "kernels_common.hpp"

namespace hs {

/// Second half of the landmark pass: given the wave-reduced H_ll, b_l and
/// this lane's W rows, forms V = S_l H_ll S_l + D_l^2 = L L', stores L, y-hat, the scaled gradient and the Y-hat rows.
template <int PS>
HSD void landmark_finish(const Tables& T, int dl, int lane, bool active, bool fresh, double radius, const double* sl_old, int yoff, int rows,
                         const double* h, const double* b, double (*w)[3], bool lead = true) {
  double sl[3];
  if (fresh) {
    sl[0] = 1.0 / (1.0 + sqrt(h[0])), sl[1] = 1.0 / (1.0 + sqrt(h[3])), sl[2] = 1.0 / (1.0 + sqrt(h[5]));
    if (lane < 3 && lead) T.lm_scale[3 * dl + lane] = sl[lane];
  } else {
    sl[0] = sl_old[0], sl[1] = sl_old[1], sl[2] = sl_old[2];
  }
  // V = S H S + clamp(diag)/radius
  double v00 = sl[0] * sl[0] * h[0], v01 = sl[0] * sl[1] * h[1], v02 = sl[0] * sl[2] * h[2];
  double v11 = sl[1] * sl[1] * h[3], v12 = sl[1] * sl[2] * h[4], v22 = sl[2] * sl[2] * h[5];
  const double inv_radius = 1.0 / radius;
  const double d0 = fmin(fmax(v00, 1e-6), 1e32) * inv_radius, d1 = fmin(fmax(v11, 1e-6), 1e32) * inv_radius, d2 = fmin(fmax(v22, 1e-6), 1e32) * inv_radius;
  v00 += d0, v11 += d1, v22 += d2;
  // Cholesky V = L L' with reciprocal pivots: every lane runs this redundantly, and a double-precision divide or square root
  // costs ~35 instructions, so the 3x3 factor and the row solves below use 1 / l_ii from the hardware rsq estimate + one
  // third-order correction (error ~ e^3, full double accuracy) and multiply.
  auto rsqrt_refined = [](double d) {
    const double y = __builtin_amdgcn_rsq(d);
    const double e = fma(-d * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
  };
  const double i00 = rsqrt_refined(v00), l00 = v00 * i00, l10 = v01 * i00, l20 = v02 * i00;
  const double p11 = v11 - l10 * l10, i11 = rsqrt_refined(p11), l11 = p11 * i11, l21 = (v12 - l20 * l10) * i11;
  const double p22 = v22 - l20 * l20 - l21 * l21, i22 = rsqrt_refined(p22), l22 = p22 * i22;
  const double sb0 = sl[0] * b[0], sb1 = sl[1] * b[1], sb2 = sl[2] * b[2];
  const double y0 = sb0 * i00, y1 = (sb1 - l10 * y0) * i11, y2 = (sb2 - l20 * y0 - l21 * y1) * i22;
  if (lane == 0 && lead) {  // (lead = false: a wave that only owns further rows of W, k_landmark_rows)
    double* L = T.lm_L + 6 * dl;
    L[0] = l00, L[1] = l10, L[2] = l11, L[3] = l20, L[4] = l21, L[5] = l22;
    T.lm_yhat[3 * dl] = active ? y0 : 0.0, T.lm_yhat[3 * dl + 1] = active ? y1 : 0.0, T.lm_yhat[3 * dl + 2] = active ? y2 : 0.0;
    T.lm_sb[3 * dl] = sb0, T.lm_sb[3 * dl + 1] = sb1, T.lm_sb[3 * dl + 2] = sb2;
    T.lm_D2[3 * dl] = d0, T.lm_D2[3 * dl + 1] = d1, T.lm_D2[3 * dl + 2] = d2;
    // gradient max norm: per-landmark value, max-reduced by k_pack_exchange (thousands of atomics on one word would
    // serialise at ~12 ns each and dominate this pass)
    T.lm_gmax[dl] = active ? fmax(fabs(b[0]), fmax(fabs(b[1]), fabs(b[2]))) : 0.0;
  }
  // W rows -> Y-hat rows
  double* Y = T.Y + yoff;
#pragma unroll
  for (int ps = 0; ps < PS; ++ps) {
    const int rho = lane + 64 * ps;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (rho < rows) {
      const double w0 = w[ps][0] * sl[0], w1 = w[ps][1] * sl[1], w2 = w[ps][2] * sl[2];
      // y L' = w  (forward substitution on the columns of L')
      a0 = w0 * i00, a1 = (w1 - a0 * l10) * i11, a2 = (w2 - a0 * l20 - a1 * l21) * i22;
      if (!active) a0 = a1 = a2 = 0.0;
      Y[3 * rho] = a0, Y[3 * rho + 1] = a1, Y[3 * rho + 2] = a2;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Landmark pass: one wave per landmark.  H_ll = sum Jl'Jl, b_l = sum Jl'r, W_l = sum Jp'Jl over the landmark's
// residuals; V = S_l H_ll S_l + D_l^2 = L L';  Y-hat = W S_l L^-T (pose-side row scaling is applied by the consumer),
// y-hat = L^-1 S_l b_l.  Jacobi scaling S_l is fixed at iteration 0 (TrustRegionMinimizer, jacobi_scaling = true).
// PS = 64-row passes a lane owns (rows of W = 6 * control points the landmark touches <= 64 * PS).
// ---------------------------------------------------------------------------------------------------------------------
template <int K, int PS, int U>
HSD void landmark_eliminate(const Tables& T, int dl, int lane) {
  constexpr int REC = 8 + 12 * K;
  const int q0 = T.lm_ptr[dl], q1 = T.lm_ptr[dl + 1];
  const int c_first = T.lm_cfirst[dl], rows = 6 * T.lm_ncp[dl];
  // operands of the finishing step: requested up front, they do not depend on the records
  const bool fresh = !T.st->scaling_ready;
  const double radius = T.st->radius;
  const bool is_const = T.lm_const[dl];
  const int yoff = T.lm_yoff[dl];
  double sl_old[3] = {1.0, 1.0, 1.0};
  if (!fresh) sl_old[0] = T.lm_scale[3 * dl], sl_old[1] = T.lm_scale[3 * dl + 1], sl_old[2] = T.lm_scale[3 * dl + 2];
  // One pass over the landmark's residuals: lane q of a 64-chunk fetches (first control point, record slot) of residual q
  // once; the chunk is then walked with register broadcasts, every lane accumulating its own W row(s) (rho = lane, lane + 64)
  // and lane q the H_ll / b_l terms of residual q. All loads are unconditional on clamped indices and masked afterwards:
  // straight-line code, so the loads of U records (all 64-row passes) are in flight together instead of one round trip per
  // record and pass. (The kernel is nevertheless bound by instruction issue, not by this chain: U = 1, 2, 4 and the branchy
  // original all take 17.3 us at 5 000 landmarks x 10 records — only 6K of 64 lanes carry a W row of a given record.)
  double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  double w[PS][3];
#pragma unroll
  for (int ps = 0; ps < PS; ++ps) w[ps][0] = w[ps][1] = w[ps][2] = 0.0;
  for (int base = q0; base < q1; base += 64) {
    const int myq = min(base + lane, q1 - 1);
    const bool mine = base + lane < q1;
    const int my_first = T.v_first[myq], my_pos = T.v_pos[myq];
    const double* myrec = current_visual_records(T) + size_t(my_pos) * REC;
    double own[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) own[e] = myrec[e];
    const int cnt = min(64, q1 - base);
    for (int t0 = 0; t0 < cnt; t0 += U) {
      double ja[U][PS], jb[U][PS], jl[U][6];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = min(t0 + u, cnt - 1);
        const int ft = __builtin_amdgcn_readlane(my_first, t), pt = __builtin_amdgcn_readlane(my_pos, t);  // wave-uniform
        const double* rec = current_visual_records(T) + size_t(pt) * REC;
        const int off = 6 * (ft - c_first);
#pragma unroll
        for (int e = 0; e < 6; ++e) jl[u][e] = rec[2 + e];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
          const int c = lane + 64 * ps - off;
          const bool ok = t0 + u < cnt && c >= 0 && c < 6 * K && lane + 64 * ps < rows;
          const int cc = ok ? c : 0;
          const double va = rec[8 + cc], vb = rec[8 + 6 * K + cc];
          ja[u][ps] = ok ? va : 0.0, jb[u][ps] = ok ? vb : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {  // masked slots add exact zeros
          w[ps][0] = fma(ja[u][ps], jl[u][0], fma(jb[u][ps], jl[u][3], w[ps][0]));
          w[ps][1] = fma(ja[u][ps], jl[u][1], fma(jb[u][ps], jl[u][4], w[ps][1]));
          w[ps][2] = fma(ja[u][ps], jl[u][2], fma(jb[u][ps], jl[u][5], w[ps][2]));
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double rr = mine ? own[r] : 0.0, j0 = mine ? own[2 + 3 * r] : 0.0, j1 = mine ? own[3 + 3 * r] : 0.0, j2 = mine ? own[4 + 3 * r] : 0.0;
      h[0] = fma(j0, j0, h[0]), h[1] = fma(j0, j1, h[1]), h[2] = fma(j0, j2, h[2]);
      h[3] = fma(j1, j1, h[3]), h[4] = fma(j1, j2, h[4]), h[5] = fma(j2, j2, h[5]);
      b[0] = fma(j0, rr, b[0]), b[1] = fma(j1, rr, b[1]), b[2] = fma(j2, rr, b[2]);
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) h[i] = wave_sum(h[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) b[i] = wave_sum(b[i]);
  landmark_finish<PS>(T, dl, lane, (q1 > q0) && !is_const, fresh, radius, sl_old, yoff, rows, h, b, w);
}

template <int K, int PS, int U>
__global__ void __launch_bounds__(kBlock) k_landmark(Tables T) {
  if (T.st->done) return;
  const int dl = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (dl >= T.n_lm) return;
  landmark_eliminate<K, PS, U>(T, dl, threadIdx.x & 63);
}

/// Long feature tracks (sliding-window replay: a landmark is seen for up to 3 s, ~120 records, and couples up to ~33 control points):
/// one WORKGROUP per landmark, wave p owns rows [64 p, 64 p + 64) of W. A record touches 6 K consecutive rows, i.e. one pass (two when
/// it straddles a boundary): each wave walks only the records that touch its rows (ballot over the chunk), where k_landmark<K,4,1>
/// walks every record in all four passes on one wave — the longest track sets the kernel time (60 us in the replay). The
/// accumulation order per row is the same as there (records in table order), so the results are bit-identical. H_ll and b_l are
/// accumulated by every wave (15 FMAs per 64 records); wave 0 stores the per-landmark outputs.
template <int K, int U>
__global__ void __launch_bounds__(kBlock) k_landmark_rows(Tables T) {
  constexpr int REC = 8 + 12 * K;
  if (T.st->done) return;
  const int dl = blockIdx.x, lane = threadIdx.x & 63, pass = threadIdx.x >> 6;
  const int q0 = T.lm_ptr[dl], q1 = T.lm_ptr[dl + 1];
  const int c_first = T.lm_cfirst[dl], rows = 6 * T.lm_ncp[dl];
  const int r0 = 64 * pass;
  if (pass > 0 && r0 >= rows) return;
  const bool fresh = !T.st->scaling_ready;
  const double radius = T.st->radius;
  const bool is_const = T.lm_const[dl];
  const int yoff = T.lm_yoff[dl];
  double sl_old[3] = {1.0, 1.0, 1.0};
  if (!fresh) sl_old[0] = T.lm_scale[3 * dl], sl_old[1] = T.lm_scale[3 * dl + 1], sl_old[2] = T.lm_scale[3 * dl + 2];
  double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
  double w[1][3] = {{0.0, 0.0, 0.0}};
  const int rho = r0 + lane;
  for (int base = q0; base < q1; base += 64) {
    const int myq = min(base + lane, q1 - 1);
    const bool mine = base + lane < q1;
    const int my_first = T.v_first[myq], my_pos = T.v_pos[myq];
    const double* myrec = current_visual_records(T) + size_t(my_pos) * REC;
    double own[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) own[e] = myrec[e];
    const int my_off = 6 * (my_first - c_first);
    unsigned long long todo = __ballot(mine && my_off < r0 + 64 && my_off + 6 * K > r0);  // records of this chunk that touch the wave's rows
    while (todo) {
      double ja[U], jb[U], jl[U][6];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool have = todo != 0;
        const int t = have ? __builtin_ctzll(todo) : 0;
        todo = have ? todo & (todo - 1) : 0;
        const int ft = __builtin_amdgcn_readlane(my_first, t), pt = __builtin_amdgcn_readlane(my_pos, t);  // wave-uniform
        const double* rec = current_visual_records(T) + size_t(pt) * REC;
#pragma unroll
        for (int e = 0; e < 6; ++e) jl[u][e] = rec[2 + e];
        const int c = rho - 6 * (ft - c_first);
        const bool ok = have && c >= 0 && c < 6 * K && rho < rows;
        const int cc = ok ? c : 0;
        const double va = rec[8 + cc], vb = rec[8 + 6 * K + cc];
        ja[u] = ok ? va : 0.0, jb[u] = ok ? vb : 0.0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {  // masked slots add exact zeros
        w[0][0] = fma(ja[u], jl[u][0], fma(jb[u], jl[u][3], w[0][0]));
        w[0][1] = fma(ja[u], jl[u][1], fma(jb[u], jl[u][4], w[0][1]));
        w[0][2] = fma(ja[u], jl[u][2], fma(jb[u], jl[u][5], w[0][2]));
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const double rr = mine ? own[r] : 0.0, j0 = mine ? own[2 + 3 * r] : 0.0, j1 = mine ? own[3 + 3 * r] : 0.0, j2 = mine ? own[4 + 3 * r] : 0.0;
      h[0] = fma(j0, j0, h[0]), h[1] = fma(j0, j1, h[1]), h[2] = fma(j0, j2, h[2]);
      h[3] = fma(j1, j1, h[3]), h[4] = fma(j1, j2, h[4]), h[5] = fma(j2, j2, h[5]);
      b[0] = fma(j0, rr, b[0]), b[1] = fma(j1, rr, b[1]), b[2] = fma(j2, rr, b[2]);
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) h[i] = wave_sum(h[i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) b[i] = wave_sum(b[i]);
  landmark_finish<1>(T, dl, lane, (q1 > q0) && !is_const, fresh, radius, sl_old, yoff + 3 * r0, rows - r0, h, b, w, pass == 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Reduced system  S = Sp (J_p'J_p) Sp + D_p^2 - Sp (sum_l Yh_l Yh_l') Sp,   g = Sp (g_p - sum_l Yh_l yh_l)  (raw, unscaled parts here;
// scaling and damping in k_finalize_reduced). Owner-computes formulation: every record and every Y-hat row is read ONCE.
//   k_seg_gram<K>   : one workgroup per (segment, split): P = sum J_p' J_p (6K x 6K) and J_p' r over the segment's records
//   k_group_gram<NT>: one workgroup per (first control point c, split): Q = - sum_l Yh_l Yh_l' over the landmarks whose
//                     track starts at c (6 bw x 6 bw window, upper 6x6 tiles), q = - sum_l Yh_l yh_l
//   k_assemble<K>   : block row i = sum of the <= K segment partials and <= bw group partials that overlap it, in a fixed
//                     order (bit-reproducible, no floating-point atomics), written straight into the exchange buffer
// (The first version gathered per block row and re-read each record K times and each Y-hat row once per covered control point.)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSegStage = 6144;  // doubles of record data staged in LDS per round (48 KB)

/// (body shared by k_seg_gram and the combined launch k_gram_pair; bid = workgroup index in the segment work list)
template <int K>
HSD void seg_gram_body(const Tables& T, const int bid) {
  HS_DYNAMIC_LDS(stage);  // kSegStage doubles: a contiguous run of records
  __shared__ __attribute__((aligned(16))) double red[kBlock * 12 + kBlock * 3];
  if (T.st->done) return;
  // 3 x TC register tiles, NS record streams. TC = 4 where it divides 6 K (orders 4 and 6), 2 for order 5 (6 K = 30)
  constexpr int NCA = 6 * K, TC = NCA % 4 == 0 ? 4 : 2, RG = NCA / 3, CG = NCA / TC, TPS = RG * CG, NS = kBlock / TPS;
  static_assert(NCA % 3 == 0 && NCA % TC == 0 && TPS <= kBlock, "tile shape of the segment Gram kernel");
  constexpr int VREC = 8 + 12 * K, PREC = 6 + 36 * K;
  // work list: workgroup w serves segment sw_seg[w] as split sp of nsp (splits proportional to the segment's record count: the
  // first and last segment of a window collect the clamped stamps)
  const int first = T.sw_seg[bid], sp = bid - T.sw_ptr[first], nsp = T.sw_ptr[first + 1] - T.sw_ptr[first];
  const int tid = threadIdx.x;
  const int stream = tid / TPS, tb = tid % TPS, rg = tb / CG, cg = tb % CG;
  const bool sprof = prof_enabled(T.debug_flags, 32) && tid == 0 && sp == 0 && first < 128;
  long long* slog = reinterpret_cast<long long*>(T.xpart) + 8 * 1024 + 8 * first;
  if (sprof) slog[0] = wall_clock64();
  // a tile is needed if some column block >= the row block (upper block triangle); column group 0 also carries J'r
  const bool live = stream < NS && (cg == 0 || (TC * cg + TC - 1) / 6 >= (3 * rg) / 6);
  double acc[3][4], gacc[3] = {0, 0, 0};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[r][c] = 0.0;
  // Records of a segment are contiguous (segment-major), split `sp` takes a contiguous share: one coalesced sweep brings a run
  // of records into LDS (a single HBM round trip instead of one per record), the streams then walk it from LDS.
  auto run = [&](const double* recs, int r0, int r1, int REC, int n_rows, int joff) {
    const int n = r1 - r0, lo = r0 + int((long long)n * sp / nsp), hi = r0 + int((long long)n * (sp + 1) / nsp);
    const int per = kSegStage / REC;
    for (int c0 = lo; c0 < hi; c0 += per) {
      const int cnt = min(per, hi - c0);
      __syncthreads();
      const double2* src = reinterpret_cast<const double2*>(recs + size_t(c0) * REC);
      const int n2 = cnt * REC / 2;
      for (int e0 = tid; e0 < n2; e0 += 8 * kBlock) {  // eight independent 16-byte loads in flight per lane
        double2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kBlock;
          v[u] = e < n2 ? src[e] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kBlock;
          if (e < n2) reinterpret_cast<double2*>(stage)[e] = v[u];
        }
      }
      __syncthreads();
      if (sprof) slog[1] = wall_clock64();
      if (live)
        for (int c = stream; c < cnt; c += NS) {
          const double* rec = stage + c * REC;
#pragma unroll 2
          for (int r = 0; r < n_rows; ++r) {
            const double* j = rec + joff + r * NCA;
            const double a0 = j[3 * rg], a1 = j[3 * rg + 1], a2 = j[3 * rg + 2];
            const double2 b01 = *reinterpret_cast<const double2*>(j + TC * cg);
            acc[0][0] = fma(a0, b01.x, acc[0][0]), acc[0][1] = fma(a0, b01.y, acc[0][1]);
            acc[1][0] = fma(a1, b01.x, acc[1][0]), acc[1][1] = fma(a1, b01.y, acc[1][1]);
            acc[2][0] = fma(a2, b01.x, acc[2][0]), acc[2][1] = fma(a2, b01.y, acc[2][1]);
            if (TC == 4) {
              const double2 b23 = *reinterpret_cast<const double2*>(j + TC * cg + 2);
              acc[0][2] = fma(a0, b23.x, acc[0][2]), acc[0][3] = fma(a0, b23.y, acc[0][3]);
              acc[1][2] = fma(a1, b23.x, acc[1][2]), acc[1][3] = fma(a1, b23.y, acc[1][3]);
              acc[2][2] = fma(a2, b23.x, acc[2][2]), acc[2][3] = fma(a2, b23.y, acc[2][3]);
            }
            if (cg == 0) {
              const double rr = rec[r];
              gacc[0] = fma(a0, rr, gacc[0]), gacc[1] = fma(a1, rr, gacc[1]), gacc[2] = fma(a2, rr, gacc[2]);
            }
          }
        }
    }
  };
  if (!T.fused) run(current_visual_records(T), T.v_seg_ptr[first], T.v_seg_ptr[first + 1], VREC, 2, 8);  // (fused build: J_p'J_p of the visual factors comes with the chunk partials)
  if (T.n_pri) run(T.p_rec, T.p_seg_ptr[first], T.p_seg_ptr[first + 1], PREC, 6, 6);
  if (T.n_ine) run(T.i_rec, T.i_seg_ptr[first], T.i_seg_ptr[first + 1], 18 + 36 * K + 2 * T.kb, 6, 6);
  if (sprof) slog[2] = wall_clock64();
  // combine the record streams in index order
  double* racc = red;                 // [stream][TPS][12]
  double* rg3 = red + kBlock * 12;    // [stream][RG][3]
  if (stream < NS) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) racc[(stream * TPS + tb) * 12 + 4 * r + c] = acc[r][c];
    if (cg == 0)
#pragma unroll
      for (int r = 0; r < 3; ++r) rg3[(stream * RG + rg) * 3 + r] = gacc[r];
  }
  __syncthreads();
  double* P = T.segP + size_t(bid) * (NCA * NCA + NCA);
  for (int e = tid; e < NCA * NCA; e += kBlock) {
    const int a = e / NCA, c = e % NCA;
    const int t = (a / 3) * CG + c / TC, in = 4 * (a % 3) + c % TC;
    double v = 0.0;
#pragma unroll
    for (int st = 0; st < NS; ++st) v += racc[(st * TPS + t) * 12 + in];
    P[e] = v;  // tiles below the block diagonal were never accumulated (zeros) and are never read
  }
  if (tid < NCA) {
    double v = 0.0;
#pragma unroll
    for (int st = 0; st < NS; ++st) v += rg3[(st * RG + tid / 3) * 3 + tid % 3];
    P[NCA * NCA + tid] = v;
  }
  if (sprof) slog[3] = wall_clock64();
}

template <int K>
__global__ void __launch_bounds__(kBlock) k_seg_gram(Tables T) {
  seg_gram_body<K>(T, blockIdx.x);
}

HSD int ok_index(int b, int nb) { return b < nb ? b : 0; }

/// Upper 6x6 tiles of the 6 bw x 6 bw window of a landmark group: tile index of (rb, cb), rb <= cb < bw.
HSD int group_tile_index(int rb, int cb, int bw) { return rb * bw - rb * (rb - 1) / 2 + (cb - rb); }

constexpr int kGroupBatch = 16;  // landmarks staged in LDS per round (host caps it so that the stage fits 48 KB)

template <int NT>  // tiles per thread: NT == 1: two landmark streams of 128 lanes (bw <= 15); NT > 1: one stream, bw (bw + 1) / 2 <= NT * kBlock
HSD void group_gram_body(const Tables& T, const int batch, const int bid) {
  HS_DYNAMIC_LDS(smem);
  __shared__ int m_ncp[kBlock], m_off[kBlock];
  if (T.st->done) return;
  // work list: workgroup w serves group cf = gw_cf[w] as split sp of nsp (splits proportional to the group's landmark count:
  // the first control point of a window collects every track that started before it)
  const int cf = T.gw_cf[bid], sp = bid - T.gw_ptr[cf], nsp = T.gw_ptr[cf + 1] - T.gw_ptr[cf];
  const int tid = threadIdx.x;
  const int bw = T.bw, R = 6 * bw, ntile = bw * (bw + 1) / 2;
  double* ybuf = smem;                          // batch x (R x 3): Y-hat rows (zero past the landmark's rows)
  double* yh = smem + size_t(batch) * R * 3;    // batch x 4: y-hat
  const bool two = NT == 1 && ntile <= kBlock / 2;  // two landmark streams
  const int stream = two ? tid / (kBlock / 2) : 0, nstream = two ? 2 : 1;
  const int lt = two ? tid % (kBlock / 2) : tid, lthreads = two ? kBlock / 2 : kBlock;
  int t_rb[NT], t_cb[NT];
  bool t_ok[NT];
  double acc[NT][36], qacc[NT][6];
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    const int t = lt + m * lthreads;
    t_ok[m] = t < ntile;
    int rb = 0, rem = t_ok[m] ? t : 0;
    while (rem >= bw - rb) rem -= bw - rb, ++rb;  // row rb holds bw - rb tiles
    t_rb[m] = rb, t_cb[m] = rb + rem;
#pragma unroll
    for (int e = 0; e < 36; ++e) acc[m][e] = 0.0;
#pragma unroll
    for (int e = 0; e < 6; ++e) qacc[m][e] = 0.0;
  }
  const bool gprof = prof_enabled(T.debug_flags, 32) && tid == 0 && sp == 0 && cf < 128;
  long long* glog = reinterpret_cast<long long*>(T.xpart) + 8 * cf;
  if (gprof) glog[0] = wall_clock64();
  const int dl0 = T.cf_ptr[cf], dl1 = T.cf_ptr[cf + 1];
  const int n_mine = dl1 > dl0 + sp ? (dl1 - dl0 - sp + nsp - 1) / nsp : 0;  // landmarks dl = dl0 + sp + t * nsp
  for (int t0 = 0; t0 < n_mine; t0 += kBlock) {  // (one pass unless a group holds more than 256 landmarks per split)
    __syncthreads();
    if (t0 + tid < n_mine) {
      const int dl = dl0 + sp + (t0 + tid) * nsp;
      m_ncp[tid] = T.lm_ncp[dl], m_off[tid] = T.lm_yoff[dl];
    }
    __syncthreads();
    const int n_pass = min(kBlock, n_mine - t0);
    if (gprof) glog[1] = wall_clock64();
    for (int b0 = 0; b0 < n_pass; b0 += batch) {
      const int nb = min(batch, n_pass - b0);
      __syncthreads();
      for (int e0 = tid; e0 < nb * R * 3; e0 += 8 * kBlock) {  // eight independent loads in flight per lane
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kBlock, b = e / (R * 3), w = e % (R * 3);
          const bool ok = e < nb * R * 3 && w < 18 * m_ncp[b0 + (ok_index(b, nb))];
          v[u] = ok ? T.Y[m_off[b0 + ok_index(b, nb)] + w] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = e0 + u * kBlock;
          if (e < nb * R * 3) ybuf[e] = v[u];
        }
      }
      if (tid < 3 * nb) yh[4 * (tid / 3) + tid % 3] = T.lm_yhat[3 * (dl0 + sp + (t0 + b0 + tid / 3) * nsp) + tid % 3];
      __syncthreads();
      if (gprof) glog[2] = wall_clock64();
      for (int b = stream; b < nb; b += nstream) {
        const int ncp = m_ncp[b0 + b];
        const double* Yb = ybuf + size_t(b) * R * 3;
#pragma unroll
        for (int m = 0; m < NT; ++m) {
          if (!t_ok[m] || t_cb[m] >= ncp) continue;
          double B[18];
#pragma unroll
          for (int e = 0; e < 18; e += 2) {
            const double2 vb = *reinterpret_cast<const double2*>(Yb + 18 * t_cb[m] + e);
            B[e] = vb.x, B[e + 1] = vb.y;
          }
          const bool diag = t_rb[m] == t_cb[m];
          const double y0 = yh[4 * b], y1 = yh[4 * b + 1], y2 = yh[4 * b + 2];
#pragma unroll
          for (int rp = 0; rp < 3; ++rp) {  // two rows of the A operand at a time: 148 instead of 190 registers, three workgroups per CU
            double A[6];
#pragma unroll
            for (int e = 0; e < 6; e += 2) {
              const double2 va = *reinterpret_cast<const double2*>(Yb + 18 * t_rb[m] + 6 * rp + e);
              A[e] = va.x, A[e + 1] = va.y;
            }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const int r = 2 * rp + rr;
#pragma unroll
              for (int c = 0; c < 6; ++c)
                acc[m][6 * r + c] = fma(-A[3 * rr + 2], B[3 * c + 2], fma(-A[3 * rr + 1], B[3 * c + 1], fma(-A[3 * rr], B[3 * c], acc[m][6 * r + c])));
              if (diag) qacc[m][r] = fma(-A[3 * rr + 2], y2, fma(-A[3 * rr + 1], y1, fma(-A[3 * rr], y0, qacc[m][r])));
            }
          }
        }
      }
    }
  }
  if (gprof) glog[3] = wall_clock64(), glog[5] = n_mine;
  double* Q = T.grpQ + size_t(bid) * (size_t(ntile) * 36 + R);
  if (two) {  // stream 1 hands its partial to stream 0 through LDS (fixed order: stream 0 + stream 1)
    __syncthreads();
    double* xch = smem;  // 128 x 42 doubles <= the stage
    if (stream == 1 && t_ok[0]) {
#pragma unroll
      for (int e = 0; e < 36; ++e) xch[lt * 42 + e] = acc[0][e];
#pragma unroll
      for (int e = 0; e < 6; ++e) xch[lt * 42 + 36 + e] = qacc[0][e];
    }
    __syncthreads();
    if (stream == 0 && t_ok[0]) {
#pragma unroll
      for (int e = 0; e < 36; ++e) acc[0][e] += xch[lt * 42 + e];
#pragma unroll
      for (int e = 0; e < 6; ++e) qacc[0][e] += xch[lt * 42 + 36 + e];
    }
    if (stream == 1) return;
  }
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    if (!t_ok[m]) continue;
    const int t = lt + m * lthreads;
#pragma unroll
    for (int e = 0; e < 36; e += 2) *reinterpret_cast<double2*>(Q + size_t(t) * 36 + e) = make_double2(acc[m][e], acc[m][e + 1]);
    if (t_rb[m] == t_cb[m])
#pragma unroll
      for (int r = 0; r < 6; ++r) Q[size_t(ntile) * 36 + 6 * t_rb[m] + r] = qacc[m][r];
  }
  if (gprof) glog[4] = wall_clock64();
}

template <int NT>
__global__ void __launch_bounds__(kBlock, NT == 1 ? 3 : 1) k_group_gram(Tables T, int batch) {
  group_gram_body<NT>(T, batch, blockIdx.x);
}

/// k_group_gram and k_seg_gram in ONE launch (workgroups [0, n_group) serve the landmark groups, the rest the segments): the two are
/// independent and each fills only part of the chip; as two launches they ran on two streams with an event fork / join, whose barrier
/// packets cost more (~6 us each on the critical path, rocprofv3 kernel trace) than the overlap was worth.
template <int K, int NT>
__global__ void __launch_bounds__(kBlock) k_gram_pair(Tables T, int batch, int n_group) {
  if (int(blockIdx.x) < n_group)
    group_gram_body<NT>(T, batch, blockIdx.x);
  else
    seg_gram_body<K>(T, blockIdx.x - n_group);
}

HSD void dense_padding_corner(double* D, int n_dense, int n_pad, int first, int stride);  // (below, with k_finalize_reduced)

constexpr int kAsmThreads = 512, kAsmU = 6;  // lanes per scalar row, loads in flight per lane (6, and at most 80 VGPRs: six waves per SIMD = three workgroups per CU —
                                              // at 97 VGPRs a CU held two, and the 768 workgroups of a 128-control-point window ran in two rounds: 13.7 us)

/// Scalar row rho = 6 i + a of the raw (unscaled, undamped) reduced system from the segment and group partials; writes xbuf
/// directly. Grid (n_cp, 6). The sources of an entry are dealt round-robin to `nsl` thread slices (loads of a slice are
/// issued kAsmU at a time), the slices are combined through LDS in index order: fixed summation order, bit-reproducible.
/// direct = 1 (single shard, no border unknowns, Jacobi scaling fixed — every linearisation of a solve but the first — and a factorisation that
/// does the iteration bookkeeping itself, Tables::bookkeep): the row is scaled, damped and written in the factorisation's layout right here,
///     S = Sp Sraw Sp + D_p^2,  g = Sp (g_p + g_schur),  g_full = Sp g_p,  D_p^2 = clamp(Sp^2 diag(J'J), 1e-6, 1e32) / radius
/// (what k_finalize_reduced does with a launch of its own: ~6 us on the chain of an iteration, most of it launch latency).
/// QB = 0: the kernel of every window whose rows collect a few dozen partials. QB > 0 (k_assemble_wide): window-wide bands on the fused
/// build — a row collects every chunk of the window, ~70 sources per lane: the sources behind the first round are fetched QB at a time, the
/// work-list entries of batch n + 1 together with the values of batch n (one memory round trip per QB sources where entry -> value per
/// batch of kAsmU takes two). Same sources in the same order per lane: the sums are bit-identical.
template <int K, int QB, int THREADS>
HSD void assemble_body(const Tables& T, int direct) {
  __shared__ double part[3][THREADS];
  __shared__ double gpair[2];
  // phase timestamps (profiling builds, HS_DEBUG_FLAGS 64; tools/assemble_phase_timing.py): lane 0 of every workgroup
  const bool aprof = prof_enabled(T.debug_flags, 256) && threadIdx.x == 0;
  long long* alog = reinterpret_cast<long long*>(T.xpart) + 128 * 1024 + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
  if (aprof) alog[0] = wall_clock64();
  if (T.st->done) return;
  if (aprof) alog[1] = wall_clock64();
  constexpr int NCA = 6 * K;
  const int i = blockIdx.x, a = blockIdx.y, tid = threadIdx.x;
  const int bw = T.bw, ncb = 6 * bw, R = 6 * bw, ntile = bw * (bw + 1) / 2;
  const size_t pstride = NCA * NCA + NCA, qstride = size_t(ntile) * 36 + (T.fused ? 3 : 1) * R;
  const int f0 = max(0, i - K + 1), f1 = min(i, T.n_seg - 1);
  const int c0 = max(0, i - bw + 1);
  const int nent = ncb + 2;  // band entries + [J'r | Y-hat y-hat] of this row
  // wide_q: only the 6 K band entries of J_p'J_p, J_p'r and diag J_p'J_p have sources — the lanes are dealt over THOSE entries, so that a row's
  // ~170 chunks make 4 - 5 sources per lane (one round of loads) where the 6 bw + 2 entries of the full row left 33 per lane (three to six rounds)
  const int nent_s = T.wide_q ? 6 * K + 2 : nent;
  const int nsl = max(1, THREADS / nent_s), sl = tid / nent_s, cs = tid % nent_s;
  const int c = T.wide_q ? (cs < 6 * K ? cs : ncb + (cs - 6 * K)) : cs;
  // direct mode: scaling of this row and of the lane's column, trust-region radius (requested with the work lists)
  double d_sr = 1.0, d_sc = 1.0, d_radius = 1.0;
  if (direct) {
    d_sr = T.scale_p[6 * i + a];
    if (tid < ncb && 6 * i + tid < T.np) d_sc = T.scale_p[6 * i + tid];
    d_radius = T.st->radius;
  }
  double q_wide = 0.0;  // wide_q: this lane's entry of the window's landmark term (band entry tid of row 6 i + a, or the row's -Yh yh)
  if (T.wide_q && T.n_lm > 0 && tid < nent && tid != ncb) q_wide = tid < ncb ? (6 * i + tid < T.np ? T.Qw[size_t(6 * i + a) * ncb + tid] : 0.0) : T.Qw[size_t(T.np) * ncb + 6 * i + a];  // (entries right of the matrix: never written)
  double va = 0.0, vb = 0.0;  // J'J part / Schur part
  double vp = 0.0;            // fused build: the chunk partials carry J_p'J_p inside their tiles; J_p'r (lane ncb) and diag J_p'J_p (lane a) ride along
  if (sl < nsl) {
    const int kk = c / 6, cc = c % 6;
    // segment partials: the workgroups of segments f0 .. f1 are contiguous in the work list; landmark-group partials: those of
    // groups c0 .. i. A lane's sources are p = sl, sl + nsl, ... The first kAsmU segment sources and 2 kAsmU group sources are
    // fetched in two rounds (all work-list entries, then all partial values: two memory round trips instead of one pair per
    // batch); the sums run in the same fixed order as a plain loop over p.
    const bool a_live = (c < ncb ? kk < K : c == ncb) && f1 >= f0;
    const bool x_live = T.fused && (c == ncb || c == a);  // lanes that also collect J_p'r / diag J_p'J_p of the chunk partials
    // wide_q (window-wide bands on the fused build): a chunk partial holds the band tiles of J_p'J_p only, the landmark term of the whole
    // window comes from T.Qw (k_landmark_gram_wide) and is added behind the combination of the slices
    const bool b_live = T.n_lm > 0 && (T.wide_q ? (c < ncb && kk < K) || x_live : c < ncb || c == ncb + 1 || x_live);
    const int p_lo = a_live ? T.sw_ptr[f0] : 0, np_ = a_live ? T.sw_ptr[f1 + 1] - p_lo : 0;
    const int q_lo = b_live ? T.gw_ptr[c0] : 0, nq = b_live ? T.gw_ptr[i + 1] - q_lo : 0;
    // (plain macros, not lambdas: a by-reference closure kept these operands in scratch memory)
#define HS_SEG_VALUE(p, seg) \
  ((p) < np_ && (c == ncb || i - (seg) + kk < K) \
       ? T.segP[(p_lo + (p)) * int(pstride) + (c == ncb ? NCA * NCA + 6 * (i - (seg)) + a : (6 * (i - (seg)) + a) * NCA + 6 * (i - (seg)) + c)] \
       : 0.0)
#define HS_GRP_VALUE(q, cf) \
  ((q) < nq && (c > ncb || i - (cf) + kk < bw) \
       ? T.grpQ[(q_lo + (q)) * int(qstride) + \
                (c > ncb ? ntile * 36 + 6 * (i - (cf)) + a                                                               \
                 : T.fused ? group_tile_index(i - (cf), i - (cf), bw) * 36 + a * 6 * (bw - (i - (cf))) + c /* row-major rows: kernels_build.hpp */ \
                           : group_tile_index(i - (cf), i - (cf) + kk, bw) * 36 + 6 * a + cc)] \
       : 0.0)
#define HS_GRP_EXTRA(q, cf) ((q) < nq ? T.grpQ[(q_lo + (q)) * int(qstride) + ntile * 36 + (c == ncb ? 1 : 2) * R + 6 * (i - (cf)) + a] : 0.0)
    if (aprof && np_ + nq >= 0) alog[2] = wall_clock64();  // ranges of the work lists here
    int si[kAsmU], gi[2 * kAsmU];
#pragma unroll
    for (int u = 0; u < kAsmU; ++u) si[u] = sl + u * nsl < np_ ? T.sw_seg[p_lo + sl + u * nsl] : 0;
#pragma unroll
    for (int u = 0; u < 2 * kAsmU; ++u) gi[u] = sl + u * nsl < nq ? T.gw_cf[q_lo + sl + u * nsl] : 0;
    if (aprof && si[0] + gi[0] >= 0) alog[3] = wall_clock64();  // work-list entries here
    double sv[kAsmU], gv[2 * kAsmU];
#pragma unroll
    for (int u = 0; u < kAsmU; ++u) sv[u] = HS_SEG_VALUE(sl + u * nsl, si[u]);
#pragma unroll
    for (int u = 0; u < 2 * kAsmU; ++u) gv[u] = c == ncb && T.fused ? 0.0 : HS_GRP_VALUE(sl + u * nsl, gi[u]);
    if (x_live) {
      double xv[2 * kAsmU];
#pragma unroll
      for (int u = 0; u < 2 * kAsmU; ++u) xv[u] = HS_GRP_EXTRA(sl + u * nsl, gi[u]);
#pragma unroll
      for (int u = 0; u < 2 * kAsmU; ++u) vp += xv[u];
    }
#pragma unroll
    for (int u = 0; u < kAsmU; ++u) va += sv[u];
#pragma unroll
    for (int u = 0; u < 2 * kAsmU; ++u) vb += gv[u];
    // the rest (segments / groups split into unusually many workgroups)
    for (int p0 = sl + kAsmU * nsl; p0 < np_; p0 += kAsmU * nsl) {
      double v[kAsmU];
#pragma unroll
      for (int u = 0; u < kAsmU; ++u) {
        const int p = p0 + u * nsl;
        const int seg = p < np_ ? T.sw_seg[p_lo + p] : 0;
        v[u] = HS_SEG_VALUE(p, seg);
      }
#pragma unroll
      for (int u = 0; u < kAsmU; ++u) va += v[u];
    }
    if constexpr (QB == 0) {
      for (int q0 = sl + 2 * kAsmU * nsl; q0 < nq; q0 += kAsmU * nsl) {
        double v[kAsmU];
#pragma unroll
        for (int u = 0; u < kAsmU; ++u) {
          const int q = q0 + u * nsl;
          const int cf = q < nq ? T.gw_cf[q_lo + q] : 0;
          v[u] = c == ncb && T.fused ? 0.0 : HS_GRP_VALUE(q, cf);
          if (x_live) vp += HS_GRP_EXTRA(q, cf);
        }
#pragma unroll
        for (int u = 0; u < kAsmU; ++u) vb += v[u];
      }
    } else {
      int q0 = sl + 2 * kAsmU * nsl;
      int cfn[QB > 0 ? QB : 1];
#pragma unroll
      for (int u = 0; u < QB; ++u) cfn[u] = q0 + u * nsl < nq ? T.gw_cf[q_lo + q0 + u * nsl] : 0;
      for (; q0 < nq; q0 += QB * nsl) {
        double v[QB > 0 ? QB : 1], xv[QB > 0 ? QB : 1];
        int cfc[QB > 0 ? QB : 1];
#pragma unroll
        for (int u = 0; u < QB; ++u) cfc[u] = cfn[u];
#pragma unroll
        for (int u = 0; u < QB; ++u) v[u] = c == ncb && T.fused ? 0.0 : HS_GRP_VALUE(q0 + u * nsl, cfc[u]);
#pragma unroll
        for (int u = 0; u < QB; ++u) xv[u] = x_live ? HS_GRP_EXTRA(q0 + u * nsl, cfc[u]) : 0.0;
#pragma unroll
        for (int u = 0; u < QB; ++u) {
          const int q = q0 + (QB + u) * nsl;
          cfn[u] = q < nq ? T.gw_cf[q_lo + q] : 0;
        }
#pragma unroll
        for (int u = 0; u < QB; ++u) vb += v[u];
#pragma unroll
        for (int u = 0; u < QB; ++u) vp += xv[u];
      }
    }
#undef HS_SEG_VALUE
#undef HS_GRP_VALUE
#undef HS_GRP_EXTRA
  }
  if (aprof && va + vb + vp != 1e300) alog[4] = wall_clock64();  // values here
  part[0][tid] = va, part[1][tid] = vb, part[2][tid] = vp;
  __syncthreads();
  if (aprof) alog[5] = wall_clock64();
  if (tid < nent) {
    double sa = 0.0, sb = 0.0, sx = 0.0;
    const int es = !T.wide_q ? tid : tid < 6 * K ? tid : tid >= ncb ? 6 * K + (tid - ncb) : -1;  // this entry's lane slot inside a slice
    for (int q = 0; q < nsl && es >= 0; ++q) sa += part[0][q * nent_s + es], sb += part[1][q * nent_s + es], sx += part[2][q * nent_s + es];
    const int rho = 6 * i + a;
    sb += q_wide;
    if (direct) {
      if (tid < ncb) {
        const int sigma = 6 * i + tid;
        double out = 0.0;
        if (sigma < T.np) {
          out = d_sr * d_sc * (sa + sb);
          if (tid == a) {
            const double d = sa + sx;  // diag J'J
            if (d > 0.0) {
              const double d2 = fmin(fmax(d_sr * d_sr * d, 1e-6), 1e32) / d_radius;
              out += d2;
              T.D2p[rho] = d2;
            } else {  // structurally zero column (constant / unobserved): keep the system non-singular, step = 0
              out = 1.0;
              T.D2p[rho] = 0.0;
            }
          }
        }
        T.Sb[size_t(rho) * ncb + tid] = out;
        if (T.dense && i >= T.dense_f0 && sigma < T.np && sigma >= rho) {  // dense copy for k_dense_solve_mx (k_finalize_reduced): entry and mirror image
          const int ii = rho - 6 * T.dense_f0, jj = sigma - 6 * T.dense_f0;
          T.dense[size_t(ii) * kDenseLd + jj] = out, T.dense[size_t(jj) * kDenseLd + ii] = out;
        }
        if (T.Sb2 && sigma < T.np) {  // reversed copy for the far end of the two-ended factorisation (k_finalize_reduced)
          const int rv = T.np - 1 - sigma, cv = T.np - 1 - rho;
          T.Sb2[size_t(rv) * ncb + (cv - 6 * (rv / 6))] = out;
        }
      } else if (tid == ncb) {
        gpair[0] = sa + sx;
      } else {
        gpair[1] = sb;
      }
    } else if (tid < ncb) {
      T.xbuf[size_t(rho) * ncb + tid] = sa + sb;
      if (tid == a) T.xbuf[T.xo_dj + rho] = sa + sx;  // diag J'J: segment partials (priors, inertial; records path: visual too) + chunk partials
    } else if (tid == ncb) {
      T.xbuf[T.xo_g + rho] = sa + sx;
    } else {
      T.xbuf[T.xo_gs + rho] = sb;
    }
  }
  if (direct) {
    __syncthreads();
    if (tid == 0) {
      const int rho = 6 * i + a;
      const double gp = gpair[0], gs = gpair[1];
      T.g_full[rho] = d_sr * gp;
      T.g_s[rho] = d_sr * (gp + gs);
      if (T.Sb2) T.g2[T.np - 1 - rho] = d_sr * (gp + gs);
      T.gabs[rho] = fabs(gp);
      if (T.dense && i >= T.dense_f0) {  // the right-hand side as column (and row) n_dense of the dense copy
        const int ii = rho - 6 * T.dense_f0, n_dense = T.np - 6 * T.dense_f0 + T.nb;
        T.dense[size_t(ii) * kDenseLd + n_dense] = d_sr * (gp + gs), T.dense[size_t(n_dense) * kDenseLd + ii] = d_sr * (gp + gs);
      }
    }
    if (T.dense && i >= T.dense_f0) {  // this row of the dense copy right of the band and in the padding: zero (k_finalize_reduced does six rows at once)
      const int n_pose = T.np - 6 * T.dense_f0, n_dense = n_pose + T.nb, n_pad = 16 * ((n_dense + 1 + 15) / 16);
      const int j0 = min(6 * (i - T.dense_f0) + ncb, n_pose), n_right = n_pose - j0, n_padc = n_pad - (n_dense + 1), ii = 6 * (i - T.dense_f0) + a;
      for (int q = tid; q < n_right + n_padc; q += THREADS) {
        const int jj = q < n_right ? j0 + q : n_dense + 1 + (q - n_right);
        T.dense[size_t(ii) * kDenseLd + jj] = 0.0, T.dense[size_t(jj) * kDenseLd + ii] = 0.0;
      }
      if (T.nb == 0 && i == T.sp.n_cp - 1 && a == 5) dense_padding_corner(T.dense, n_dense, n_pad, tid, THREADS);
    }
  }
  if (aprof) alog[6] = wall_clock64();
}
template <int K>
__global__ void __launch_bounds__(kAsmThreads, 6) k_assemble(Tables T, int direct) {  // (see kAsmThreads)
  assemble_body<K, 0, kAsmThreads>(T, direct);
}
constexpr int kAsmWideThreads = 1024, kAsmWideBatch = 12;  // (sixteen waves: 128 VGPRs)
template <int K>
__global__ void __launch_bounds__(kAsmWideThreads) k_assemble_wide(Tables T, int direct) {  // (few workgroups — a short window — and many sources per row)
  assemble_body<K, kAsmWideBatch, kAsmWideThreads>(T, direct);
}

/// wide_q — the landmark term of the reduced system for window-wide bands, ONCE per window instead of once per chunk:
///   Q(rho, sigma) = - sum_l Yh_l(rho) . Yh_l(sigma),   q(rho) = - sum_l Yh_l(rho) . yh_l      (Yh_l: 6 ncp_l x 3 rows in T.Y, yh_l in T.lm_yhat)
/// On a sliding window's steady state every track is as long as the window: a chunk of k_build_visual holds three landmarks, its Yh Yh' is
/// a rank-9 update of ALL 561 window tiles, and 165 chunks sent 27 MB of such partials through HBM for a 0.3 MB reduced system — 8.7 us of a
/// 33 us chunk to form them, 21 of k_assemble_wide's 24 us to read them back (stamps: tools/build_phase_timing.py r, assemble_phase_timing.py r).
/// The factors are in HBM anyway (k_update_visual's back-substitution reads them): one workgroup per 6 x 6 tile (bi, bj >= bi) of the band,
/// thread s takes the landmarks l = s, s + 128, ... that cover both blocks (table entries of four in one round, then their blocks in pairs),
/// the 128 streams are summed through LDS in a fixed order (bit-reproducible).
/// Output in band-row storage: Qw[(6 bi + r) * 6 bw + 6 (bj - bi) + c]; the diagonal tiles add q behind the matrix.
constexpr int kGramWideThreads = 128;  // (two waves per tile; a 256-thread version with a wave butterfly per sum took 23 us, this one 17)
__global__ void __launch_bounds__(kGramWideThreads) k_landmark_gram_wide(Tables T) {
  // (With the four landmarks of a thread as independent code the compiler kept every load in flight: 504 VGPRs, one wave per SIMD, two
  //  workgroups per CU, the tiles of a full window in two rounds — 17.8 us; capped at two waves per SIMD (256 VGPRs) it spilled: 55 us, at four:
  //  115 us; a compiler barrier behind each landmark's products changed nothing (480). Hence a real loop over the thread's landmarks: the blocks
  //  are requested with the table entries — unconditionally, from rows of T.Yt that may never have been written — and masked afterwards.)
  constexpr int NT = kGramWideThreads, NP = NT / 32;
  __shared__ double red[NT * 21], red2[NP * 21];
  if (T.st->done) return;
  const int tid = threadIdx.x;
  const int bw = T.bw, ncb = 6 * bw, n_cp = T.sp.n_cp;
  int t = int(blockIdx.x), bi = 0;  // block row bi holds min(bw, n_cp - bi) tiles
  while (bi < n_cp && t >= min(bw, n_cp - bi)) t -= min(bw, n_cp - bi), ++bi;
  if (bi >= n_cp) return;
  const int bj = bi + t;
  const bool diag = bi == bj;
  double acc[36], qa[6];
#pragma unroll
  for (int e = 0; e < 36; ++e) acc[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 6; ++e) qa[e] = 0.0;
#pragma unroll 1
  for (int l = tid; l < T.n_obs_lm; l += NT) {  // a real loop, one landmark's blocks live at a time (see the note at the head of the kernel)
    const int cf = T.lm_cfirst[l], ncp = T.lm_ncp[l];
    const double y0 = diag ? T.lm_yhat[3 * l] : 0.0, y1 = diag ? T.lm_yhat[3 * l + 1] : 0.0, y2 = diag ? T.lm_yhat[3 * l + 2] : 0.0;
    // (T.Yt: block row, pair of doubles, landmark — a wave's 64 landmarks are 1 KB of consecutive memory per load)
    const double2* pa = reinterpret_cast<const double2*>(T.Yt) + size_t(bi) * 9 * T.yt_stride + l;
    const double2* pb = reinterpret_cast<const double2*>(T.Yt) + size_t(bj) * 9 * T.yt_stride + l;
    double2 A2[9], B2[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) A2[e] = pa[size_t(e) * T.yt_stride], B2[e] = pb[size_t(e) * T.yt_stride];  // (rows outside the landmark's track: never written, masked below)
    const bool on = cf <= bi && bj < cf + ncp;
    double A[18], B[18];  // (a landmark that does not cover the tile adds zeros: the order of the sums does not depend on the data)
#pragma unroll
    for (int e = 0; e < 9; ++e) A[2 * e] = on ? A2[e].x : 0.0, A[2 * e + 1] = on ? A2[e].y : 0.0, B[2 * e] = on ? B2[e].x : 0.0, B[2 * e + 1] = on ? B2[e].y : 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int c = 0; c < 6; ++c)
        acc[6 * r + c] = fma(-A[3 * r + 2], B[3 * c + 2], fma(-A[3 * r + 1], B[3 * c + 1], fma(-A[3 * r], B[3 * c], acc[6 * r + c])));
      qa[r] = fma(-A[3 * r + 2], y2, fma(-A[3 * r + 1], y1, fma(-A[3 * r], y0, qa[r])));
    }
  }
  // 42 sums over the threads, through LDS in two passes of 21 values: thread (part, v) adds the 32 threads of its part in thread order,
  // 21 threads add the parts in part order (fixed order: bit-reproducible; 42 wave butterflies would be 250 cross-lane steps)
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int v = 0; v < 21; ++v) red[tid * 21 + v] = pass == 0 ? acc[v] : v < 15 ? acc[v < 15 ? 21 + v : 0] : qa[v < 15 ? 0 : v - 15];
    __syncthreads();
    if (tid < NP * 21) {
      const int part = tid / 21, v = tid - 21 * part;
      double sum = 0.0;
      for (int j = 0; j < 32; ++j) sum += red[(32 * part + j) * 21 + v];
      red2[tid] = sum;
    }
    __syncthreads();
    if (tid < 21) {
      double sum = red2[tid];
#pragma unroll
      for (int part = 1; part < NP; ++part) sum += red2[21 * part + tid];
      const int g = 21 * pass + tid;
      if (g < 36)
        T.Qw[size_t(6 * bi + g / 6) * ncb + 6 * (bj - bi) + g % 6] = sum;
      else if (diag)
        T.Qw[size_t(T.np) * ncb + 6 * bi + (g - 36)] = sum;
    }
  }
}
/// Workgroups of k_landmark_gram_wide: one per band tile.
__host__ __device__ inline int landmark_gram_wide_grid(int n_cp, int bw) {
  int n = 0;
  for (int bi = 0; bi < n_cp; ++bi) n += bw < n_cp - bi ? bw : n_cp - bi;
  return n;
}

/// xbuf[e] = sum over the accumulation splits (fixed order => bit-reproducible). The result is additive across residual shards.
__global__ void __launch_bounds__(kBlock) k_reduce_partials(Tables T, int nsp, int e0) {
  if (T.st->done) return;
  const int n = T.xo_bb;  // [Sraw | g_p | g_schur | diag | Hpb]; e0 = xo_pb when the pose part comes from k_assemble
  for (int e = e0 + blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int k0 = 0; k0 < nsp; k0 += 16) {  // (every split in flight, added in split order — finalize_border_body: a plain loop pays a memory round trip per split)
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = k0 + u < nsp ? T.xpart[size_t(k0 + u) * T.x_count1 + e] : 0.0;
#pragma unroll
      for (int u = 0; u < 16; ++u) s += v[u];
    }
    T.xbuf[e] = s;
  }
}

HSD void begin_iteration(const Tables& T, double cost, double gmax, bool set_scaling_ready);
HSD void finalize_border_body(const Tables& T, int wg, int n_wg, int n_splits);  // kernels_border.hpp

/// Local cost and landmark-side gradient max norm into the exchange buffer (slot per rank so that a SUM all-reduce
/// delivers every rank's value to every rank). reduce_here (single shard, no border unknowns): nothing is exchanged, so the
/// iteration bookkeeping of k_cost_reduce is done right here (the pose-side gradient is already in the buffer).
HSD void pack_exchange_body(const Tables& T, int reduce_here) {
  __shared__ double red[kBlock / 64];
  DevState* st = T.st;
  if (st->done) return;
  // Speculative solves linearise at the candidate: from their second iteration on the cost partials of the CURRENT point are not
  // recomputed — this shard's part of its cost is what decide_step took over when the candidate was accepted (st->local_cost). The
  // exchanged sum is then right on every shard, whichever way the shard linearises (a shard that holds the priors does it the plain way).
  const bool kept_cost = (st->spec == 1 || st->spec == 2) && st->iteration > 0;
  double s = kept_cost ? (threadIdx.x == 0 ? st->local_cost : 0.0) : strided_sum(T.cost_part, T.n_cost_part);
  double gm = strided_max<24>(T.lm_gmax, T.n_obs_lm);  // one value per landmark: a single round of loads at 5 000 landmarks
  if (reduce_here) {
    const double* gp = T.xbuf + T.xo_g;
    for (int i0 = threadIdx.x; i0 < T.np; i0 += 8 * blockDim.x) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * blockDim.x;
        v[u] = i < T.np ? fabs(gp[i]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) gm = fmax(gm, v[u]);
    }
    for (int b = threadIdx.x; b < T.nb; b += blockDim.x) gm = fmax(gm, fabs(T.xbuf[T.xo_gb + b]));  // border unknowns (bias points, gravity)
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    T.xbuf[T.xo_cost] = s;
    if (!kept_cost) st->local_cost = s;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gm = fmax(gm, __shfl_xor(gm, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = gm;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < int(blockDim.x >> 6); ++i) gm = fmax(gm, red[i]);
    for (int r = 0; r < T.world; ++r) T.xbuf[T.xo_gmax + r] = (r == T.rank) ? gm : 0.0;
    if (reduce_here) begin_iteration(T, s, gm, /*set_scaling_ready=*/false);  // k_finalize_reduced of this linearisation still needs the flag
  }
}
__global__ void __launch_bounds__(kBlock) k_pack_exchange(Tables T, int reduce_here) { pack_exchange_body(T, reduce_here); }

/// Corner of the dense copy of a small system (Tables::dense) behind its n_dense unknowns: the right-hand side's own diagonal entry — the
/// right-hand side rides through k_dense_solve_mx's factorisation as column n_dense, and any entry that keeps its pivot positive will do —
/// zero against the padding, the identity on the padding.
HSD void dense_padding_corner(double* D, int n_dense, int n_pad, int first, int stride) {
  const int n = n_pad - n_dense;
  for (int e = first; e < n * n; e += stride) {
    const int a = e / n, b = e % n;
    D[size_t(n_dense + a) * kDenseLd + n_dense + b] = a != b ? 0.0 : (a == 0 ? 1e300 : 1.0);
  }
}

/// After the (optional) all-reduce: Jacobi scaling (fixed at iteration 0), LM diagonal, inactive coordinates.
///   S = Sp Sraw Sp + D_p^2,  g = Sp (g_p + g_schur),  g_full = Sp g_p,  D_p^2 = clamp(Sp^2 diag(J'J), 1e-6, 1e32) / radius.
/// A workgroup past the last block row (single shard: gridDim.x = n_cp + 1 [+ border workgroups]) does the work of
/// k_pack_exchange + k_cost_reduce concurrently: with nothing exchanged, neither side reads what the other writes (the block
/// rows use the radius and the scaling flag, which the bookkeeping leaves alone; `done` only makes them skip unused work).
__global__ void __launch_bounds__(kBlock) k_finalize_reduced(Tables T, int n_splits) {
  DevState* st = T.st;
  if (int(blockIdx.x) >= T.sp.n_cp) {
    const int extra = int(blockIdx.x) - T.sp.n_cp;
    if (extra == 0)
      pack_exchange_body(T, 1);
    else  // border blocks of a single shard: H_pb straight from the accumulation splits (no k_reduce_partials in front)
      finalize_border_body(T, extra - 1, int(gridDim.x) - T.sp.n_cp - 1, n_splits);
    return;
  }
  if (st->done) return;
  const int i = blockIdx.x, tid = threadIdx.x;
  const int ncb = 6 * T.bw;
  const double* X = T.xbuf;
  const double radius = st->radius;
  const bool fresh = !st->scaling_ready;
  auto scale_of = [&](int rho) { return fresh ? 1.0 / (1.0 + sqrt(X[T.xo_dj + rho])) : T.scale_p[rho]; };
  for (int e = tid; e < 6 * ncb; e += kBlock) {
    const int a = e / ncb, c = e % ncb;
    const int rho = 6 * i + a, sigma = 6 * i + c;
    double out = 0.0;
    if (sigma < T.np) {
      const double sr = scale_of(rho), sc = scale_of(sigma);
      out = sr * sc * X[size_t(rho) * ncb + c];
      if (c == a) {
        const double d = X[T.xo_dj + rho];
        if (d > 0.0) {
          const double d2 = fmin(fmax(sr * sr * d, 1e-6), 1e32) / radius;
          out += d2;
          T.D2p[rho] = d2;
        } else {  // structurally zero column (constant / unobserved): keep the system non-singular, step = 0
          out = 1.0;
          T.D2p[rho] = 0.0;
        }
      }
    }
    T.Sb[size_t(rho) * ncb + c] = out;
    if (T.dense && i >= T.dense_f0 && sigma < T.np && sigma >= rho) {  // dense copy for k_dense_solve_mx: entry and mirror image
      const int ii = rho - 6 * T.dense_f0, jj = sigma - 6 * T.dense_f0;
      T.dense[size_t(ii) * kDenseLd + jj] = out, T.dense[size_t(jj) * kDenseLd + ii] = out;
    }
    if (T.Sb2 && sigma < T.np) {  // reversed copy for the far end of the two-ended factorisation: (rho, sigma) -> (np-1-sigma, np-1-rho)
      const int rv = T.np - 1 - sigma, cv = T.np - 1 - rho;
      T.Sb2[size_t(rv) * ncb + (cv - 6 * (rv / 6))] = out;
    }
  }
  if (T.dense && i >= T.dense_f0) {
    // what the band does not reach in these six rows of the dense copy — pose columns right of it, the padding columns — is zero (and so are the
    // mirror images); the last block row also writes the identity on the padding (without border unknowns: finalize_border_body otherwise)
    // (column / row n_dense of the dense copy is the right-hand side, written below; the padding proper starts behind it)
    const int n_pose = T.np - 6 * T.dense_f0, n_dense = n_pose + T.nb, n_pad = 16 * ((n_dense + 1 + 15) / 16);
    const int j0 = min(6 * (i - T.dense_f0) + ncb, n_pose), n_right = n_pose - j0, n_padc = n_pad - (n_dense + 1), per_row = n_right + n_padc;
    for (int e = tid; e < 6 * per_row; e += kBlock) {
      const int a = e / per_row, q = e % per_row, ii = 6 * (i - T.dense_f0) + a, jj = q < n_right ? j0 + q : n_dense + 1 + (q - n_right);
      T.dense[size_t(ii) * kDenseLd + jj] = 0.0, T.dense[size_t(jj) * kDenseLd + ii] = 0.0;
    }
    if (T.nb == 0 && i == T.sp.n_cp - 1) dense_padding_corner(T.dense, n_dense, n_pad, tid, kBlock);
  }
  if (tid < 6) {
    const int rho = 6 * i + tid;
    const double sr = scale_of(rho);
    const double gp = X[T.xo_g + rho];
    T.g_full[rho] = sr * gp;
    T.g_s[rho] = sr * (gp + X[T.xo_gs + rho]);
    if (T.Sb2) T.g2[T.np - 1 - rho] = sr * (gp + X[T.xo_gs + rho]);
    if (fresh) T.scale_p[rho] = sr;
    T.gabs[rho] = fabs(gp);
    if (T.dense && i >= T.dense_f0) {  // the right-hand side as column (and row) n_dense of the dense copy
      const int ii = rho - 6 * T.dense_f0, n_dense = T.np - 6 * T.dense_f0 + T.nb;
      T.dense[size_t(ii) * kDenseLd + n_dense] = sr * (gp + X[T.xo_gs + rho]), T.dense[size_t(n_dense) * kDenseLd + ii] = sr * (gp + X[T.xo_gs + rho]);
    }
  }
}

}  // namespace hs
