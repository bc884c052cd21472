// kernels_border.hpp — border unknowns (bias splines, gravity): gathers, scaling, bordered solve (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include "kernels_common.hpp"

namespace hs {

// ---------------------------------------------------------------------------------------------------------------------
// Border unknowns (IMU bias-spline control points + gravity; SURVEY a-4): the inertial factor couples every control point of
// the window with a few *dense* unknowns, ordered last:  [gyro bias 3 n_bias | accel bias 3 n_bias | gravity 2] = nb.
//   H_pb (np x nb), H_bb (nb x nb), g_b (nb) are gathered deterministically from the inertial records.
// Record structure exploited: d r_ang / d b_g,j = wg[j] I_3, d r_lin / d b_a,j = wa[j] I_3 (only the weights are stored).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kPbThreads = 192;  // two waves of bias columns + one wave for the two gravity columns

/// w[j] for j in 0 .. 3, zero outside — from registers. k_border_bb needs the weight of ONE bias control point per record, and which one
/// (j = b - first_bias[record]) is a table entry: taken as rec[.. + j] the weight was a second memory round trip behind that entry in every
/// batch of records; all four (the bias splines have order four on the device: 32 bytes next to each other) are requested with the entry instead.
HSD double sel4(const double (&w)[4], int j) {
  const double lo = j == 0 ? w[0] : w[1], hi = j == 2 ? w[2] : w[3];
  return (j < 0 || j > 3) ? 0.0 : (j < 2 ? lo : hi);
}

template <int K>
__global__ void __launch_bounds__(kPbThreads) k_border_pb(Tables T) {
  // block (i, split): rows 6 i .. 6 i + 5 of H_pb. Waves 0-1: thread <-> bias column, one flat loop over the records of the <= K
  // segments that reach control point i (they are contiguous in the segment-major table; eight records in flight per lane). Wave 2: the
  // two gravity columns — every record contributes (42 loads each), so the LANES take records and the sums are combined across the wave;
  // as two more columns of the loop above they kept one wave busy for ~60 of the kernel's 89 us at configs[2].
  // Blocks i >= n_cp (split 0 only) zero the border-border block and its gradient in the exchange buffer for k_border_bb, which
  // follows on the same stream (a launch of its own cost 6 us on the side-stream chain)
  if (T.st->done) return;
  if (int(blockIdx.x) >= T.sp.n_cp) {
    if (blockIdx.y != 0) return;
    const int n = T.nb * T.nb + T.nb, nz = int(gridDim.x) - T.sp.n_cp;
    for (int e = (int(blockIdx.x) - T.sp.n_cp) * blockDim.x + threadIdx.x; e < n; e += nz * blockDim.x) T.xbuf[T.xo_bb + e] = 0.0;
    if (int(blockIdx.x) == T.sp.n_cp && threadIdx.x == 0) T.join_flag[2] = 0u;  // arrival counter of k_border_bb's workgroups
    return;
  }
  const int i = blockIdx.x, sp = blockIdx.y, nsp = gridDim.y;
  const int kb = T.kb, nbias = T.n_bias, nb = T.nb;
  const int IREC = 18 + 36 * K + 2 * kb;
  double* out = T.xpart + size_t(sp) * T.x_count1 + T.xo_pb;
  const int f0 = max(0, i - K + 1), f1 = min(i, T.n_seg - 1);
  const int p0 = T.i_seg_ptr[f0], p1 = T.i_seg_ptr[f1 + 1];
  int seg_end[K];  // end of segment f0 + j in the record table: first control point of record pos = f0 + #{j : pos >= seg_end[j]}
#pragma unroll
  for (int j = 0; j < K; ++j) seg_end[j] = T.i_seg_ptr[min(f0 + j + 1, T.n_seg)];
  auto first_of = [&](int pos) {
    int f = f0;
#pragma unroll
    for (int j = 0; j < K - 1; ++j) f += pos >= seg_end[j] ? 1 : 0;
    return min(f, f1);
  };
  if (threadIdx.x >= 128) {  // ---- gravity columns ----
    const int lane = threadIdx.x - 128;
    double acc[2][6];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int a = 0; a < 6; ++a) acc[c][a] = 0.0;
    for (int pos = p0 + sp + lane * nsp; pos < p1; pos += 64 * nsp) {
      const double* rec = T.i_rec + size_t(pos) * IREC;
      const double* jp = rec + 6 + 6 * (i - first_of(pos));
      const double* jg = rec + 6 + 36 * K + 2 * kb;
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double g0 = jg[2 * r], g1 = jg[2 * r + 1];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
          const double v = jp[r * 6 * K + a];
          acc[0][a] = fma(v, g0, acc[0][a]), acc[1][a] = fma(v, g1, acc[1][a]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int a = 0; a < 6; ++a) {
        const double v = wave_sum_fast(acc[c][a]);
        if (lane == 0) out[size_t(6 * i + a) * nb + 6 * nbias + c] = v;
      }
    return;
  }
  for (int beta = threadIdx.x; beta < 6 * nbias; beta += 128) {  // ---- bias columns ----
    double acc[6] = {0, 0, 0, 0, 0, 0};
    const int kind = beta < 3 * nbias ? 0 : 1;
    const int bb = (beta - 3 * nbias * kind) / 3, cc = (beta - 3 * nbias * kind) % 3;
    // Only the records whose bias window covers point bb contribute (first_bias in [bb - kb + 1, bb]: one contiguous range of the table,
    // T.i_bias_ptr); a control point meets the records of K segments, i.e. one or two of the n_bias bias intervals, so most columns
    // of a row block have nothing to add. Same members of the split (pos = p0 + sp mod nsp) in the same order: the skipped terms were exact zeros.
    const int r_lo = max(p0, T.i_bias_ptr[max(0, bb - kb + 1)]), r_hi = min(p1, T.i_bias_ptr[bb + 1]);
    const int start = r_lo + ((p0 + sp - r_lo) % nsp + nsp) % nsp;
    // branch-free body (clamped weight index, masked weight) so that the loads of eight records are in flight together
#pragma unroll 8
    for (int pos = start; pos < r_hi; pos += nsp) {
      const double* rec = T.i_rec + size_t(pos) * IREC;
      const int j = bb - T.i_first_bias[pos];
      const bool ok = j >= 0 && j < kb;
      const double wv = rec[6 + 36 * K + kind * kb + (ok ? j : 0)];  // (all four weights + a select, as in k_border_bb: 54 us instead of 33 — four times the loads)
      const double wgt = ok ? wv : 0.0;
      const double* row = rec + 6 + (3 * kind + cc) * 6 * K + 6 * (i - first_of(pos));
#pragma unroll
      for (int a = 0; a < 6; ++a) acc[a] = fma(row[a], wgt, acc[a]);
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) out[size_t(6 * i + a) * nb + beta] = acc[a];
  }
}

/// (Round 6 dealt the records of a bias point to six workgroups — 108 instead of 18, a ticket per bias point, the last one adds the sums up in member
///  order — to shorten what looks like nine dependent rounds of record loads per lane: 59.6 us instead of 46 at configs[2]. The loop is not what
///  bounds this kernel. Not kept.)
/// H_bb and J_b' r. One workgroup per bias control point b (gyro and accel parts): the records whose bias window covers b are
/// dealt to 256 lanes, sums are combined wave by wave in a fixed order; each entry of the exchange buffer has a single writer
/// (the region is zero-filled first by the extra workgroups of k_border_pb). The gravity block is accumulated per b over the records that START at b
/// (every record exactly once) into T.gravity_part[b][5] and summed by the workgroup that finishes last.
template <int K>
__global__ void __launch_bounds__(kBlock) k_border_bb(Tables T) {
  constexpr int NV = 2 * hsd::kMaxOrder + 18 + 5;
  __shared__ double red[kBlock / 64][NV];
  if (T.st->done) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int kb = T.kb, nbias = T.n_bias, nb = T.nb;
  const int IREC = 18 + 36 * K + 2 * kb;
  double* Hbb = T.xbuf + T.xo_bb;
  double* gb = T.xbuf + T.xo_gb;
  const int og = 0, oa = 3 * nbias, ogr = 6 * nbias;
  // records whose bias window covers b: first_bias in [b - kb + 1, b]
  const int p0 = T.i_bias_ptr[max(0, b - kb + 1)], p1 = T.i_bias_ptr[b + 1], pown = T.i_bias_ptr[b];
  // [gg(kMaxOrder) | aa(kMaxOrder) | ggr 6 | agr 6 | rg 3 | ra 3 | gravity h00 h01 h11 g0 g1] — separate register arrays: as slices of ONE array
  // addressed through pointers the 39 sums lived in scratch memory (320 bytes per lane), a load and a store per multiply-add
  constexpr int KM = hsd::kMaxOrder;
  double gg[KM], aa[KM], ggr[6], agr[6], rg[3], ra[3], hg[5];
#pragma unroll
  for (int e = 0; e < KM; ++e) gg[e] = aa[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 6; ++e) ggr[e] = agr[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 3; ++e) rg[e] = ra[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 5; ++e) hg[e] = 0.0;
#pragma unroll 2
  for (int pos = p0 + tid; pos < p1; pos += kBlock) {
    const double* rec = T.i_rec + size_t(pos) * IREC;
    const int j = b - T.i_first_bias[pos];
    const double* wgp = rec + 6 + 36 * K;
    const double* wap = wgp + kb;
    const double* jg = wap + kb;
    double wgb, wab;
    if (kb == 4) {  // (wave uniform; sel4: no load depends on j. A weight outside the record's window multiplies by an exact zero)
      const double wg4[4] = {wgp[0], wgp[1], wgp[2], wgp[3]}, wa4[4] = {wap[0], wap[1], wap[2], wap[3]};
      wgb = sel4(wg4, j), wab = sel4(wa4, j);
#pragma unroll
      for (int d = 0; d < 4; ++d) gg[d] = fma(wgb, sel4(wg4, j + d), gg[d]), aa[d] = fma(wab, sel4(wa4, j + d), aa[d]);
    } else {
      wgb = wgp[j], wab = wap[j];
#pragma unroll
      for (int d = 0; d < KM; ++d)
        if (d < kb && j + d < kb) gg[d] = fma(wgb, wgp[j + d], gg[d]), aa[d] = fma(wab, wap[j + d], aa[d]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      ggr[2 * c] = fma(wgb, jg[2 * c], ggr[2 * c]), ggr[2 * c + 1] = fma(wgb, jg[2 * c + 1], ggr[2 * c + 1]);
      agr[2 * c] = fma(wab, jg[2 * (3 + c)], agr[2 * c]), agr[2 * c + 1] = fma(wab, jg[2 * (3 + c) + 1], agr[2 * c + 1]);
      rg[c] = fma(wgb, rec[c], rg[c]), ra[c] = fma(wab, rec[3 + c], ra[c]);
    }
    if (pos >= pown) {  // j == 0: this record starts at b -> its gravity terms are counted here
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        hg[0] = fma(jg[2 * r], jg[2 * r], hg[0]), hg[1] = fma(jg[2 * r], jg[2 * r + 1], hg[1]), hg[2] = fma(jg[2 * r + 1], jg[2 * r + 1], hg[2]);
        hg[3] = fma(jg[2 * r], rec[r], hg[3]), hg[4] = fma(jg[2 * r + 1], rec[r], hg[4]);
      }
    }
  }
  // wave sums, then the waves in index order (fixed order: bit-reproducible); slot e of red[wave] as in the layout above. (wave_sum_fast: the
  // butterfly of wave_sum with four of its six levels on the DPP cross bar — as LDS permutes the 39 sums were 8 - 13 us of the kernel whatever
  // the number of records. Sums per row of sixteen lanes + sixteen rows added up by lane 0 were tried: 59.8 us instead of 42.7. Not kept.)
#define HS_BB_REDUCE(arr, n, base)                                   \
  _Pragma("unroll") for (int e = 0; e < (n); ++e) {                  \
    arr[e] = wave_sum_fast(arr[e]);                                  \
    if (lane == 0) red[wave][(base) + e] = arr[e];                   \
  }
  HS_BB_REDUCE(gg, KM, 0)
  HS_BB_REDUCE(aa, KM, KM)
  HS_BB_REDUCE(ggr, 6, 2 * KM)
  HS_BB_REDUCE(agr, 6, 2 * KM + 6)
  HS_BB_REDUCE(rg, 3, 2 * KM + 12)
  HS_BB_REDUCE(ra, 3, 2 * KM + 15)
  HS_BB_REDUCE(hg, 5, 2 * KM + 18)
#undef HS_BB_REDUCE
  __syncthreads();
  if (tid != 0) return;
#define HS_BB_TOTAL(arr, n, base)                                    \
  _Pragma("unroll") for (int e = 0; e < (n); ++e) {                  \
    double t = 0.0;                                                  \
    for (int w = 0; w < kBlock / 64; ++w) t += red[w][(base) + e];   \
    arr[e] = t;                                                      \
  }
  HS_BB_TOTAL(gg, KM, 0)
  HS_BB_TOTAL(aa, KM, KM)
  HS_BB_TOTAL(ggr, 6, 2 * KM)
  HS_BB_TOTAL(agr, 6, 2 * KM + 6)
  HS_BB_TOTAL(rg, 3, 2 * KM + 12)
  HS_BB_TOTAL(ra, 3, 2 * KM + 15)
  HS_BB_TOTAL(hg, 5, 2 * KM + 18)
#undef HS_BB_TOTAL
#pragma unroll
  for (int d = 0; d < KM; ++d)  // (compile-time d: a run-time index would put gg / aa back into scratch memory)
    if (d < kb && b + d < nbias)
    for (int c = 0; c < 3; ++c) {
      const int r0 = og + 3 * b + c, c0 = og + 3 * (b + d) + c;
      Hbb[size_t(r0) * nb + c0] = gg[d], Hbb[size_t(c0) * nb + r0] = gg[d];
      const int r1 = oa + 3 * b + c, c1 = oa + 3 * (b + d) + c;
      Hbb[size_t(r1) * nb + c1] = aa[d], Hbb[size_t(c1) * nb + r1] = aa[d];
    }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      Hbb[size_t(og + 3 * b + c) * nb + ogr + e] = ggr[2 * c + e], Hbb[size_t(ogr + e) * nb + og + 3 * b + c] = ggr[2 * c + e];
      Hbb[size_t(oa + 3 * b + c) * nb + ogr + e] = agr[2 * c + e], Hbb[size_t(ogr + e) * nb + oa + 3 * b + c] = agr[2 * c + e];
    }
#pragma unroll
  for (int c = 0; c < 3; ++c) gb[og + 3 * b + c] = rg[c], gb[oa + 3 * b + c] = ra[c];
#pragma unroll
  for (int e = 0; e < 5; ++e) T.gravity_part[5 * b + e] = hg[e];
  // Gravity-gravity block and J_g' r = sum of the per-bias-point partials in index order, by the workgroup that arrives last (the counter is
  // reset by the zero-fill workgroups of k_border_pb, in front of this kernel on the same stream): a launch of its own ended the side-stream
  // chain with 4.6 us of latency
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  wait_vmem();
  if (atomicAdd(T.join_flag + 2, 1u) != gridDim.x - 1) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  double h[5] = {0, 0, 0, 0, 0};
  for (int bb0 = 0; bb0 < nbias; bb0 += 8) {  // the loads of eight bias points in flight together (one dependent load per term cost ~1 us each)
    double t[8][5];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 5; ++e) t[u][e] = bb0 + u < nbias ? __builtin_nontemporal_load(T.gravity_part + 5 * (bb0 + u) + e) : 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 5; ++e) h[e] += t[u][e];
  }
  Hbb[size_t(ogr) * nb + ogr] = h[0], Hbb[size_t(ogr) * nb + ogr + 1] = h[1];
  Hbb[size_t(ogr + 1) * nb + ogr] = h[1], Hbb[size_t(ogr + 1) * nb + ogr + 1] = h[2];
  gb[ogr] = h[3], gb[ogr + 1] = h[4];
  // Both gathers are complete (k_border_pb ended in front of this launch, every other workgroup of this one released its part before its
  // ticket): say so to the border workgroups of k_finalize_reduced on the other stream (Tables::gather_epoch, gather_wait)
  if (T.gather_epoch) __hip_atomic_store(T.join_flag + kGatherFlag, T.gather_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

/// Bounded wait (2 s) of a workgroup for `*flag == epoch`, then an agent-scope acquire by every wave.
HSD void flag_wait(const Tables& T, const unsigned* flag, unsigned epoch) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > 200000000ll) {
        give_up(T.st);
        break;
      }
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

/// The border workgroups of k_finalize_reduced wait here for the gathers of the side stream (round 6). As an event between the two streams the
/// same dependency cost 13 us on the chain of every iteration — main stream idle from the end of k_assemble to the event's arrival — and the
/// launch of k_finalize_reduced behind it; now the launch is in flight and the pose rows are done when the flag arrives. Bounded: 2 s.
HSD void gather_wait(const Tables& T) { flag_wait(T, T.join_flag + kGatherFlag, T.gather_epoch); }

/// Scaling / damping of the border blocks after the exchange:  S_pb = Sp H_pb Sb,  S_bb = Sb H_bb Sb + D_b^2,  g_b = Sb g_b.
/// (n_splits = 0: H_pb was reduced into the exchange buffer by k_reduce_partials; > 0: summed here over the accumulation splits, same order)
HSD void finalize_border_body(const Tables& T, const int wg, const int n_wg, const int n_splits) {
  DevState* st = T.st;
  if (st->done) return;
  const bool fresh = !st->scaling_ready;  // (solver state: requested before the wait — it does not depend on the gathers)
  const double radius = st->radius;
  if (T.gather_epoch) gather_wait(T);
  const int nb = T.nb, np = T.np;
  const double* X = T.xbuf;
  auto sb_of = [&](int b) { return fresh ? 1.0 / (1.0 + sqrt(X[T.xo_bb + size_t(b) * nb + b])) : T.scale_b[b]; };
  auto sp_of = [&](int rho) { return fresh ? 1.0 / (1.0 + sqrt(X[T.xo_dj + rho])) : T.scale_p[rho]; };
  const int total = (np + nb) * nb;
  for (int e = wg * blockDim.x + threadIdx.x; e < total; e += n_wg * blockDim.x) {
    const int row = e / nb, c = e % nb;
    if (row < np) {
      double hpb = 0.0;
      if (n_splits > 0) {
        // (every split in flight, added in split order: as a plain loop over k the loads went out one per memory round trip — with sixteen
        //  accumulation splits the border workgroups of k_finalize_reduced took 12 - 16 us for one or two entries per lane; stamps, round 6)
        for (int k0 = 0; k0 < n_splits; k0 += 16) {  // (n_split <= 16, prepare())
          double v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = k0 + u < n_splits ? T.xpart[size_t(k0 + u) * T.x_count1 + T.xo_pb + e] : 0.0;
#pragma unroll
          for (int u = 0; u < 16; ++u) hpb += v[u];
        }
      } else {
        hpb = X[T.xo_pb + e];
      }
      const double spb = sp_of(row) * hpb * sb_of(c);
      T.Spb[e] = spb;
      if (T.dense && row >= 6 * T.dense_f0) {
        const int ii = row - 6 * T.dense_f0, jj = np - 6 * T.dense_f0 + c;
        T.dense[size_t(ii) * kDenseLd + jj] = spb, T.dense[size_t(jj) * kDenseLd + ii] = spb;
      }
    } else {
      const int b = row - np;
      const double sr = sb_of(b), sc = sb_of(c);
      double out = sr * sc * X[T.xo_bb + size_t(b) * nb + c];
      if (b == c) {
        const double d = X[T.xo_bb + size_t(b) * nb + b];
        if (d > 0.0) {
          const double d2 = fmin(fmax(sr * sr * d, 1e-6), 1e32) / radius;
          out += d2;
          T.D2b[b] = d2;
        } else {
          out = 1.0;
          T.D2b[b] = 0.0;
        }
        const double g = X[T.xo_gb + b];
        T.gb_s[b] = sr * g;
        if (T.dense) {  // the right-hand side as column (and row) n_dense of the dense copy
          const int ii = np - 6 * T.dense_f0 + b, n_dense = np - 6 * T.dense_f0 + nb;
          T.dense[size_t(ii) * kDenseLd + n_dense] = sr * g, T.dense[size_t(n_dense) * kDenseLd + ii] = sr * g;
        }
        if (fresh) T.scale_b[b] = sr;
        T.gabs[T.np + b] = fabs(g);
      }
      T.Sbb[size_t(b) * nb + c] = out;
      if (T.dense) T.dense[size_t(np - 6 * T.dense_f0 + b) * kDenseLd + (np - 6 * T.dense_f0 + c)] = out;
    }
  }
  if (T.dense) {  // padding of the dense copy behind the right-hand side column: zero against the border unknowns (the pose rows: k_finalize_reduced)
    const int n_pose = np - 6 * T.dense_f0, n_dense = n_pose + nb, n_pad = 16 * ((n_dense + 1 + 15) / 16), n_padc = n_pad - (n_dense + 1);
    for (int e = wg * blockDim.x + threadIdx.x; e < n_padc * nb; e += n_wg * blockDim.x) {
      const int ii = n_dense + 1 + e / nb, jj = n_pose + e % nb;
      T.dense[size_t(ii) * kDenseLd + jj] = 0.0, T.dense[size_t(jj) * kDenseLd + ii] = 0.0;
    }
    dense_padding_corner(T.dense, n_dense, n_pad, wg * blockDim.x + threadIdx.x, n_wg * blockDim.x);
  }
}

__global__ void __launch_bounds__(kBlock) k_finalize_border(Tables T) { finalize_border_body(T, blockIdx.x, gridDim.x, 0); }

// ---------------------------------------------------------------------------------------------------------------------
// Bordered solve:  [S_pp S_pb; S_bp S_bb][x_p; x_b] = [g_p; g_b] with S_pp = U'U banded.
//   Z = U^-T S_pb  (k_border_forward: one workgroup per group of border columns, column-oriented forward sweep)
//   C = S_bb - Z'Z, h = g_b - Z'y  (k_border_schur, one workgroup per border row)
//   C x_b = h (dense Cholesky in LDS), y' = y - Z x_b  (k_border_solve, one workgroup)   then the banded backward sweep on y'.
// ---------------------------------------------------------------------------------------------------------------------
/// Row rho of the forward-solved right-hand side y. One-ended factorisation: T.ybuf. Two-ended: the near end's rows (top and middle,
/// rho < T.y_split) are in T.ybuf, the far end's in T.ybuf2 in reversed order.
HSD double* y_slot(const Tables& T, int rho) { return rho < T.y_split ? T.ybuf + rho : T.ybuf2 + (T.np - 1 - rho); }

constexpr int kBorderCols = 2;  // right-hand sides per workgroup in the forward sweep
constexpr int kBorderLd = 6;    // LDS row stride of the pending rows (doubles): rows of a power-of-two size put every fourth lane on the same banks
                                // (16-way conflict on the row read-modify-write of every step); + 16 bytes keeps the alignment and spreads them

/// The sweep of a workgroup starts at block row m0: rows above it are zero in its columns of S_pb, hence in Z — the rows of the leading
/// constant control points (j_lo, decoupled: k_factor_decoupled_rows) and the rows before the first residual that involves the
/// workgroup's bias points (T.bfwd_start: a bias point meets the pose rows of its own few seconds only).
/// (local_rows = 0 on a shard of a distributed solve: the record table of one shard says nothing about the rows the other shards fill)
__global__ void __launch_bounds__(kBlock) k_border_forward(Tables T, int j_lo, int local_rows) {  // blockDim = 64 x waves covering the 6 (bw - 1) pending rows (>= 128)
  HS_DYNAMIC_LDS(smem);
  if (T.st->done) return;
  const int tid = threadIdx.x;
  const int bw = T.bw, ncb = 6 * bw, nb = T.nb, np = T.np, n_blk = np / 6;
  const int c0 = blockIdx.x * kBorderCols, ncols = min(kBorderCols, nb - c0);
  const int m0 = min(local_rows ? max(j_lo, T.bfwd_start[blockIdx.x]) : j_lo, n_blk - 1);
  double* z = smem;  // np x kBorderLd: pending right-hand side rows, overwritten by the solution
  for (int e = tid; e < np * kBorderCols; e += blockDim.x) {
    const int rho = e / kBorderCols, c = e % kBorderCols;
    z[rho * kBorderLd + c] = c < ncols ? T.Spb[size_t(rho) * nb + c0 + c] : 0.0;
  }
  __syncthreads();
  __shared__ __attribute__((aligned(16))) double zi[2][6 * kBorderCols];  // z_m, double buffered: ONE workgroup barrier per block row
  const int n_pend = 6 * (bw - 1);
  // Operands of step m are requested D steps ahead (the sweep is a dependency chain over the block rows: a load issued inside
  // the step would put a full L2 round trip on it). Thread t < n_pend: the six factor entries U[6m + a][6 + t]; thread
  // (a, c) < 6 x kBorderCols (all in wave 0): column a of W_(m+1) = U_(m+1,m+1)^-1 — the diagonal solve of block row m + 1 is done by
  // wave 0 right after its own part of update m (the six rows of block row m + 1 are pending rows 0 .. 5: lanes of wave 0), so that
  // z_(m+1) is published by the same barrier that ends step m.
  constexpr int D = 4;
  const bool pend = tid < n_pend, diag = tid < 6 * kBorderCols;
  const int da = diag ? tid / kBorderCols : 0, dc = diag ? tid % kBorderCols : 0;
  double ur[D][6], wr[D][6];
  auto request = [&](int m, double* u, double* w) {
    const int mm = m < n_blk ? m : 0, mw = m + 1 < n_blk ? m + 1 : 0;
    const double* src = T.Ub + size_t(6 * mm) * ncb + 6 + (pend ? tid : 0);
#pragma unroll
    for (int a = 0; a < 6; ++a) u[a] = src[size_t(a) * ncb];
    // (W')[a][k] = W[k][a], k <= a ; packed index of (k, a) = k*6 - k(k-1)/2 + (a - k)
    const double* W = T.Ubk + size_t(mw) * 24;
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = W[k <= da ? k * 6 - k * (k - 1) / 2 + (da - k) : 0];
  };
  auto diag_solve = [&](int m, const double* w) {  // wave 0 (every lane calls): z_m = U_mm^-T s_m = W' s_m, into zi[m & 1] and over s_m
    double v = 0.0;
    if (diag) {
#pragma unroll
      for (int k = 0; k < 6; ++k) v = fma(k <= da ? w[k] : 0.0, z[(6 * m + k) * kBorderLd + dc], v);
    }
    wait_lds();  // every lane has read s_m before any lane overwrites it (the device runs the wave in lock step: the counter is already zero)
    if (diag) {
      zi[m & 1][tid] = v;
      z[(6 * m + da) * kBorderLd + dc] = v;
    }
  };
  {  // z_m0
    double w0[6];
    const double* W = T.Ubk + size_t(m0) * 24;
#pragma unroll
    for (int k = 0; k < 6; ++k) w0[k] = W[k <= da ? k * 6 - k * (k - 1) / 2 + (da - k) : 0];
    if (tid < 64) diag_solve(m0, w0);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) request(m0 + d, ur[d], wr[d]);
  __syncthreads();
  for (int mb = m0; mb < n_blk; mb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int m = mb + d;
      if (m >= n_blk) break;
      // pending rows of blocks m+1 .. m+bw-1: s_(i,c') -= sum_a U[6m+a][6(i-m)+c'] z_m[a]
      if (pend) {
        const int rho = 6 * (m + 1) + tid;
        if (rho < np) {  // 16-byte LDS operations: 24 reads of z_m, 4 + 4 for the row (the scalar form issued 112 per step)
          double2* zr = reinterpret_cast<double2*>(z + rho * kBorderLd);
          const double2* zm = reinterpret_cast<const double2*>(zi[m & 1]);
          double2 acc[kBorderCols / 2];
#pragma unroll
          for (int c = 0; c < kBorderCols / 2; ++c) acc[c] = zr[c];
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = 0; c < kBorderCols / 2; ++c) {
              const double2 t = zm[a * (kBorderCols / 2) + c];
              acc[c].x = fma(-ur[d][a], t.x, acc[c].x), acc[c].y = fma(-ur[d][a], t.y, acc[c].y);
            }
#pragma unroll
          for (int c = 0; c < kBorderCols / 2; ++c) zr[c] = acc[c];
        }
      }
      if (tid < 64 && m + 1 < n_blk) {
        wait_lds();  // wave 0: its rows of block row m + 1 are final
        diag_solve(m + 1, wr[d]);
      }
      request(m + D, ur[d], wr[d]);
      __syncthreads();
    }
  }
  for (int e = tid; e < np * kBorderCols; e += blockDim.x) {
    const int rho = e / kBorderCols, c = e % kBorderCols;
    if (c < ncols) T.Zb[size_t(rho) * nb + c0 + c] = z[rho * kBorderLd + c];
  }
}

/// Forward sweep Z = U^-T S_pb for a factorisation from BOTH ENDS (k_band_factor_la, grid 2): the elimination order is top rows and
/// (reversed) bottom rows side by side, then the middle rows, and the sweep of the border columns has to follow it. grid = (column
/// groups, 2): workgroup (g, 1) sweeps the far end's own block rows in reversed coordinates with the far end's factor and hands the
/// updates that its last rows leave on the middle rows over (T.join_flag[kBfFlagBase + g], agent scope); workgroup (g, 0) sweeps the
/// top rows, adds the hand-over in front of the first middle block row and carries on through the middle rows with the near end's
/// factor — the same recurrence as k_border_forward, one block row per step.
struct BfJob {
  const double* Ub;
  const double* Ubk;
  int n_rows;    // block rows this job eliminates (near end: top + middle)
  int reversed;  // row rho of this job is row np - 1 - rho of the system
  // Pipelined behind the factorisation (k_band_factor_mx on the main stream, this kernel on the side stream): progress[0] / progress[kProgressStride] -
  // progress_base = leading block rows of this job whose factor row / inverted diagonal block is complete in memory (MfmaJob::progress).
  // An extra wave of the workgroup (the last one) does nothing but poll the two words, in front of the barrier that ends a step; the factor
  // is then read with agent-scope loads (lines of the previous iteration's factor may still sit in this XCD's L2). nullptr: the factor is complete.
  const unsigned* progress;
  unsigned progress_base;
};
constexpr int kBfFlagBase = 4 + 2 * 512;  // behind the super-block flags of the backward sweep (kernels_backward_sb.hpp)
static_assert(kGatherFlag == kBfFlagBase + 512 + 4 * kProgressStride, "kGatherFlag: the word behind the progress words");

__global__ void __launch_bounds__(kBlock) k_border_forward2(Tables T, BfJob j0, BfJob j1, int m_junction, int j_lo, int local_rows, double* handover) {
  HS_DYNAMIC_LDS(smem);
  if (T.st->done) return;
  // The far-end (producer) jobs are the LOWER-numbered workgroups of the grid (x runs fastest in dispatch order): every consumer that spins
  // on a hand-over flag is dispatched behind its producer, so the spin cannot starve the producers of workgroup slots.
  const int tid = threadIdx.x, grp = blockIdx.x, far = 1 - int(blockIdx.y);
  const BfJob J = far ? j1 : j0;
  const int bw = T.bw, ncb = 6 * bw, nb = T.nb, np = T.np, w_mid = bw - 1;
  const int n_rows = J.n_rows, nz = 6 * (n_rows + (far ? w_mid : 0));  // rows of z: own rows (+ the middle rows the far end leaves updates on)
  const int c0 = grp * kBorderCols, ncols = min(kBorderCols, nb - c0);
  const int m0 = far ? 0 : min(min(local_rows ? max(j_lo, T.bfwd_start[grp]) : j_lo, n_rows - 1), m_junction - 1);
  constexpr int D = 4;  // operands of a step are requested D steps ahead (below)
  const bool pipe = J.progress != nullptr;
  const bool poller = pipe && tid >= int(blockDim.x) - 64;  // (the launch adds the wave)
  int seen = 0;  // poller: complete block rows (factor row AND inverted block) at the last look
  auto await = [&](int rows) {  // poller wave: until the leading `rows` block rows of the factor are complete
    const int need = min(rows, n_rows);
    if (seen >= need) return;
    const long long t0 = wall_clock64();
    for (;;) {
      // word = launch epoch << 12 | complete rows (n_cp <= 1024 < 4096). The epoch is compared for EQUALITY: the words are never reset, and a word
      // left by any earlier launch — however long ago, e.g. after a stretch of windows that did not take the pipelined path — counts as "nothing yet"
      // (a signed difference to the base would read a word 2^19 or more epochs old as "everything complete").
      const unsigned wa = __hip_atomic_load(J.progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned wb = __hip_atomic_load(J.progress + kProgressStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int a = (wa >> 12) == (J.progress_base >> 12) ? int(wa & 4095u) : 0;
      const int b = (wb >> 12) == (J.progress_base >> 12) ? int(wb & 4095u) : 0;
      seen = min(a, b);
      if (seen >= need) break;
      __builtin_amdgcn_s_sleep(16);  // (~0.4 us: a hundred pollers at full rate on the words the factorisation writes delayed its stores)
      if (wall_clock64() - t0 > 200000000ll) {
        give_up(T.st);
        break;
      }
    }
  };
  double* z = smem;
  if (poller) await(m0 + D + 2);  // W of row m0, the requests of rows m0 .. m0 + D - 1 (each with W of the row behind it) and the first one of the loop
  if (pipe) {
    // Pipelined: this launch is NOT ordered behind k_finalize_reduced by the host (round 6: the event that did it was 7 us of idle main stream
    // between the finalisation and the factorisation) — it follows the border gathers on its own stream and may be polling long before the
    // factorisation starts. The first rows of the factor are its proof that S_pb is final: read it behind them, with an acquire.
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  for (int e = tid; e < nz * kBorderCols; e += blockDim.x) {
    const int rho = e / kBorderCols, c = e % kBorderCols;
    const bool own = rho < 6 * n_rows;
    const int row = far ? np - 1 - rho : rho;
    z[rho * kBorderLd + c] = (c < ncols && own) ? T.Spb[size_t(row) * nb + c0 + c] : 0.0;
  }
  __syncthreads();
  __shared__ __attribute__((aligned(16))) double zi[2][6 * kBorderCols];
  const int n_pend = 6 * w_mid;
  const bool pend = tid < n_pend, diag = tid < 6 * kBorderCols;
  const int da = diag ? tid / kBorderCols : 0, dc = diag ? tid % kBorderCols : 0;
  double ur[D][6], wr[D][6];
  auto request = [&](int m, double* u, double* w) {
    if (poller) return;  // (its loads would queue in front of the polls)
    const int mm = m < n_rows ? m : 0, mw = m + 1 < n_rows ? m + 1 : 0;
    const double* src = J.Ub + size_t(6 * mm) * ncb + 6 + (pend ? tid : 0);
    const double* W = J.Ubk + size_t(mw) * 24;
    if (pipe) {
#pragma unroll
      for (int a = 0; a < 6; ++a) u[a] = __hip_atomic_load(src + size_t(a) * ncb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int k = 0; k < 6; ++k) w[k] = __hip_atomic_load(W + (k <= da ? k * 6 - k * (k - 1) / 2 + (da - k) : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
      for (int a = 0; a < 6; ++a) u[a] = src[size_t(a) * ncb];
#pragma unroll
      for (int k = 0; k < 6; ++k) w[k] = W[k <= da ? k * 6 - k * (k - 1) / 2 + (da - k) : 0];
    }
  };
  auto diag_solve = [&](int m, const double* w) {  // (wave 0, every lane calls: see k_border_forward)
    double v = 0.0;
    if (diag) {
#pragma unroll
      for (int k = 0; k < 6; ++k) v = fma(k <= da ? w[k] : 0.0, z[(6 * m + k) * kBorderLd + dc], v);
    }
    wait_lds();
    if (diag) {
      zi[m & 1][tid] = v;
      z[(6 * m + da) * kBorderLd + dc] = v;
    }
  };
  {
    double w0[6];
    const double* W = J.Ubk + size_t(m0) * 24;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const double* wp = W + (k <= da ? k * 6 - k * (k - 1) / 2 + (da - k) : 0);
      w0[k] = pipe ? __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *wp;
    }
    if (tid < 64) diag_solve(m0, w0);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) request(m0 + d, ur[d], wr[d]);
  __syncthreads();
  double* ho = handover + size_t(grp) * (6 * w_mid * kBorderCols);
  const unsigned* flag = T.join_flag + kBfFlagBase + grp;
  for (int mb = m0; mb < n_rows; mb += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int m = mb + d;
      if (m >= n_rows) break;
      if (pend) {
        const int rho = 6 * (m + 1) + tid;
        if (rho < nz) {
          double2* zr = reinterpret_cast<double2*>(z + rho * kBorderLd);
          const double2* zm = reinterpret_cast<const double2*>(zi[m & 1]);
          double2 acc[kBorderCols / 2];
#pragma unroll
          for (int c = 0; c < kBorderCols / 2; ++c) acc[c] = zr[c];
#pragma unroll
          for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = 0; c < kBorderCols / 2; ++c) {
              const double2 t = zm[a * (kBorderCols / 2) + c];
              acc[c].x = fma(-ur[d][a], t.x, acc[c].x), acc[c].y = fma(-ur[d][a], t.y, acc[c].y);
            }
#pragma unroll
          for (int c = 0; c < kBorderCols / 2; ++c) zr[c] = acc[c];
        }
      }
      if (!far && m + 1 == m_junction) {  // ---- junction: what the far end's sweep left on the middle rows (reversed order) ----
        __syncthreads();
        if (tid == 0) {
          const long long t0 = wall_clock64();
          while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < T.join_epoch) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > 200000000ll) {
              give_up(T.st);
              break;
            }
          }
        }
        __syncthreads();
        for (int e = tid; e < 6 * w_mid * kBorderCols; e += blockDim.x) {
          const int r = e / kBorderCols, c = e % kBorderCols;
          z[(6 * (m_junction + w_mid) - 1 - r) * kBorderLd + c] += __builtin_nontemporal_load(ho + e);
        }
        __syncthreads();
      }
      if (tid < 64 && m + 1 < n_rows) {
        wait_lds();
        diag_solve(m + 1, wr[d]);
      }
      request(m + D, ur[d], wr[d]);
      if (poller) await(m + 1 + D + 2);  // what the next step requests: row m + 1 + D and W of the row behind it
      __syncthreads();
    }
  }
  for (int e = tid; e < 6 * n_rows * kBorderCols; e += blockDim.x) {
    const int rho = e / kBorderCols, c = e % kBorderCols;
    if (c < ncols) T.Zb[size_t(far ? np - 1 - rho : rho) * nb + c0 + c] = z[rho * kBorderLd + c];
  }
  if (far) {
    for (int e = tid; e < 6 * w_mid * kBorderCols; e += blockDim.x) ho[e] = z[(6 * n_rows + e / kBorderCols) * kBorderLd + e % kBorderCols];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // (every wave: its stores have reached the L2; the ONE agent-scope release — an L2 write-back on this part — is lane 0's below)
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      __hip_atomic_store(const_cast<unsigned*>(flag), T.join_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (T.sweep_epoch) {
    // Z is complete when the last near-end workgroup is (each of them waited for its far-end partner at the junction): ticket, and the last one
    // tells k_border_schur on the main stream (flag_wait there) — as an event between the streams this was 11 - 13 us of every iteration.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) {
      __threadfence();
      if (atomicAdd(T.join_flag + kGatherFlag + 1, 1u) == gridDim.x - 1) {
        __hip_atomic_store(T.join_flag + kGatherFlag + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(T.join_flag + kGatherFlag + 2, T.sweep_epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}

/// C = S_bb - Z'Z (16 x 16 tile per workgroup, upper tile triangle mirrored) and h = g_b - Z'y. Z rows are staged through LDS
/// in chunks (coalesced, eight loads in flight per lane), the tile is accumulated from LDS.
constexpr int kSchurTile = 16, kSchurRows = 128;

/// (rows of Z above the first non-zero row of either column group are zero — j_lo / T.bfwd_start as in k_border_forward — and are skipped.
///  row_cap: block rows from row_cap on are never skipped. Two-ended elimination: the far end eliminates from the last row upwards, so a
///  column of Z fills every far and middle row ABOVE its first non-zero row of S_pb as well; only rows of the near end's top section,
///  row_cap = the first middle block row, stay zero above it.)
__global__ void __launch_bounds__(kBlock) k_border_schur(Tables T, int j_lo, int local_rows, int row_cap) {
  __shared__ double za[kSchurRows][kSchurTile + 1], zc[kSchurRows][kSchurTile + 1], ys[kSchurRows];
  if (T.st->done) return;
  const int nb = T.nb, np = T.np, tid = threadIdx.x;
  const int bt = blockIdx.x, ct = blockIdx.y;
  if (ct < bt) return;  // lower tiles are written by their mirror
  if (T.sweep_epoch) flag_wait(T, T.join_flag + kGatherFlag + 2, T.sweep_epoch);  // Z from the side stream (k_border_forward2's last workgroup)
  const int ti = tid / kSchurTile, tj = tid % kSchurTile;
  const int b = bt * kSchurTile + ti, c = ct * kSchurTile + tj;
  static_assert(kSchurTile % kBorderCols == 0, "a Schur tile covers whole column groups of the forward sweep");
  constexpr int G = kSchurTile / kBorderCols;
  const int n_groups = (nb + kBorderCols - 1) / kBorderCols;
  int sb = np / 6, sc = np / 6;  // first block row with a non-zero entry in the tile's row / column groups
  for (int g = 0; g < G; ++g) {
    if (bt * G + g < n_groups) sb = min(sb, T.bfwd_start[bt * G + g]);
    if (ct * G + g < n_groups) sc = min(sc, T.bfwd_start[ct * G + g]);
  }
  if (!local_rows) sb = sc = 0;
  const int row0 = 6 * min(min(max(j_lo, max(sb, sc)), row_cap), np / 6);
  double acc = 0.0, hacc = 0.0;
  // 2 x (kSchurRows x 16) operand entries + y per chunk: 16 + 1 loads per lane, issued together — and one chunk AHEAD of the products, so
  // that the memory round trip of chunk r + 1 runs under the 128 FMAs of chunk r (the slowest tile sets the kernel time: six chunks at
  // configs[2], each a full round trip + the products before)
  double va[8], vc[8], yv;
  auto request = [&](int r0) {
    const int nr = min(kSchurRows, np - r0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + u * kBlock, r = e / kSchurTile, k = e % kSchurTile;
      const bool ok = r < nr;
      va[u] = ok && bt * kSchurTile + k < nb ? T.Zb[size_t(r0 + r) * nb + bt * kSchurTile + k] : 0.0;
      vc[u] = ok && ct * kSchurTile + k < nb ? T.Zb[size_t(r0 + r) * nb + ct * kSchurTile + k] : 0.0;
    }
    yv = tid < nr ? *y_slot(T, r0 + tid) : 0.0;
  };
  if (row0 < np) request(row0);
  for (int r0 = row0; r0 < np; r0 += kSchurRows) {
    __syncthreads();  // the products of the previous chunk are done with the LDS operands
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = tid + u * kBlock, r = e / kSchurTile, k = e % kSchurTile;
      za[r][k] = va[u], zc[r][k] = vc[u];
    }
    if (tid < kSchurRows) ys[tid] = yv;
    __syncthreads();
    if (r0 + kSchurRows < np) request(r0 + kSchurRows);
    if (ct == bt) {  // diagonal tiles also form Z'y: by every lane (only column 0 keeps it) — a per-lane condition inside the loop made it a
                     // divergent branch per row, and the diagonal tiles set the kernel time
#pragma unroll 8
      for (int r = 0; r < kSchurRows; ++r) {
        const double a = za[r][ti];
        acc = fma(a, zc[r][tj], acc);
        hacc = fma(a, ys[r], hacc);
      }
    } else {
#pragma unroll 8
      for (int r = 0; r < kSchurRows; ++r) acc = fma(za[r][ti], zc[r][tj], acc);
    }
  }
  if (b < nb && c < nb) {
    const double v = T.Sbb[size_t(b) * nb + c] - acc;
    T.Cb[size_t(b) * nb + c] = v;
    if (ct != bt) T.Cb[size_t(c) * nb + b] = v;
    if (T.dense_border) {  // the same entries in the dense layout k_dense_solve_mx loads its tiles from (padding + corner: written once, prepare())
      T.dense[size_t(b) * kDenseLd + c] = v;
      if (ct != bt) T.dense[size_t(c) * kDenseLd + b] = v;
    }
  }
  if (ct == bt && tj == 0 && b < nb) {
    const double h = T.gb_s[b] - hacc;
    T.hb[b] = h;
    if (T.dense_border) T.dense[size_t(b) * kDenseLd + nb] = h, T.dense[size_t(nb) * kDenseLd + b] = h;  // right-hand side: column (and row) nb
  }
}

/// Dense Cholesky of the border Schur complement C (nb x nb, in LDS, augmented with h as an extra row so that the forward
/// solve comes out of the elimination), column-oriented backward solve, x_b. One barrier per column in both sweeps.
__global__ void __launch_bounds__(kBlock) k_border_solve(Tables T) {
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const int nb = T.nb, tid = threadIdx.x;
  const int ld = nb + 1, n1 = nb + 1;  // rows 0 .. nb-1: C (lower), row nb: h'
  double* C = smem;                    // (nb + 1) x ld
  for (int e = tid; e < nb * nb; e += blockDim.x) C[(e / nb) * ld + e % nb] = T.Cb[e];
  for (int e = tid; e < nb; e += blockDim.x) C[nb * ld + e] = T.hb[e];
  __shared__ int bad;
  if (tid == 0) bad = 0;
  __syncthreads();
  const int ti = tid / 16, tj = tid % 16;  // 16 x 16 lanes over the trailing (i, c) entries
  for (int j = 0; j < nb; ++j) {           // right-looking on the lower triangle; column j is scaled on the fly
    const double d = C[j * ld + j];
    if (tid == 0 && !(d > 0.0)) bad = 1;
    const double inv = 1.0 / (d > 0.0 ? d : 1.0);  // 1 / l_jj^2
    for (int i = j + 1 + ti; i < n1; i += 16) {
      const double lij = C[i * ld + j];
      for (int c = j + 1 + tj; c <= i && c < nb; c += 16) C[i * ld + c] = fma(-lij * inv, C[c * ld + j], C[i * ld + c]);
    }
    lds_barrier();
    // scale column j (not read again by later columns' updates except through these scaled values in the backward sweep)
    const double rs = sqrt(inv);
    for (int i = j + tid; i < n1; i += blockDim.x) C[i * ld + j] = i == j ? d * rs : C[i * ld + j] * rs;
    // (no barrier needed here: column j is not touched by the update of column j + 1, which reads columns > j only ... except
    //  C[c][j+1] entries, which were finalised by the update above and published by the barrier)
  }
  lds_barrier();
  // backward: L' x = y, y = row nb; column oriented, one barrier per column (x goes to its own array)
  double* y = C + nb * ld;
  double* x = C + n1 * ld;
  for (int j = nb - 1; j >= 0; --j) {
    const double xj = y[j] / C[j * ld + j];
    if (tid == 0) x[j] = xj;
    for (int i = tid; i < j; i += blockDim.x) y[i] = fma(-C[j * ld + i], xj, y[i]);
    lds_barrier();
  }
  if (tid == 0 && bad) st->chol_failed = 1;
  for (int bq = tid; bq < nb; bq += blockDim.x) T.xb[bq] = x[bq];
}

/// The same solve with the trailing matrix in REGISTERS (nb + 1 <= 16 R): lane (ti, tj) of the 16 x 16 arrangement owns the entries
/// (i, c) = (ti + 16 r, tj + 16 q), r, q < R, of the augmented lower triangle (row nb = h'). Per column j its owners publish the
/// column through a double-buffered LDS vector (ONE barrier per column, no LDS read-modify-write: the version above spent 1.6 us
/// per column — 160 us at the 98 border unknowns of configs[2] — on three LDS operations per updated entry and two loops per
/// barrier), everybody updates its registers with a[r][q] -= l_ij l_cj / d_j. The scaled columns are kept in LDS for the backward
/// sweep L' x = y, which ONE wave runs without barriers: lane i carries y_i (two per lane), x_j is broadcast with v_readlane.
HSD double cj_or_zero(const double* col, int i, int n) { return col[i < n ? i : 0]; }  // (i < 16 R always; keeps the reads in range by construction)

template <int R>
__global__ void __launch_bounds__(kBlock) k_border_solve_reg(Tables T) {
  HS_DYNAMIC_LDS(smem);
  DevState* st = T.st;
  if (st->done) return;
  const int nb = T.nb, tid = threadIdx.x, n1 = nb + 1;
  constexpr int N = 16 * R;
  const int ld = N + 1;                // odd: a lane-strided walk down a column of Lc is conflict free
  double* Lc = smem + 4 * N;           // nb x ld : Lc[j][i] = l_ij, i >= j (i = nb: forward-solved right-hand side y_j); during the elimination:
                                       // the columns as their owners published them (unscaled) — a scaled copy inside the loop cost 0.24 us per pair
  double* invd = Lc + size_t(nb) * ld; // nb : 1 / l_jj
  const int ti = tid / 16, tj = tid % 16;
  double a[R][R];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int q = 0; q < R; ++q) {
      const int i = ti + 16 * r, c = tj + 16 * q;
      a[r][q] = (i < n1 && c < nb && c <= i) ? (i < nb ? T.Cb[size_t(i) * nb + c] : T.hb[c]) : 0.0;
    }
  __shared__ int bad;
  if (tid == 0) bad = 0;
  auto rsqrt_refined = [](double d) {
    // 1 / l_jj by the hardware estimate + one Newton-Halley step (as in the 6 x 6 panels): the division + square root of the plain
    // formula were a third of the instructions of a column
    const double dd = d > 0.0 ? d : 1.0, y0r = __builtin_amdgcn_rsq(dd), er = fma(-dd * y0r, y0r, 1.0);
    return fma(y0r * er, fma(0.375, er, 0.5), y0r);
  };
  // (publishing a column: its owners are the lanes with tj == (j & 15); the compile-time q loop keeps a[][] in registers — and so does
  //  writing the loop out at each use: a lambda that captures a[][] by reference puts the array into scratch memory)
#define HS_PUBLISH_COLUMN(J, CJ)                                   \
  if (tj == ((J) & 15)) {                                          \
    _Pragma("unroll") for (int q = 0; q < R; ++q) if (q == ((J) >> 4)) { \
      _Pragma("unroll") for (int r = 0; r < R; ++r)(CJ)[ti + 16 * r] = a[r][q]; \
    }                                                              \
  }
  // Two columns per barrier: the owners publish columns j and j + 1 as they are BEFORE the update by column j; every lane forms the
  // updated column j + 1 itself (c1'[i] = c1[i] - l_ij c0[j + 1], the very FMA its owners would have done) and applies both updates.
  // The chain per column is publish -> barrier -> 1 / d -> operands -> update (0.63 us at 57 unknowns, 1.04 us at 99): half the barriers.
  int j = 0;
  const bool bprof = prof_enabled(T.debug_flags, 16) && tid == 0;  // phase stamps -> xpart[8 (700 + j / 2) + ..] (tools/border_phase_timing.py)
  long long* blog = reinterpret_cast<long long*>(T.xpart) + 8 * 700;
  // The loop over the pairs is split by the 16-column block Q the pair lies in (compile time): its owners publish a[.][Q] without a
  // run-time choice of the register column — as a chain of `if (q == j >> 4)` around the stores that choice cost 14 branches per pair,
  // 0.4 us of the 1.5 us a pair took — and tiles left of / above block Q are finished and take no part in the update.
#pragma unroll
  for (int Q = 0; Q < R; ++Q) {
    const int j_end = min(16 * (Q + 1), nb);
    for (j = 16 * Q; j + 1 < j_end; j += 2) {
      double* c0 = Lc + size_t(j) * ld;  // the columns are published straight into their rows of Lc (UNSCALED, column j + 1 as it is
      double* c1 = c0 + ld;              // before the update by column j); the scaling pass behind the elimination finishes them
      if (bprof) blog[8 * (j >> 1) + 0] = wall_clock64();
      if (tj == (j & 15)) {
#pragma unroll
        for (int r = Q; r < R; ++r) c0[ti + 16 * r] = a[r][Q];  // (rows above block Q are finished and never read)
      }
      if (tj == ((j + 1) & 15)) {
#pragma unroll
        for (int r = Q; r < R; ++r) c1[ti + 16 * r] = a[r][Q];
      }
      if (bprof) blog[8 * (j >> 1) + 1] = wall_clock64();
      lds_barrier();
      if (bprof) blog[8 * (j >> 1) + 2] = wall_clock64();
      const double d0 = c0[j];
      const double rs0 = rsqrt_refined(d0), inv0 = rs0 * rs0;
      const double m01 = c0[j + 1];                                // entry (j + 1, j), unscaled
      const double d1 = fma(-(m01 * inv0), m01, c1[j + 1]);        // pivot of column j + 1 after the update by column j
      const double rs1 = rsqrt_refined(d1), inv1 = rs1 * rs1;
      if (tid == 0 && (!(d0 > 0.0) || !(d1 > 0.0))) bad = 1;
      double li0[R], lc0[R], li1[R], lc1[R];
#pragma unroll
      for (int r = Q; r < R; ++r) {
        const int i = ti + 16 * r, c = tj + 16 * r;
        const double ci0 = cj_or_zero(c0, i, N), cc0 = cj_or_zero(c0, c, N);
        li0[r] = i > j ? ci0 * inv0 : 0.0;  // rows / columns <= j are finished: zero operands leave them alone
        lc0[r] = c > j ? cc0 : 0.0;
        const double ci1 = fma(-li0[r], m01, cj_or_zero(c1, i, N)), cc1 = fma(-(cc0 * inv0), m01, cj_or_zero(c1, c, N));
        li1[r] = i > j + 1 ? ci1 * inv1 : 0.0;
        lc1[r] = c > j + 1 ? cc1 : 0.0;
      }
      if (bprof) blog[8 * (j >> 1) + 3] = wall_clock64();
#pragma unroll
      for (int r = Q; r < R; ++r)
#pragma unroll
        for (int q = Q; q < R; ++q) a[r][q] = fma(-li1[r], lc1[q], fma(-li0[r], lc0[q], a[r][q]));
      if (bprof) blog[8 * (j >> 1) + 4] = wall_clock64();
      if (tid == 0) invd[j] = rs0, invd[j + 1] = rs1, smem[j >> 1] = m01;  // (smem[0 .. 4 N): scalars of the pairs for the scaling pass)
      if (bprof) blog[8 * (j >> 1) + 5] = wall_clock64();
    }
    if (j_end == nb) break;  // (an odd last column is handled below)
  }
  for (; j < nb; ++j) {  // (odd number of unknowns: the last column on its own)
    double* cj = Lc + size_t(j) * ld;
    HS_PUBLISH_COLUMN(j, cj)
    lds_barrier();
    const double d = cj[j];
    if (tid == 0 && !(d > 0.0)) bad = 1;
    const double rs = rsqrt_refined(d), inv = rs * rs;
    double li[R], lc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = ti + 16 * r, c = tj + 16 * r;
      li[r] = i > j ? cj[i] * inv : 0.0;
      lc[r] = c > j ? cj[c] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < R; ++q) a[r][q] = fma(-li[r], lc[q], a[r][q]);
    if (tid == 0) invd[j] = rs;
  }
#undef HS_PUBLISH_COLUMN
  lds_barrier();
  // ---- scaling pass: l_ij = c_ij / l_jj; the second column of a pair was published before the update by the first:
  //      c'_(i, j+1) = c_(i, j+1) - c_(i, j) c_(j+1, j) / d_j ----
  for (int e = tid; e < ((nb + 1) / 2) * N; e += kBlock) {
    const int pj = 2 * (e / N), i = e % N;
    if (i >= n1 || i < pj) continue;
    const double rs0 = invd[pj], v0 = Lc[size_t(pj) * ld + i];
    if (pj + 1 < nb) {
      const double m01 = smem[pj >> 1], rs1 = invd[pj + 1], v1 = Lc[size_t(pj + 1) * ld + i];  // (entry (j + 1, j) as published: its slot in Lc is being scaled)
      if (i >= pj + 1) Lc[size_t(pj + 1) * ld + i] = fma(-(v0 * (rs0 * rs0)), m01, v1) * rs1;  // (i = pj + 1: d1 rs1 = l_(j+1, j+1))
    }
    Lc[size_t(pj) * ld + i] = v0 * rs0;  // (i = pj: d0 rs0 = l_jj)
  }
  lds_barrier();
  if (tid == 0 && bad) st->chol_failed = 1;
  if (tid >= 64) return;
  // ---- backward sweep, one wave: y_i in lanes i and i + 64, column j of L' = row j of L: l_ji = Lc[i][j], i < j ----
  const int lane = tid;
  double y0 = lane < nb ? Lc[size_t(lane) * ld + nb] : 0.0, y1 = lane + 64 < nb ? Lc[size_t(lane + 64) * ld + nb] : 0.0;
  double x0 = 0.0, x1 = 0.0;
  constexpr int U = 4;  // columns whose operands are in flight
  for (int j0 = nb - 1; j0 >= 0; j0 -= U) {
    double l0[U], l1[U], iv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 - u;
      l0[u] = (j >= 0 && lane < j) ? Lc[size_t(lane) * ld + j] : 0.0;
      l1[u] = (j >= 0 && lane + 64 < j) ? Lc[size_t(lane + 64) * ld + j] : 0.0;
      iv[u] = j >= 0 ? invd[j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = j0 - u;
      if (j < 0) break;
      const double ysel = j < 64 ? y0 : y1;  // (wave-uniform choice)
      const int jl = j & 63;
      const double yj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ysel), jl), __builtin_amdgcn_readlane(__double2loint(ysel), jl));
      const double xj = yj * iv[u];
      y0 = fma(-l0[u], xj, y0), y1 = fma(-l1[u], xj, y1);
      x0 = lane == j ? xj : x0, x1 = lane + 64 == j ? xj : x1;
    }
  }
  if (lane < nb) T.xb[lane] = x0;
  if (lane + 64 < nb) T.xb[lane + 64] = x1;
}

/// y' = y - Z x_b (one wave per row of Z, lanes over the border columns).
__global__ void __launch_bounds__(kBlock) k_border_apply(Tables T) {
  if (T.st->done) return;
  const int lane = threadIdx.x & 63, rho = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (rho >= T.np) return;
  double v = 0.0;
  for (int bq = lane; bq < T.nb; bq += 64) v = fma(T.Zb[size_t(rho) * T.nb + bq], T.xb[bq], v);
  v = wave_sum(v);
  if (lane == 0) *y_slot(T, rho) -= v;
}

}  // namespace hs
