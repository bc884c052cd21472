// kernels_common.hpp — shared device helpers: wave / workgroup reductions, batched strided loads, control-point staging (part of kernels.hpp; included once by capi.hip through it).
#pragma once
#include <type_traits>
#include "factors.hpp"
#include "device_primitives.hpp"

namespace hs {

constexpr int kBlock = 256;

/// Phase-timestamp hooks (wall_clock64 writes behind HS_DEBUG_FLAGS 16 / 32, read by tools/*_phase_timing.py) exist only in
/// profiling builds: with the default HS_PROFILE_HOOKS = 0 every `prof` predicate below is a compile-time false and the product
/// kernels carry no timing code.
#ifndef HS_PROFILE_HOOKS
#define HS_PROFILE_HOOKS 0
#endif
constexpr int kDenseLd = 256;  // leading dimension of the dense copy of a small reduced system (Tables::dense, kernels_dense_mx.hpp)
HSD bool prof_enabled(int debug_flags, int bit) { return HS_PROFILE_HOOKS && (debug_flags & bit); }

constexpr int kGatherFlag = 4 + 2 * 512 + 512 + 4 * kProgressStride;  // word of T.join_flag behind every other use (kBfFlagBase + 512 + 4 kProgressStride)

/// A bounded wait gave up: the solve ends here (every later kernel exits on `done`, hs_solve reports the reason) — nothing downstream may
/// consume what the workgroup that did not arrive has left half written. chol_failed = 2 is never cleared within a solve.
HSD void give_up(DevState* st) {
  st->chol_failed = 2;
  st->termination = HS_FAILURE;
  st->done = 1;
}


/// The value of lane ^ 1 / ^ 2 / ^ 4 through the DPP cross bar (vector moves; a general __shfl_xor is an LDS permute: ~80 cycles of issue per
/// 32-bit half where these are 4). xor 4 = row_shl:4 for the lanes with bit 2 clear, row_shr:4 for the others.
template <int CTRL>
HSD double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
HSD double lane_xor1(double v) { return dpp_move<0xB1>(v); }  // quad_perm [1 0 3 2]
HSD double lane_xor2(double v) { return dpp_move<0x4E>(v); }  // quad_perm [2 3 0 1]
HSD double lane_xor4(double v) {
  const double up = dpp_move<0x104>(v), down = dpp_move<0x114>(v);
  return (threadIdx.x & 4) ? down : up;
}

/// Sum over the wave, in every lane: a butterfly over lane ^ 32, 16, 8, 4, 2, 1 — the four lower levels on the DPP cross bar (xor 8 = row_ror:8
/// inside a row of 16 lanes), the two upper ones as LDS permutes. (Up to round 6 wave_sum was six __shfl_xor — twelve LDS permutes, ~0.25 us —
/// and this form, same pairs in the same order, i.e. bit-identical sums, was kept for the decision's five sums only: the 39 sums of
/// k_border_bb were 8 - 13 us of that kernel.)
HSD double wave_sum(double v) {
  v += __shfl_xor(v, 32);
  v += __shfl_xor(v, 16);
  v += dpp_move<0x128>(v);
  v += lane_xor4(v);
  v += lane_xor2(v);
  v += lane_xor1(v);
  return v;
}
HSD double wave_sum_fast(double v) { return wave_sum(v); }


/// Deterministic block sum (fixed butterfly inside each wave, waves combined in index order). Result valid on thread 0.
HSD double block_sum(double v, double* lds /* >= blockDim/64 */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) lds[w] = v;
  __syncthreads();
  double s = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < int(blockDim.x >> 6); ++i) s += lds[i];
  __syncthreads();
  return s;
}

/// N block sums at once (same order of additions as N calls of block_sum, two barriers instead of 2 N). Results valid on thread 0.
template <int N>
HSD void block_sum_n(double (&v)[N], double* lds /* >= N * blockDim / 64 */) {
  const int w = threadIdx.x >> 6, nw = int(blockDim.x >> 6);
#pragma unroll
  for (int e = 0; e < N; ++e) {
    v[e] = wave_sum_fast(v[e]);
    if ((threadIdx.x & 63) == 0) lds[e * nw + w] = v[e];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int e = 0; e < N; ++e) {
      double s = 0;
      for (int i = 0; i < nw; ++i) s += lds[e * nw + i];
      v[e] = s;
    }
  }
  __syncthreads();
}

/// Per-lane partial sum / max of a strided array with eight independent loads in flight (a plain `s += p[i]` loop keeps one
/// load in flight per lane and pays the full memory latency per element). Fixed order: bit-reproducible.
HSD double strided_sum(const double* __restrict__ p, int n, int stride = 1, int offset = 0) {
  double s = 0.0;
  for (int i0 = threadIdx.x; i0 < n; i0 += 8 * blockDim.x) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * blockDim.x;
      v[u] = i < n ? p[size_t(i) * stride + offset] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  return s;
}
template <int U = 8>
HSD double strided_max(const double* __restrict__ p, int n) {  // entries >= 0; U independent loads in flight per lane
  double m = 0.0;
  for (int i0 = threadIdx.x; i0 < n; i0 += U * blockDim.x) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * blockDim.x;
      v[u] = i < n ? p[i] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) m = fmax(m, v[u]);
  }
  return m;
}

HSD void stage_cps(const double* __restrict__ src, double* dst, int n_doubles) {
  // control points are n x 8 doubles: 16-byte pieces, four loads in flight per lane (one round trip for up to 128 control points
  // per 256 lanes instead of one per piece)
  const double2* s2 = reinterpret_cast<const double2*>(src);
  double2* d2 = reinterpret_cast<double2*>(dst);
  const int n2 = n_doubles / 2;
  for (int i0 = threadIdx.x; i0 < n2; i0 += 4 * blockDim.x) {
    double2 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      v[u] = i < n2 ? s2[i] : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      if (i < n2) d2[i] = v[u];
    }
  }
  __syncthreads();
}

}  // namespace hs
