// kernels_build.hpp — fused normal-equation build of the visual factors: linearisation, landmark elimination and the landmark-group-major
// Gram accumulation  J_p'J_p - Yh Yh'  in ONE pass, records in LDS only (part of kernels.hpp; included once by capi.hip through it).
//
// Replaces, for short feature tracks (bw (bw + 1) / 2 <= 256 window tiles), the sequence
//   k_linearize_visual (records -> HBM) -> k_landmark (records -> H_ll, b_l, W_l -> Y-hat) -> k_group_gram (-Yh Yh') + k_seg_gram (J_p'J_p)
// i.e. what Ceres does between CostFunction::Evaluate and its J'J / Schur assembly behind
// /root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:278. A residual's state block always lies inside the band window of its
// landmark — the 6 bw x 6 bw window anchored at the first control point the landmark touches — so BOTH Gram terms can be accumulated
// landmark-group-major by the workgroup that linearised the residual; no segment-major pass, no record ever leaves the CU.
//
// Work unit = CHUNK: a run of consecutive device landmarks that share their first control point cf (device order sorts landmarks by cf),
// at most L_max landmarks (W / Y-hat rows in LDS) and at most R residuals (one lane each). Sized so that TWO workgroups share a CU
// (<= 79 KB of LDS, <= 256 registers): every phase below is a gather through LDS, i.e. latency bound on a lone wave per SIMD (the first
// version ran one 133 KB workgroup per CU: 51 us per chunk, of which 16 us in phase 2 at ~630 clk per record visited).
// The unit of every gather is (landmark, control-point block) or (tile, record stream), and a record is stored where its consumers
// find it without searching: LDS slot = rank in the order (segment, landmark, table order), with the table pos[segment][landmark].
//   0  segment offset o = first control point - cf and landmark of every residual -> stable counting sort (ballots + a wave scan)
//   1  lane t linearises residual t into a COMPACT record at its sorted slot (factors.hpp: 8 + 7K doubles, the translation columns
//      -B_j A are rebuilt by the consumers);  H_t = A'A, b_t = A'r per record in landmark-major order
//   2  lane <-> (landmark, control point j'): the 6 x 3 block of W_l = sum J_p' J_l from the records of segments j' - K + 1 .. j' of that
//      landmark (pos table: only records that contribute are visited);  lane <-> (landmark, entry): H_ll, b_l
//   3  lane <-> (band tile (rb, cb), cb - rb < K; record stream s): P += J_p' J_p over the runs of the segments that cover both
//      blocks; the streams of a tile sit in adjacent lanes and are combined with butterfly steps (fixed order, no LDS). Streams per
//      tile: dealt per chunk from the record counts of its tiles (<= 64 band tiles; by the wave that idles during phase 1), a fixed
//      number per diagonal offset otherwise (build_streams)
//   4  lane <-> landmark: damped 3 x 3 Cholesky;  lane <-> (landmark, control point): Y-hat block -> LDS and HBM (k_backsub_retract)
//   5  lane <-> (window tile, landmark stream): Q -= Yh Yh', q -= Yh yh (two streams in adjacent lanes)
//   6  the chunk's partial [rows of P + Q | -Yh yh | J_p'r | diag J_p'J_p] -> HBM, summed over the chunks of the overlapping groups
//      by k_assemble in a fixed order: bit-reproducible, no floating-point atomics.
// HBM traffic per residual: 32 B of inputs + its share of Y-hat and of the chunk partial; the 448-byte record (k = 4) is never written.
#pragma once
#include "kernels_common.hpp"

namespace hs {

template <int K>
constexpr int build_rec_stride() { return compact_record<K>() + 2; }  // LDS record stride (16-byte aligned, bank-spread)

/// Band tiles of the chunk window that J_p'J_p touches: (rb, d = cb - rb), d < K, rb + d < bw; index = tiles of smaller d first.
__host__ __device__ inline int band_tile_count(int bw, int k) { return k * bw - k * (k - 1) / 2; }
HSD int band_tile_index(int rb, int d, int bw) { return d * bw - d * (d - 1) / 2 + rb; }
/// Record streams per band tile, by diagonal offset d (adjacent lanes of one wave, combined by butterflies). A tile of offset d sees the
/// records of K - d segments, so the diagonals get streams in proportion: greedy doubling of the most loaded diagonal (load (K - d) / ns[d])
/// while the lanes last, at most 8 streams, never more than the diagonal before it (so that groups stay aligned to their size when the
/// diagonals are laid out one after the other). K = 4, bw = 14: {8, 4, 4, 2} = 234 lanes, 5.5 - 8.3 records per lane where four streams for
/// every tile gave 2.8 - 11 — and the longest lane sets the phase.
struct BuildStreams {
  int ns[hsd::kMaxOrder], off[hsd::kMaxOrder + 1];  // streams per tile / first lane of diagonal d
};
__host__ __device__ inline BuildStreams build_streams(int bw, int k) {
  BuildStreams b;
  for (int d = 0; d < hsd::kMaxOrder; ++d) b.ns[d] = d < k ? 1 : 0;
  int used = band_tile_count(bw, k);
  bool stuck[hsd::kMaxOrder] = {false, false, false, false, false, false, false, false};
  for (;;) {
    int best = -1;
    for (int d = 0; d < k; ++d)  // most loaded diagonal that may still grow: compare (k - d) / ns[d] as cross products
      if (!stuck[d] && (best < 0 || (k - d) * b.ns[best] > (k - best) * b.ns[d])) best = d;
    if (best < 0) break;
    const bool fits = used + (bw - best) * b.ns[best] <= kBlock && 2 * b.ns[best] <= 8 && (best == 0 || 2 * b.ns[best] <= b.ns[best - 1]);
    if (!fits) {
      // the most loaded diagonal cannot grow any more: growing others would not shorten the phase
      bool any_heavier = false;
      for (int d = 0; d < k; ++d) any_heavier |= !stuck[d] && d != best && (k - d) * b.ns[best] >= (k - best) * b.ns[d];
      stuck[best] = true;
      if (!any_heavier) break;
      continue;
    }
    used += (bw - best) * b.ns[best];
    b.ns[best] *= 2;
  }
  b.off[0] = 0;
  for (int d = 0; d < hsd::kMaxOrder; ++d) b.off[d + 1] = b.off[d] + (d < k ? (bw - d) * b.ns[d] : 0);
  return b;
}
/// log2 of the stream counts, two bits per diagonal: what the kernel takes (Tables::build_stream_lg; computed by the host — as a table built
/// inside the kernel it lived in scratch memory, and the dependent scratch accesses of the loop above cost 7 us per workgroup)
inline int build_streams_packed(int bw, int k) {
  const BuildStreams b = build_streams(bw, k);
  int packed = 0;
  for (int d = 0; d < k; ++d) packed |= (b.ns[d] == 8 ? 3 : b.ns[d] == 4 ? 2 : b.ns[d] == 2 ? 1 : 0) << (2 * d);
  return packed;
}

constexpr int kLinv = 14;  // per landmark: 1/l00 l10 1/l11 l20 l21 1/l22 | s_l (3) | free flag | y-hat (3) | pad
constexpr int kBuildCams = 4;  // cameras staged in LDS (a residual of a later camera reads the table in HBM)

/// LDS layout of k_build_visual (offsets in doubles) — the host sizes the launch with the same function.
struct BuildLds {
  int rec, wy, cps, relp, lmp, cam, hb, linv, cpart, ints, total_doubles;
};
__host__ __device__ inline BuildLds build_lds_layout(int K, int bw, int R, int Lmax) {
  const int cs = 8 + 8 * K + 2, nband = band_tile_count(bw, K), nseg = bw - K + 1;
  BuildLds o;
  int off = 0;
  o.rec = off, off += (R * cs > nband * 42 ? R * cs : nband * 42);  // records | combined P tiles [nband][36 + 6] once phase 3 is over
  // W rows, then Y-hat rows [Lmax][6 bw][3]; before them (phase 2a): A'A (6), A'r (3), cost per record, landmark-major
  o.wy = off, off += (Lmax * 6 * bw * 3 > R * 10 ? Lmax * 6 * bw * 3 : R * 10);
  o.cps = off, off += 8 * bw;                // the window's control points
  o.relp = off, off += 8 * bw;               // relative rotations of consecutive control points (device_spline.hpp RelPre)
  o.lmp = off, off += 4 * Lmax;              // the chunk's landmark positions
  o.cam = off, off += 16 * kBuildCams;       // cameras
  o.hb = off, off += Lmax * 10;              // H_ll (6), b_l (3) per landmark
  o.linv = off, off += Lmax * kLinv;
  o.cpart = off, off += 8;                   // cost partials
  off += off & 1;
  o.ints = off;  // int tables (two per double): lp[Lmax + 1] ncp[Lmax] yoff[Lmax] seg_start[bw + 2] wave_cnt[4][bw] slot_lm[R] pos[nseg][Lmax + 1] frozen[bw] lane_tile[kBlock]
  off += (3 * Lmax + 1 + (bw + 2) + 4 * bw + R + nseg * (Lmax + 1) + bw + 1 + kBlock) / 2 + 1;  // (+ lane_tile[kBlock]: phase 3's lane -> (tile, streams) table)
  o.total_doubles = off;
  return o;
}

/// Operands of one record for a J_p'J_p band tile (blocks a, b of its state Jacobian) / for a W block (block jj): loaded as a unit so that
/// the loads of the NEXT record can be issued before the products of the current one (a lone wave per SIMD pays the full LDS latency per
/// record otherwise).
struct TileOps {
  double2 r, A0, A1, A2, a00, a01, a10, a11, b00, b01, b10, b11;
};
HSD TileOps tile_ops_load(const double* rec, int a, int b) {
  TileOps x;
  const double2* p = reinterpret_cast<const double2*>(rec);
  x.r = p[0], x.A0 = p[1], x.A1 = p[2], x.A2 = p[3];
  const double2* pa = reinterpret_cast<const double2*>(rec + 8 + 8 * a);
  const double2* pb = reinterpret_cast<const double2*>(rec + 8 + 8 * b);
  x.a00 = pa[0], x.a01 = pa[1], x.a10 = pa[2], x.a11 = pa[3];
  x.b00 = pb[0], x.b01 = pb[1], x.b10 = pb[2], x.b11 = pb[3];
  return x;
}
HSD void tile_ops_apply(const TileOps& x, bool diag, double* pacc, double* pg) {
  const double A[6] = {x.A0.x, x.A0.y, x.A1.x, x.A1.y, x.A2.x, x.A2.y};
  double ja[2][6], jb[2][6];
  ja[0][0] = x.a00.x, ja[0][1] = x.a00.y, ja[0][2] = x.a01.x, ja[1][0] = x.a10.x, ja[1][1] = x.a10.y, ja[1][2] = x.a11.x;
  jb[0][0] = x.b00.x, jb[0][1] = x.b00.y, jb[0][2] = x.b01.x, jb[1][0] = x.b10.x, jb[1][1] = x.b10.y, jb[1][2] = x.b11.x;
  const double na = -x.a01.y, nb = -x.b01.y;  // (both rows of a block carry the same B_eff)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) ja[i][3 + c] = na * A[3 * i + c], jb[i][3 + c] = nb * A[3 * i + c];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 6; ++c) pacc[6 * r + c] = fma(ja[0][r], jb[0][c], fma(ja[1][r], jb[1][c], pacc[6 * r + c]));
  if (diag) {
#pragma unroll
    for (int r = 0; r < 6; ++r) pg[r] = fma(ja[0][r], x.r.x, fma(ja[1][r], x.r.y, pg[r]));
  }
}
struct WOps {
  double2 A0, A1, A2, j00, j01, j10, j11;
};
HSD WOps w_ops_load(const double* rec, int jj) {
  WOps x;
  const double2* p = reinterpret_cast<const double2*>(rec);
  x.A0 = p[1], x.A1 = p[2], x.A2 = p[3];
  const double2* pj = reinterpret_cast<const double2*>(rec + 8 + 8 * jj);
  x.j00 = pj[0], x.j01 = pj[1], x.j10 = pj[2], x.j11 = pj[3];
  return x;
}
HSD void w_ops_apply(const WOps& x, double* wacc) {
  const double A[6] = {x.A0.x, x.A0.y, x.A1.x, x.A1.y, x.A2.x, x.A2.y};
  double jp[2][6];
  jp[0][0] = x.j00.x, jp[0][1] = x.j00.y, jp[0][2] = x.j01.x, jp[1][0] = x.j10.x, jp[1][1] = x.j10.y, jp[1][2] = x.j11.x;
  const double nb = -x.j01.y;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 3; ++c) jp[i][3 + c] = nb * A[3 * i + c];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) wacc[3 * r + c] = fma(jp[0][r], A[c], fma(jp[1][r], A[3 + c], wacc[3 * r + c]));
}

constexpr int kFoldFlag = 3;  // T.join_flag[kFoldFlag] = Tables::fold_epoch once the decision workgroup of a fold-mode k_build_visual has written the state
HSD void pack_decision_body(const Tables& T, int decide_here, double* red, unsigned* publish);  // kernels_update.hpp
/// Bounded wait for the decision workgroup (see wait_for_partner, kernels_factor.hpp): 2 s, then the solve is marked as failed and the caller
/// carries on. ONE lane of the workgroup polls, with relaxed loads: every lane of every chunk workgroup polling with acquire loads (75 000 lanes,
/// each poll an invalidation of the caches) delayed the decision workgroup itself — 0.261 instead of 0.204 ms per iteration at configs[1].
/// The flag word carries the outcome: (epoch << 2) | (done << 1) | accepted, handed to the other lanes through `slot` (LDS).
HSD unsigned fold_wait(const Tables& T, unsigned* slot) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    unsigned f;
    while (((f = __hip_atomic_load(T.join_flag + kFoldFlag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 2) < T.fold_epoch) {
      __builtin_amdgcn_s_sleep(2);
      if (wall_clock64() - t0 > 200000000ll) {
        give_up(T.st);                 // (done + HS_FAILURE: k_assemble and everything behind it exit)
        f = (T.fold_epoch << 2) | 2u;  // this workgroup carries on as if the decision had ended the solve
        break;
      }
    }
    *slot = f;
  }
  __syncthreads();
  return *slot;
}

template <int K>
__global__ void __launch_bounds__(kBlock, 2) k_build_visual(Tables T, int R, int Lmax, int robustify) {
  HS_DYNAMIC_LDS(smem);
  // Fold mode (Tables::fold_decision, visual-only windows on one shard, every iteration of a solve but the first): workgroup 0 of the launch
  // is the trust-region decision of the previous iteration — the work of k_pack_decision, which was not launched behind that iteration's
  // update (a launch boundary and a one-workgroup kernel, 11 us, off the chain) — and the chunk workgroups are 1 .. n. They request their
  // descriptor and the inputs of their residuals (tables that no decision changes), then wait for the decision's flag before they read the
  // solver state, the control points and the landmarks. Workgroup 0 is dispatched first and waits for nobody.
  const int fold = T.fold_decision;
  const int w = int(blockIdx.x) - fold, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  DevState* st = T.st;
  if (w < 0) {  // (also when an earlier iteration ended the solve: the chunk workgroups wait for the flag before they look)
    pack_decision_body(T, 3, smem, T.join_flag + kFoldFlag);
    return;
  }
  // the chunk descriptor and the solver state are requested together (the descriptor table is padded to the grid: always in bounds)
  const int4 d0 = *reinterpret_cast<const int4*>(T.ch_desc + 8 * w);
  const int4 d1 = *reinterpret_cast<const int4*>(T.ch_desc + 8 * w + 4);
  const int nres = d1.x;      // <= R (host: build_chunks)
  const int chunk_id = d1.y;  // slot of this chunk's partial (the descriptors are in dispatch order: order_chunks_for_dispatch)
  int st_done = 0, st_spec = 0, st_accepted = 0, st_ready = 0;
  double radius = 0.0;
  if (!fold) {
    st_done = st->done, st_spec = st->spec, st_accepted = st->accepted, st_ready = st->scaling_ready;
    radius = st->radius;
    if (st_done) return;
    if (w >= T.n_chunk) {  // padding workgroups of the visual section of the cost-partial table
      if (tid == 0) T.cost_part[w] = 0.0, T.ch_gmax[w] = 0.0;
      return;
    }
  }
  constexpr int CS = build_rec_stride<K>();
  const int bw = T.bw, R6 = 6 * bw, ntile = bw * (bw + 1) / 2, nband = band_tile_count(bw, K), nseg = bw - K + 1;
  const unsigned bw_magic = ((1u << 20) + bw - 1) / bw;  // tau / bw = (tau * magic) >> 20 for tau < 2^20 / bw (tasks: <= Lmax * bw)
  const BuildLds lay = build_lds_layout(K, bw, R, Lmax);
  double* recs = smem + lay.rec;
  double* Wy = smem + lay.wy;
  double* hbr = Wy;  // (phase 2a only)
  double* cps_l = smem + lay.cps;
  RelPre* relp = reinterpret_cast<RelPre*>(smem + lay.relp);
  double* lmp = smem + lay.lmp;
  double* cams = smem + lay.cam;
  double* Hb = smem + lay.hb;
  double* Linv = smem + lay.linv;
  double* cpart = smem + lay.cpart;
  int* lp = reinterpret_cast<int*>(smem + lay.ints);
  int* l_ncp = lp + Lmax + 1;
  int* l_yoff = l_ncp + Lmax;
  int* seg_start = l_yoff + Lmax;      // nseg + 1
  int* wave_cnt = seg_start + bw + 2;  // [4][bw]
  int* slot_lm = wave_cnt + 4 * bw;    // landmark of the record in sorted slot s
  int* pos = slot_lm + R;              // [nseg][Lmax + 1]: first slot of segment o whose landmark is >= l
  uint8_t* frozen = reinterpret_cast<uint8_t*>(pos + nseg * (Lmax + 1));  // constancy flags of the window's control points
  int* lane_tile = pos + nseg * (Lmax + 1) + (bw + 3) / 4;  // phase 3 (at most 64 band tiles): rb | d << 8 | log2(streams) << 12 of the lane's tile, -1: idle

  // phase timestamps (profiling builds only, HS_DEBUG_FLAGS 32; tools/build_phase_timing.py): lane 0 of every wave of the first 1024 chunks
  const bool bprof = prof_enabled(T.debug_flags, 32) && lane == 0 && w < 1023;  // (row 1023: the decision workgroup of a fold-mode launch, pack_decision_body)
  long long* blog = reinterpret_cast<long long*>(T.xpart) + 48 * 1024 + 64 * w + 16 * wave;
#define HS_BSTAMP(i) \
  if (bprof) blog[i] = wall_clock64()
  HS_BSTAMP(0);

  // ---- 0: every table of the chunk and every input of its residuals in ONE round of independent loads ----
  const int lo = d0.x, nl = d0.y, cf = d0.z, q0 = d0.w;
  // Record t of the chunk is linearised by lane t & 63 of wave 0 (t < 64) or wave 2: a chunk holds <= 128 records, and the hardware puts
  // waves {0, 2} of the two workgroups that share a CU on disjoint SIMD pairs (measured: wave -> SIMD is (0 2 1 3), (2 1 3 0), (3 0 2 1) or
  // (1 3 0 2), the co-resident workgroup one rotation on), where waves {0, 1} of both met on one SIMD and the linearisation — the phase in
  // which only the record lanes work — ran at half speed. (Wider chunks, HS_BUILD_R > 128: lane t.)
  const int rt = R > 128 ? tid : wave == 0 ? lane : wave == 2 ? 64 + lane : kBlock;
  const bool has_rec = rt < nres;
  int my_o = -1, my_l = 0;
  VisualIn in;
  int camid = 0;
  if (has_rec) {
    const int q = q0 + rt;
    in.first = T.v_first[q], my_l = T.v_lm[q] - lo;
    my_o = in.first - cf;
    const int info = T.v_info[q];
    in.type = info >> 16, camid = info & 0xffff;
    in.stamp = T.v_stamp[q];
    in.meas[0] = T.v_meas[3 * q], in.meas[1] = T.v_meas[3 * q + 1], in.meas[2] = T.v_meas[3 * q + 2];
  }
  // Fold mode: the decision of the previous iteration (workgroup 0 of this launch) is waited for as late as possible — behind the sort of
  // the record slots, which only needs the records' segments — with everything else already requested: the tables no decision changes go to
  // LDS, the point waits in registers in both versions (current / candidate of the previous iteration). Behind the flag word, which carries the
  // outcome, only the new radius is read (a coherent load, first used phases later: it was written by a workgroup on another XCD while this
  // one ran; no acquire fence — 1 200 waves invalidating their XCD's L2 for it cost as much as the launch this mode removes).
  double f_c0x = 0.0, f_c0y = 0.0, f_c1x = 0.0, f_c1y = 0.0;
  double f_l0x = 0.0, f_l0y = 0.0, f_l0z = 0.0, f_l1x = 0.0, f_l1y = 0.0, f_l1z = 0.0;  // (scalars: as an array the six doubles went to scratch memory)
  const int ncp_w = min(bw, T.sp.n_cp - cf);  // (4 ncp_w <= kBlock: at most one 16-byte piece of the window's control points per lane)
  if (fold) {
    st_spec = st->spec;  // (fixed for the solve: any cached copy is current)
    if (w >= T.n_chunk) {  // padding workgroups (zeros in a slot nobody reads once a solve has ended: no need to wait for the decision)
      if (tid == 0) T.cost_part[w] = 0.0, T.ch_gmax[w] = 0.0;
      return;
    }
    st_ready = 1;  // (a step has been computed: the scaling of this solve is fixed)
    if (tid < 4 * ncp_w) {
      const double2 c0 = reinterpret_cast<const double2*>(T.cp + 8 * cf)[tid], c1 = reinterpret_cast<const double2*>(T.cp_cand + 8 * cf)[tid];
      f_c0x = c0.x, f_c0y = c0.y, f_c1x = c1.x, f_c1y = c1.y;
    }
    if (tid < nl) {
      const double *l0 = T.lm + 3 * (lo + tid), *l1 = T.lm_cand + 3 * (lo + tid);
      f_l0x = l0[0], f_l0y = l0[1], f_l0z = l0[2], f_l1x = l1[0], f_l1y = l1[1], f_l1z = l1[2];
    }
  }
  // Deferred commit of the landmarks (DevState::spec == 4): the candidate accepted by the previous iteration is still only in lm_cand
  // (k_update_visual copies it on its way); the control points are committed by the decision and always current in T.cp — in fold mode the
  // decision workgroup copies an accepted candidate there WHILE this workgroup runs, so an accepted point is taken from cp_cand
  const bool fresh = !st_ready;
  {
    if (tid < ncp_w) frozen[tid] = T.cp_const[cf + tid];
    for (int e = tid; e < 16 * min(T.n_cam, kBuildCams); e += kBlock) cams[e] = T.cam[e];
    if (tid <= nl) lp[tid] = T.lm_ptr[lo + tid] - q0;
    if (tid < nl) {
      const int dl = lo + tid;
      l_ncp[tid] = T.lm_ncp[dl], l_yoff[tid] = T.lm_yoff[dl];
      double* li = Linv + kLinv * tid;
      li[6] = fresh ? 1.0 : T.lm_scale[3 * dl], li[7] = fresh ? 1.0 : T.lm_scale[3 * dl + 1], li[8] = fresh ? 1.0 : T.lm_scale[3 * dl + 2];
      li[9] = T.lm_const[dl] ? 0.0 : 1.0;
    }
    if (!fold) {
      const bool pend = st_spec == 4 && st_accepted;
      const double* lm_src = pend ? T.lm_cand : T.lm;
      const double2* s2 = reinterpret_cast<const double2*>(T.cp + 8 * cf);
      for (int e = tid; e < 4 * ncp_w; e += kBlock) reinterpret_cast<double2*>(cps_l)[e] = s2[e];
      if (tid < nl) lmp[4 * tid] = lm_src[3 * (lo + tid)], lmp[4 * tid + 1] = lm_src[3 * (lo + tid) + 1], lmp[4 * tid + 2] = lm_src[3 * (lo + tid) + 2];
    }
  }
  // stable counting sort of the record slots by (segment offset, landmark-major index): rank inside the wave from ballots
  int my_rank = 0;
  for (int o = 0; o < nseg; ++o) {
    const unsigned long long m = __ballot(my_o == o);
    if (my_o == o) my_rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave * bw + o] = __popcll(m);
  }
  __syncthreads();
  HS_BSTAMP(1);
  if (wave == 1 && !fold) {  // relative rotations of the window's consecutive control points: once per pair, not once per residual and pair
    if (lane + 1 < ncp_w) relp[lane] = rel_precompute(cps_l + 8 * lane, cps_l + 8 * lane + 8);
  }
  if (wave == 0) {  // seg_start[o] = records with a smaller segment offset: inclusive scan over the lanes of wave 0
    const int tot = lane < nseg ? wave_cnt[lane] + wave_cnt[bw + lane] + wave_cnt[2 * bw + lane] + wave_cnt[3 * bw + lane] : 0;
    int inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(inc, d);
      if (lane >= d) inc += up;
    }
    if (lane < nseg) seg_start[lane + 1] = inc;
    if (lane == 0) seg_start[0] = 0;
  }
  __syncthreads();
  int my_slot = 0;
  if (has_rec) {
    my_slot = seg_start[my_o] + my_rank;
    for (int ww = 0; ww < wave; ++ww) my_slot += wave_cnt[ww * bw + my_o];
    slot_lm[my_slot] = my_l;
  }
  for (int e = tid; e < nseg * (Lmax + 1); e += kBlock) pos[e] = seg_start[e / (Lmax + 1) + 1];  // default: the end of the segment's run
  if (fold) {  // (thread 0's poll runs next to the loop above; the barrier inside the wait is the one this phase ends with anyway)
    const unsigned f = fold_wait(T, reinterpret_cast<unsigned*>(cpart));
    // INVARIANT (what may be read behind the flag word, and how). The poll is relaxed, so nothing the decision workgroup wrote is ordered
    // behind it by the memory model; two things are read, and each has its own reason to be right:
    //   * the outcome (accepted / done) travels IN the flag word itself;
    //   * st->radius, with the agent-scope (sc1) load below: on gfx942 / gfx950 such a load is served by the coherent L2 path, and the
    //     decision workgroup wrote the radius in program order before its release store of the flag (pack_decision_body), which waits for
    //     the write to reach that level. First use of the value is phases later.
    // Everything else this workgroup reads (the point in both versions, the tables) comes from buffers the PREVIOUS kernel wrote and is
    // requested before the flag is looked at. A new field read here needs one of the two mechanisms above (or an acquire in thread 0 of
    // fold_wait, measured at 12 us per launch for one fence per wave) — a plain load would be a stale read waiting to happen. The CPU
    // emulation runs the decision workgroup first and cannot catch a violation.
    radius = __hip_atomic_load(&st->radius, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((f >> 1) & 1) return;  // the solve has ended (nothing has been written to memory yet)
    const bool acc = f & 1;    // (spec == 4: an accepted candidate's landmarks are still only in lm_cand)
    if (tid < 4 * ncp_w) reinterpret_cast<double2*>(cps_l)[tid] = make_double2(acc ? f_c1x : f_c0x, acc ? f_c1y : f_c0y);
    if (tid < nl) lmp[4 * tid] = acc ? f_l1x : f_l0x, lmp[4 * tid + 1] = acc ? f_l1y : f_l0y, lmp[4 * tid + 2] = acc ? f_l1z : f_l0z;
    __syncthreads();
    if (wave == 1 && lane + 1 < ncp_w) relp[lane] = rel_precompute(cps_l + 8 * lane, cps_l + 8 * lane + 8);
  }
  __syncthreads();
  HS_BSTAMP(2);
  // ---- streams of phase 3, per chunk (at most 64 band tiles: one lane of the last wave per tile, next to the linearisation, which that wave
  //      has no part in). A tile's lanes walk the records of the segments that cover both of its blocks, and the longest lane is the phase:
  //      with a FIXED number of streams per diagonal (build_streams: {8, 4, 4, 2} at K = 4, still the rule for wider bands) a chunk whose
  //      landmarks were first seen late in their tracks crowds its records into two or three segments, a tile of the outer diagonal walked 30
  //      - 60 of them on two lanes, and a quarter of the chunks of configs[1] took 35 - 38 us against a median of 30 (the slowest chunk is the
  //      kernel time). Here: streams = the power of two >= records / T, at most 16, for the smallest T whose lanes fit the workgroup; the
  //      tiles are placed by descending stream count, so every group is aligned to its size (the butterflies below need that). ----
  const bool dyn = nband <= 64 && !(HS_PROFILE_HOOKS && T.debug_flags < 0);  // (A/B in profiling builds, HS_DEBUG_FLAGS sign bit: the fixed rule)
  if (dyn && wave == 3) {
    int t_rb = lane, t_d = 0;
#pragma unroll
    for (int d = 0; d + 1 < K; ++d)
      if (t_d == d && t_rb >= bw - d) t_rb -= bw - d, t_d = d + 1;
    const bool t_ok = lane < nband;
    int cnt = 0;
    if (t_ok) {
      const int o_hi = min(t_rb, nseg - 1), o_lo = max(0, t_rb + t_d - K + 1);
      if (o_hi >= o_lo) cnt = seg_start[o_hi + 1] - seg_start[o_lo];
    }
    constexpr int kCand[14] = {2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 24, 32, 64, 256};  // records per lane, tried in this order (cnt <= R <= 256: the last one always fits)
    int lg_sel = 0;
    bool found = false;
#pragma unroll
    for (int c = 0; c < 14; ++c) {
      const int q = (cnt + kCand[c] - 1) / kCand[c];
      const int lg = (q > 1) + (q > 2) + (q > 4) + (q > 8);
      const int lanes = __popcll(__ballot(t_ok && lg == 0)) + 2 * __popcll(__ballot(t_ok && lg == 1)) + 4 * __popcll(__ballot(t_ok && lg == 2)) +
                        8 * __popcll(__ballot(t_ok && lg == 3)) + 16 * __popcll(__ballot(t_ok && lg == 4));
      if (!found && lanes <= kBlock) found = true, lg_sel = lg;
    }
    int base = 0, my_off = 0;
#pragma unroll
    for (int c = 4; c >= 0; --c) {
      const unsigned long long m = __ballot(t_ok && lg_sel == c);
      if (t_ok && lg_sel == c) my_off = base + (__popcll(m & ((1ull << lane) - 1ull)) << c);
      base += __popcll(m) << c;
    }
#pragma unroll
    for (int j = 0; j < kBlock / 64; ++j) lane_tile[64 * j + lane] = -1;
    wait_lds();  // (one wave: its LDS writes land in order; the emulation needs the hand-over between its lane threads)
    if (t_ok)
      for (int e = 0; e < (1 << lg_sel); ++e) lane_tile[my_off + e] = t_rb | (t_d << 8) | (lg_sel << 12);
  }
  // ---- 1: linearise into the sorted slot (no global loads from here to phase 4); pos[o][l'] = my slot for the landmarks l' between my
  //         predecessor's (exclusive) and mine ----
  if (has_rec) {
    const int l_prev = my_slot > seg_start[my_o] ? slot_lm[my_slot - 1] : -1;
    for (int l = l_prev + 1; l <= my_l; ++l) pos[my_o * (Lmax + 1) + l] = my_slot;
    double* rec = recs + my_slot * CS;
    in.cam = camid < kBuildCams ? cams + 16 * camid : T.cam + kCamStride * camid;
    in.lm[0] = lmp[4 * my_l], in.lm[1] = lmp[4 * my_l + 1], in.lm[2] = lmp[4 * my_l + 2];
    // (cps_l / frozen are indexed by absolute control point)
    const double cost = visual_linearize_compact<K>(T, cps_l - 8 * cf, relp - cf, frozen - cf, in, robustify != 0, rec);
    // A'A, A'r and the cost of this record (the landmark's H_ll and b_l are their sums in table order)
    const double a0 = rec[2], a1 = rec[3], a2 = rec[4], a3 = rec[5], a4 = rec[6], a5 = rec[7], r0 = rec[0], r1 = rec[1];
    double* h = hbr + 10 * rt;
    h[0] = fma(a0, a0, a3 * a3), h[1] = fma(a0, a1, a3 * a4), h[2] = fma(a0, a2, a3 * a5);
    h[3] = fma(a1, a1, a4 * a4), h[4] = fma(a1, a2, a4 * a5), h[5] = fma(a2, a2, a5 * a5);
    h[6] = fma(a0, r0, a3 * r1), h[7] = fma(a1, r0, a4 * r1), h[8] = fma(a2, r0, a5 * r1);
    h[9] = cost;
  }
  HS_BSTAMP(3);
  __syncthreads();
  HS_BSTAMP(4);
  // ---- 2a: H_ll / b_l (lane <-> landmark, entry); cost partials (8 lanes, fixed order) ----
  if (tid < 9 * nl) {
    const int l = tid / 9, e = tid - 9 * l;
    double acc = 0.0;
    for (int t = lp[l]; t < lp[l + 1]; ++t) acc += hbr[10 * t + e];
    Hb[10 * l + e] = acc;
  } else if (tid >= kBlock - 8) {
    const int g = tid - (kBlock - 8), per = (nres + 7) / 8;
    double acc = 0.0;
    for (int t = g * per; t < min(nres, (g + 1) * per); ++t) acc += hbr[10 * t + 9];
    cpart[g] = acc;
  }
  __syncthreads();  // the W rows take the place of the per-record sums
  // ---- 2b: W_l blocks, lane <-> (landmark, control point): ONE loop over the records that contribute (runs of K segments) ----
  const int n_task_w = nl * bw;
  for (int tau = tid; tau < n_task_w; tau += kBlock) {
    const int l = int((unsigned(tau) * bw_magic) >> 20), jb = tau - l * bw;
    double wacc[18];
#pragma unroll
    for (int e = 0; e < 18; ++e) wacc[e] = 0.0;
    // run of (segment jb - jj, landmark l), requested for all K segments at once; cum[jj] = records in the runs 0 .. jj
    // (a block beyond the landmark's control points has no records: its rows are written as zeros, which is what lets phase 5 walk every
    //  landmark without masking its operands)
    int run_lo[K], cum[K];
    int total = 0;
    const bool in_rows = jb < l_ncp[l];
#pragma unroll
    for (int jj = 0; jj < K; ++jj) {
      const int o = jb - jj;
      const bool ok = in_rows && o >= 0 && o < nseg;
      const int* pp = pos + (ok ? o : 0) * (Lmax + 1) + l;
      const int a = pp[0], b = pp[1];
      run_lo[jj] = a, total += ok ? b - a : 0, cum[jj] = total;
    }
    // h-th contributing record: run jj, slot s. Two operand sets alternate: the loads of record h + 1 are in flight during the products
    // of record h, and nothing is copied between them.
#define HS_W_LOAD(h_, dst)                                                   \
  {                                                                          \
    int jj_ = 0, s_ = run_lo[0] + (h_);                                      \
    _Pragma("unroll") for (int x = 1; x < K; ++x) if ((h_) >= cum[x - 1]) jj_ = x, s_ = run_lo[x] + (h_)-cum[x - 1]; \
    dst = w_ops_load(recs + s_ * CS, jj_);                                   \
  }
    WOps opa, opb;
    if (total > 0) HS_W_LOAD(0, opa);
    for (int h = 0; h < total; h += 2) {
      if (h + 1 < total) HS_W_LOAD(h + 1, opb);
      w_ops_apply(opa, wacc);
      if (h + 1 < total) {
        if (h + 2 < total) HS_W_LOAD(h + 2, opa);
        w_ops_apply(opb, wacc);
      }
    }
#undef HS_W_LOAD
    double* wr = Wy + (size_t(l) * R6 + 6 * jb) * 3;
#pragma unroll
    for (int e = 0; e < 18; e += 2) *reinterpret_cast<double2*>(wr + e) = make_double2(wacc[e], wacc[e + 1]);
  }
  HS_BSTAMP(5);
  // ---- 3: J_p'J_p band tiles: lane group g = tid / NS serves tile (g mod 16) * 4 + g / 16 (NS = 4: every wave gets tiles of every
  //         diagonal offset — the diagonal tiles see K segments, the outermost one), stream = tid mod NS ----
  int p_d = 0, p_off = 0, ns_lg = 0, p_end = 0;  // diagonal of this lane's tile, first lane of that diagonal, log2 of its stream count
#pragma unroll
  for (int d = 0; d < K; ++d) {  // (no tables: registers only)
    const int lg = (T.build_stream_lg >> (2 * d)) & 3, next = p_end + ((bw - d) << lg);
    if (tid >= p_end && tid < next) p_d = d, p_off = p_end, ns_lg = lg;
    p_end = next;
  }
  bool p_lane = tid < p_end;
  int p_rb = p_lane ? (tid - p_off) >> ns_lg : 0;
  if (dyn) {  // (the table of this chunk, written by the last wave in front of the barrier behind phase 1)
    const int lt = lane_tile[tid];
    p_lane = lt >= 0;
    p_rb = p_lane ? lt & 255 : 0, p_d = p_lane ? (lt >> 8) & 15 : 0, ns_lg = p_lane ? (lt >> 12) & 7 : 0, p_off = 0;  // (groups are aligned to their size)
  }
  const int NS = 1 << ns_lg, ns0 = 1 << (T.build_stream_lg & 3);  // (ns0: the widest groups of the fixed rule, diagonal 0)
  const int p_s = (tid - p_off) & (NS - 1);
  // butterfly levels this wave needs (wave-uniform)
  const bool lv2 = dyn ? __ballot(ns_lg > 1) != 0 : ns0 > 2, lv4 = dyn ? __ballot(ns_lg > 2) != 0 : ns0 > 4, lv8 = dyn && __ballot(ns_lg > 3) != 0;
  const int p_tb = band_tile_index(p_rb, p_d, bw);
  double pacc[36], pg[6];
#pragma unroll
  for (int e = 0; e < 36; ++e) pacc[e] = 0.0;
#pragma unroll
  for (int e = 0; e < 6; ++e) pg[e] = 0.0;
  if (p_lane) {
    // the runs of the <= K segments that cover both blocks of the tile, this stream's share of each; one loop over all of them
    const int o_hi = min(p_rb, nseg - 1);
    int run_lo[K], cum[K];
    int total = 0;
#pragma unroll
    for (int x = 0; x < K; ++x) {
      const int o = o_hi - x;  // block a = p_rb - o = (p_rb - o_hi) + x
      const bool ok = o >= 0 && o >= p_rb + p_d - K + 1;
      const int a0 = seg_start[ok ? o : 0], a1 = seg_start[(ok ? o : 0) + 1];
      run_lo[x] = a0 + p_s;
      total += ok && a1 > a0 + p_s ? (a1 - a0 - p_s + NS - 1) >> ns_lg : 0, cum[x] = total;
    }
    const int a_base = p_rb - o_hi;
#define HS_T_LOAD(h_, dst)                                                   \
  {                                                                          \
    int x_ = 0, s_ = run_lo[0] + ((h_) << ns_lg);                            \
    _Pragma("unroll") for (int x = 1; x < K; ++x) if ((h_) >= cum[x - 1]) x_ = x, s_ = run_lo[x] + (((h_)-cum[x - 1]) << ns_lg); \
    dst = tile_ops_load(recs + s_ * CS, a_base + x_, a_base + x_ + p_d);     \
  }
    TileOps opa, opb;  // two operand sets alternate (no copies): record h + 1 is in flight during the products of record h
    if (total > 0) HS_T_LOAD(0, opa);
    for (int h = 0; h < total; h += 2) {
      if (h + 1 < total) HS_T_LOAD(h + 1, opb);
      tile_ops_apply(opa, p_d == 0, pacc, pg);
      if (h + 1 < total) {
        if (h + 2 < total) HS_T_LOAD(h + 2, opa);
        tile_ops_apply(opb, p_d == 0, pacc, pg);
      }
    }
#undef HS_T_LOAD
  }
  // the streams of a tile are adjacent lanes: butterfly sums, ((s0 + s1) + (s2 + s3)) + ... on every lane of the group; the groups of a wave
  // have different sizes, so a level is exchanged by every lane and added by the lanes whose group reaches that far
  {
    const bool t1 = NS > 1, t2 = NS > 2, t4 = NS > 4, t8 = NS > 8;
#pragma unroll
    for (int e = 0; e < 36; ++e) {
      const double o = lane_xor1(pacc[e]);
      pacc[e] += t1 ? o : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const double o = lane_xor1(pg[e]);
      pg[e] += t1 ? o : 0.0;
    }
    if (lv2) {
#pragma unroll
      for (int e = 0; e < 36; ++e) {
        const double o = lane_xor2(pacc[e]);
        pacc[e] += t2 ? o : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const double o = lane_xor2(pg[e]);
        pg[e] += t2 ? o : 0.0;
      }
    }
    if (lv4) {
#pragma unroll
      for (int e = 0; e < 36; ++e) {
        const double o = lane_xor4(pacc[e]);
        pacc[e] += t4 ? o : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const double o = lane_xor4(pg[e]);
        pg[e] += t4 ? o : 0.0;
      }
    }
    if (lv8) {  // (sixteen streams: lane ^ 8 = row_ror:8 inside the row of sixteen lanes)
#pragma unroll
      for (int e = 0; e < 36; ++e) {
        const double o = dpp_move<0x128>(pacc[e]);
        pacc[e] += t8 ? o : 0.0;
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const double o = dpp_move<0x128>(pg[e]);
        pg[e] += t8 ? o : 0.0;
      }
    }
  }
  HS_BSTAMP(6);
  __syncthreads();  // everybody is done with the records: the combined P tiles take their place
  HS_BSTAMP(7);
  double* Pc = recs;  // [nband][42]
  if (p_lane && p_s == 0) {
    double* dst = Pc + size_t(p_tb) * 42;
#pragma unroll
    for (int e = 0; e < 36; e += 2) *reinterpret_cast<double2*>(dst + e) = make_double2(pacc[e], pacc[e + 1]);
#pragma unroll
    for (int e = 0; e < 6; e += 2) *reinterpret_cast<double2*>(dst + 36 + e) = make_double2(pg[e], pg[e + 1]);
  }
  // ---- 4a: V = S_l H_ll S_l + D_l^2 = L L' per landmark (landmark_finish of k_landmark); the last wave, which holds the fewest tiles ----
  double lane_gmax = 0.0;  // (last wave: max |b_l| of this lane's landmark; reduced over the chunk below)
  if (tid >= kBlock - 64 && tid - (kBlock - 64) < nl) {
    const int l = tid - (kBlock - 64), dl = lo + l;
    double* li = Linv + kLinv * l;
    const double lmf = li[9];  // 0: constant landmark (J_l = 0)
    const double* h = Hb + 10 * l;
    const double h0 = lmf * h[0], h1 = lmf * h[1], h2 = lmf * h[2], h3 = lmf * h[3], h4 = lmf * h[4], h5 = lmf * h[5];
    const double b0 = lmf * h[6], b1 = lmf * h[7], b2 = lmf * h[8];
    double sl0, sl1, sl2;
    if (fresh)
      sl0 = 1.0 / (1.0 + sqrt(h0)), sl1 = 1.0 / (1.0 + sqrt(h3)), sl2 = 1.0 / (1.0 + sqrt(h5));
    else
      sl0 = li[6], sl1 = li[7], sl2 = li[8];
    double v00 = sl0 * sl0 * h0, v01 = sl0 * sl1 * h1, v02 = sl0 * sl2 * h2;
    double v11 = sl1 * sl1 * h3, v12 = sl1 * sl2 * h4, v22 = sl2 * sl2 * h5;
    const double inv_radius = 1.0 / radius;
    const double d0 = fmin(fmax(v00, 1e-6), 1e32) * inv_radius, d1 = fmin(fmax(v11, 1e-6), 1e32) * inv_radius, d2 = fmin(fmax(v22, 1e-6), 1e32) * inv_radius;
    v00 += d0, v11 += d1, v22 += d2;
    auto rsqrt_refined = [](double d) {  // hardware estimate + one third-order correction: full double accuracy, no divide / sqrt sequence
      const double y = __builtin_amdgcn_rsq(d);
      const double e = fma(-d * y, y, 1.0);
      return fma(y * e, fma(0.375, e, 0.5), y);
    };
    const double i00 = rsqrt_refined(v00), l00 = v00 * i00, l10 = v01 * i00, l20 = v02 * i00;
    const double p11 = v11 - l10 * l10, i11 = rsqrt_refined(p11), l11 = p11 * i11, l21 = (v12 - l20 * l10) * i11;
    const double p22 = v22 - l20 * l20 - l21 * l21, i22 = rsqrt_refined(p22), l22 = p22 * i22;
    const bool active = lmf != 0.0;
    const double sb0 = sl0 * b0, sb1 = sl1 * b1, sb2 = sl2 * b2;
    const double y0 = sb0 * i00, y1 = (sb1 - l10 * y0) * i11, y2 = (sb2 - l20 * y0 - l21 * y1) * i22;
    const double yy0 = active ? y0 : 0.0, yy1 = active ? y1 : 0.0, yy2 = active ? y2 : 0.0;
    li[0] = i00, li[1] = l10, li[2] = i11, li[3] = l20, li[4] = l21, li[5] = i22;
    li[6] = sl0, li[7] = sl1, li[8] = sl2;
    li[10] = yy0, li[11] = yy1, li[12] = yy2;
    double* L = T.lm_L + 6 * dl;
    L[0] = l00, L[1] = l10, L[2] = l11, L[3] = l20, L[4] = l21, L[5] = l22;
    T.lm_yhat[3 * dl] = yy0, T.lm_yhat[3 * dl + 1] = yy1, T.lm_yhat[3 * dl + 2] = yy2;
    T.lm_sb[3 * dl] = sb0, T.lm_sb[3 * dl + 1] = sb1, T.lm_sb[3 * dl + 2] = sb2;
    T.lm_D2[3 * dl] = d0, T.lm_D2[3 * dl + 1] = d1, T.lm_D2[3 * dl + 2] = d2;
    lane_gmax = active ? fmax(fabs(b0), fmax(fabs(b1), fabs(b2))) : 0.0;
    T.lm_gmax[dl] = lane_gmax;
    if (fresh) T.lm_scale[3 * dl] = sl0, T.lm_scale[3 * dl + 1] = sl1, T.lm_scale[3 * dl + 2] = sl2;
  }
  if (tid >= kBlock - 64) {  // the chunk's landmark-side gradient max norm (one value per chunk for the bookkeeping of k_band_factor_la)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lane_gmax = fmax(lane_gmax, __shfl_xor(lane_gmax, o));
    if (tid == kBlock - 64) T.ch_gmax[w] = lane_gmax;
  }
  __syncthreads();
  HS_BSTAMP(8);
  // ---- 4b: Y-hat = W S_l L^-T, one 6 x 3 block per lane ----
  for (int tau = tid; tau < n_task_w; tau += kBlock) {
    const int l = int((unsigned(tau) * bw_magic) >> 20), jb = tau - l * bw;
    if (jb >= l_ncp[l]) continue;
    const double* li = Linv + kLinv * l;
    const double i00 = li[0], l10 = li[1], i11 = li[2], l20 = li[3], l21 = li[4], i22 = li[5];
    const double s0 = li[6] * li[9], s1 = li[7] * li[9], s2 = li[8] * li[9];  // (constant landmark: zero rows)
    double* wr = Wy + (size_t(l) * R6 + 6 * jb) * 3;
    double y[18];
#pragma unroll
    for (int e = 0; e < 18; e += 2) {
      const double2 v = *reinterpret_cast<const double2*>(wr + e);
      y[e] = v.x, y[e + 1] = v.y;
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double w0 = y[3 * r] * s0, w1 = y[3 * r + 1] * s1, w2 = y[3 * r + 2] * s2;
      const double a0 = w0 * i00, a1 = (w1 - a0 * l10) * i11, a2 = (w2 - a0 * l20 - a1 * l21) * i22;
      y[3 * r] = a0, y[3 * r + 1] = a1, y[3 * r + 2] = a2;
    }
    double* Y = T.Y + l_yoff[l] + 18 * jb;
#pragma unroll
    for (int e = 0; e < 18; e += 2) {
      *reinterpret_cast<double2*>(wr + e) = make_double2(y[e], y[e + 1]);
      *reinterpret_cast<double2*>(Y + e) = make_double2(y[e], y[e + 1]);
    }
    if (T.wide_q) {  // the copy k_landmark_gram_wide reads with consecutive landmarks in consecutive lanes
      double2* Yt = reinterpret_cast<double2*>(T.Yt) + size_t(cf + jb) * 9 * T.yt_stride + (lo + l);
#pragma unroll
      for (int e = 0; e < 9; ++e) Yt[size_t(e) * T.yt_stride] = make_double2(y[2 * e], y[2 * e + 1]);
    }
  }
  __syncthreads();
  HS_BSTAMP(9);
  // ---- 5: Q = - sum_l Yh_l Yh_l' over the window tiles, q = - sum_l Yh_l yh_l (diagonal tiles): lane = QS * tile + stream.
  //         Branch-free over the landmarks (a tile outside a landmark's rows reads zero rows), two landmarks in flight.
  //         Window-wide bands (long feature tracks: more tiles than lanes — 561 for 33 control points) take the tiles in passes of kBlock. ----
  const int QS = ntile <= kBlock / 2 ? 2 : 1;
  double* G = T.grpQ + size_t(chunk_id) * (size_t(ntile) * 36 + 3 * R6);
  for (int q_base = 0; q_base < ntile; q_base += kBlock / QS) {
    const int q_t = q_base + tid / QS, q_s = tid % QS;
    const bool q_ok = q_t < ntile;
    int q_rb = 0, q_cb = 0;
    {
      int rem = q_ok ? q_t : 0;
      while (rem >= bw - q_rb) rem -= bw - q_rb, ++q_rb;  // row rb holds bw - rb tiles
      q_cb = q_rb + rem;
    }
    double acc[36], qacc[6];
#pragma unroll
    for (int e = 0; e < 36; ++e) acc[e] = 0.0;
#pragma unroll
    for (int e = 0; e < 6; ++e) qacc[e] = 0.0;
    if (q_ok && !T.wide_q) {  // (wide_q: the landmark term is formed once per window from the Y-hat rows in HBM, k_landmark_gram_wide)
      const bool diag = q_rb == q_cb;
#pragma unroll 2
      for (int l = q_s; l < nl; l += QS) {
        const double* Yb = Wy + size_t(l) * R6 * 3;  // (rows beyond the landmark's control points are zeros: phase 2b)
        double A[18], B[18];
#pragma unroll
        for (int e = 0; e < 18; e += 2) {
          const double2 va = *reinterpret_cast<const double2*>(Yb + 18 * q_rb + e), vb = *reinterpret_cast<const double2*>(Yb + 18 * q_cb + e);
          A[e] = va.x, A[e + 1] = va.y, B[e] = vb.x, B[e + 1] = vb.y;
        }
        const double* li = Linv + kLinv * l;
        const double y0 = li[10], y1 = li[11], y2 = li[12];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
#pragma unroll
          for (int c = 0; c < 6; ++c)
            acc[6 * r + c] = fma(-A[3 * r + 2], B[3 * c + 2], fma(-A[3 * r + 1], B[3 * c + 1], fma(-A[3 * r], B[3 * c], acc[6 * r + c])));
          if (diag) qacc[r] = fma(-A[3 * r + 2], y2, fma(-A[3 * r + 1], y1, fma(-A[3 * r], y0, qacc[r])));
        }
      }
    }
    if (QS == 2) {  // stream 0 + stream 1 (adjacent lanes)
#pragma unroll
      for (int e = 0; e < 36; ++e) acc[e] += lane_xor1(acc[e]);
#pragma unroll
      for (int e = 0; e < 6; ++e) qacc[e] += lane_xor1(qacc[e]);
    }
    HS_BSTAMP(10);
    // ---- 6: P + Q and the three vectors of the chunk partial -> HBM ----
    if (q_s == 0 && q_ok && (!T.wide_q || q_cb - q_rb < K)) {  // (wide_q: the band tiles of J_p'J_p are all a partial holds)
      const int d = q_cb - q_rb;
      if (d < K) {
        const double* src = Pc + size_t(band_tile_index(q_rb, d, bw)) * 42;
        if (d == 0) {
#pragma unroll
          for (int r = 0; r < 6; ++r) {
            G[size_t(ntile) * 36 + 6 * q_rb + r] = qacc[r];
            G[size_t(ntile) * 36 + R6 + 6 * q_rb + r] = src[36 + r];
            G[size_t(ntile) * 36 + 2 * R6 + 6 * q_rb + r] = src[7 * r];
          }
        }
#pragma unroll
        for (int e = 0; e < 36; e += 2) {
          const double2 v = *reinterpret_cast<const double2*>(src + e);
          acc[e] += v.x, acc[e + 1] += v.y;
        }
      }
      // rows of the window, not tiles: row (rb, r) holds its (bw - rb) blocks contiguously, so that k_assemble — one lane per band entry of a
      // scalar row — reads a chunk partial with consecutive lanes on consecutive doubles
      double* grow = G + 36 * group_tile_index(q_rb, q_rb, bw) + 6 * (q_cb - q_rb);
      const int rstride = 6 * (bw - q_rb);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int e = 0; e < 6; e += 2) *reinterpret_cast<double2*>(grow + r * rstride + e) = make_double2(acc[6 * r + e], acc[6 * r + e + 1]);
    }
  }
  HS_BSTAMP(11);
  if (tid == 0) {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += cpart[g];
    T.cost_part[w] = s;
  }
  HS_BSTAMP(12);
  if (bprof) blog[13] = nres, blog[14] = nl | (cf << 16), blog[15] = (long long)(__builtin_amdgcn_s_getreg(63492)) | ((long long)(__builtin_amdgcn_s_getreg(63508)) << 32);  // HW_ID | XCC_ID
#undef HS_BSTAMP
}

// ---------------------------------------------------------------------------------------------------------------------
// Candidate point and its cost, per chunk (the fused path's k_backsub_retract + k_cost_visual in one launch). Workgroups
// [0, n_vis_parts): chunk w — the candidate control points of its window (Plus(x, delta), recomputed per workgroup: bw quaternion
// products), the landmark back-substitution of its landmarks
//   y_l = L^-T (yh_l - Yh_l' (Sp o y_p)),  step_l = -y_l,  candidate = lm + S_l o step_l        (lane <-> (landmark, control point) partial
//   dot products, summed per landmark in a fixed order; the landmark-side terms of the decision per chunk),
// and the value-only cost of its residual blocks at that candidate. Workgroups behind them: the candidate of every replicated unknown
// (control points, bias points, gravity) and its norms — k_backsub_retract's second half — and the landmarks no chunk holds (unobserved).
// The current control points are always in T.cp (the decision kernel commits them: 8 n_cp doubles); an accepted landmark candidate stays in
// lm_cand until this kernel passes over it (DevState::spec == 4).
// ---------------------------------------------------------------------------------------------------------------------
__host__ __device__ inline int update_lds_doubles(int bw, int R, int Lmax) {
  return 14 * bw + 4 * Lmax * bw + 8 * Lmax + 16 * kBuildCams + R + 8 + Lmax + 8 + 8 * bw;  // (.. + l_ncp, l_yoff: 2 Lmax ints; + relative rotations of the candidate window)
}

/// Candidate of a replicated unknown, y = Plus(x, delta): ONE spelling for the workgroups that write the candidate tables and for those that
/// recompute it for themselves (the cost workgroups of the prior / inertial factors in the same launch, which cannot wait for the former).
HSD void candidate_control_point(const Tables& T, int j, double* y) {
  const double* x = T.cp + 8 * j;
  const double* d = T.delta_p + 6 * j;
  const Quat q = quat_plus(Quat{x[0], x[1], x[2], x[3]}, V3{d[0], d[1], d[2]});
  y[0] = q.x, y[1] = q.y, y[2] = q.z, y[3] = q.w;
  y[4] = x[4] + d[3], y[5] = x[5] + d[4], y[6] = x[6] + d[5];
  y[7] = x[7];
}
HSD void candidate_bias_point(const Tables& T, int b /* < 2 n_bias: gyroscope, then accelerometer */, double* y) {
  const bool acc = b >= T.n_bias;
  const double* x = (acc ? T.bias_a : T.bias_g) + 4 * (acc ? b - T.n_bias : b);
  const double* d = T.delta_b + 3 * b;
  y[0] = x[0] + d[0], y[1] = x[1] + d[1], y[2] = x[2] + d[2], y[3] = x[3];
}

/// Cost workgroups of the prior / inertial factors inside k_update_visual's launch (single shard; k_cost_prior / k_cost_inertial otherwise: two
/// more launches on the chain of an iteration, ~8 us each and almost all of it launch + first-load latency). blk: index behind the norm
/// workgroups — [0, nb_pri) prior, then inertial (kInertialBlock residuals per workgroup, first wave). The candidate point is recomputed into LDS.
template <int K>
HSD void update_cost_workgroup(const Tables& T, int blk, int nb_pri, int n_vis_parts, double* smem) {
  const int tid = threadIdx.x;
  __shared__ double red[kBlock / 64];
  double* cps = smem;
  double* bg = cps + 8 * T.sp.n_cp;
  double* ba = bg + 4 * T.n_bias;
  double* grav = ba + 4 * T.n_bias;
  for (int j = tid; j < T.sp.n_cp; j += kBlock) candidate_control_point(T, j, cps + 8 * j);
  if (T.nb > 0) {
    for (int b = tid; b < 2 * T.n_bias; b += kBlock) candidate_bias_point(T, b, bg + 4 * b);  // (ba follows bg: b >= n_bias lands there)
    if (tid == 0) sphere_plus(T.gravity, T.delta_b + 6 * T.n_bias, grav);
  }
  __syncthreads();
  double cost = 0.0;
  if (blk < nb_pri) {
    const int i = blk * kBlock + tid;
    if (i < T.n_pri) cost = prior_cost<K>(T, cps, i);
  } else {
    const int i = (blk - nb_pri) * kInertialBlock + tid;
    if (tid < kInertialBlock && i < T.n_ine) {
      InertialOut<K, 4> o;
      o.Jp = nullptr;  // (value-only branch)
      inertial_evaluate<K, 4, false>(T, cps, bg, ba, grav, i, false, &o);
      cost = o.cost;
    }
  }
  const double s = block_sum(cost, red);
  if (tid == 0) T.cand_part[n_vis_parts + blk] = s;
}

/// One workgroup of the update launch; false: the solve had ended before (nothing done).
template <int K>
HSD bool update_visual_workgroup(const Tables& T, int R, int Lmax, int n_vis_parts, int nb_pri, double* smem) {
  const int w = blockIdx.x, tid = threadIdx.x;
  DevState* st = T.st;
  if (int(blockIdx.x) >= n_vis_parts + T.n_norm_part) {
    if (st->done) return false;
    update_cost_workgroup<K>(T, blockIdx.x - n_vis_parts - T.n_norm_part, nb_pri, n_vis_parts, smem);
    return true;
  }
  if (int(blockIdx.x) >= n_vis_parts) {  // ---- replicated unknowns: candidate = Plus(x, delta), norms (k_backsub_retract) ----
    if (st->done) return false;
    __shared__ double red[kBlock / 64];
    const int blk = blockIdx.x - n_vis_parts;
    const int j = blk * blockDim.x + tid;
    double xs = 0.0, ss = 0.0;
    if (j < T.sp.n_cp) {
      const double* x = T.cp + 8 * j;
      double* y = T.cp_cand + 8 * j;
      bool any = false;
#pragma unroll
      for (int c = 0; c < 6; ++c) any |= (T.D2p[6 * j + c] != 0.0);
      candidate_control_point(T, j, y);
      if (any) {
#pragma unroll
        for (int c = 0; c < 8; ++c) xs = fma(x[c], x[c], xs), ss = fma(x[c] - y[c], x[c] - y[c], ss);
      }
    }
    if (T.nb > 0) {  // border unknowns (replicated like the control points): bias control points [x y z t] and gravity
      for (int b = j; b < 2 * T.n_bias; b += T.n_norm_part * blockDim.x) {
        const bool acc = b >= T.n_bias;
        const int bi = acc ? b - T.n_bias : b;
        const double* x = (acc ? T.bias_a : T.bias_g) + 4 * bi;
        double* y = (acc ? T.bias_a_cand : T.bias_g_cand) + 4 * bi;
        const bool any = T.D2b[3 * b] != 0.0 || T.D2b[3 * b + 1] != 0.0 || T.D2b[3 * b + 2] != 0.0;
        candidate_bias_point(T, b, y);
        if (any) {
#pragma unroll
          for (int c = 0; c < 4; ++c) xs = fma(x[c], x[c], xs), ss = fma(x[c] - y[c], x[c] - y[c], ss);
        }
      }
      if (j == 0) {
        const double* d = T.delta_b + 6 * T.n_bias;
        double y[3];
        sphere_plus(T.gravity, d, y);
        const bool any = T.D2b[6 * T.n_bias] != 0.0 || T.D2b[6 * T.n_bias + 1] != 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          T.gravity_cand[c] = y[c];
          if (any) xs = fma(T.gravity[c], T.gravity[c], xs), ss = fma(T.gravity[c] - y[c], T.gravity[c] - y[c], ss);
        }
      }
    }
    // landmarks without residuals are in no chunk: their candidate is the landmark itself
    for (int e = 3 * T.n_obs_lm + j; e < 3 * T.n_lm; e += T.n_norm_part * blockDim.x) T.lm_cand[e] = T.lm[e];
    xs = block_sum(xs, red), ss = block_sum(ss, red);
    if (tid == 0) T.norm_part[2 * blk] = xs, T.norm_part[2 * blk + 1] = ss;
    return true;
  }
  // ---- chunk w ----
  const bool uprof = prof_enabled(T.debug_flags, 512) && tid == 0 && w < 1024;  // phase stamps (profiling builds; tools/update_phase_timing.py)
  long long* ulog = reinterpret_cast<long long*>(T.xpart) + 192 * 1024 + 16 * w;
  if (uprof) ulog[0] = wall_clock64();
  const int4 d0 = *reinterpret_cast<const int4*>(T.ch_desc + 8 * w);
  const int nres = T.ch_desc[8 * w + 4];
  const int st_done = st->done, st_spec = st->spec, st_accepted = st->accepted;
  if (st_done) return false;
  if (w >= T.n_chunk) {  // padding workgroups of the partial tables
    if (tid == 0) T.cand_part[w] = 0.0;
    if (tid < 4) T.lm_part[4 * w + tid] = 0.0;
    return true;
  }
  const bool pend = st_spec == 4 && st_accepted;  // the landmarks' current point is still in lm_cand
  const int lo = d0.x, nl = d0.y, cf = d0.z, q0 = d0.w;
  if (uprof && lo + nl + cf + q0 >= 0) ulog[1] = wall_clock64();  // descriptor + state here
  const int bw = T.bw, R6 = 6 * bw;
  const unsigned bw_magic = ((1u << 20) + bw - 1) / bw;
  double* cps_c = smem;                       // candidate control points of the window [bw][8]
  double* yp = cps_c + 8 * bw;                // Sp o y_p over the window rows [6 bw]
  double* part = yp + 6 * bw;                 // [Lmax][bw][4]: partial Yh' (Sp o y_p) per (landmark, control point)
  double* lmc = part + 4 * Lmax * bw;         // [Lmax][8]: candidate landmark (3), decision terms (4)
  double* cams = lmc + 8 * Lmax;              // cameras
  double* costs = cams + 16 * kBuildCams;     // [R]
  double* cpart = costs + R;                  // [8]
  int* l_ncp = reinterpret_cast<int*>(cpart + 8);
  int* l_yoff = l_ncp + Lmax;
  RelPre* relp = reinterpret_cast<RelPre*>(cpart + 8 + Lmax + 8);  // relative rotations of the window's consecutive CANDIDATE control points
  // residual inputs (independent of everything below: requested first)
  const int rt = R > 128 ? tid : (tid >> 6) == 0 ? tid : (tid >> 6) == 2 ? tid - 64 : kBlock;  // (record <-> lane as in k_build_visual: waves 0 and 2)
  const bool has_rec = rt < nres;
  VisualIn in;
  int camid = 0, my_l = 0;
  if (has_rec) {
    const int q = q0 + rt;
    in.first = T.v_first[q], my_l = T.v_lm[q] - lo;
    const int info = T.v_info[q];
    in.type = info >> 16, camid = info & 0xffff;
    in.stamp = T.v_stamp[q];
    in.meas[0] = T.v_meas[3 * q], in.meas[1] = T.v_meas[3 * q + 1], in.meas[2] = T.v_meas[3 * q + 2];
  }
  const int ncp_w = min(bw, T.sp.n_cp - cf);
  if (tid < ncp_w) {  // candidate control points of the window
    const double* x = T.cp + 8 * (cf + tid);
    const double* d = T.delta_p + 6 * (cf + tid);
    const Quat q = quat_plus(Quat{x[0], x[1], x[2], x[3]}, V3{d[0], d[1], d[2]});
    double* y = cps_c + 8 * tid;
    y[0] = q.x, y[1] = q.y, y[2] = q.z, y[3] = q.w, y[4] = x[4] + d[3], y[5] = x[5] + d[4], y[6] = x[6] + d[5], y[7] = x[7];
  }
  for (int e = tid; e < 6 * ncp_w; e += kBlock) yp[e] = -T.step_p[6 * cf + e] * T.scale_p[6 * cf + e];
  for (int e = tid; e < 16 * min(T.n_cam, kBuildCams); e += kBlock) cams[e] = T.cam[e];
  if (tid < nl) l_ncp[tid] = T.lm_ncp[lo + tid], l_yoff[tid] = T.lm_yoff[lo + tid];
  // Everything the landmark lanes need behind the dot products is requested in this round too (the kernel is a chain of memory round
  // trips: descriptor -> tables -> Y-hat; as a fourth and fifth round these loads cost ~3 us of its 12.8)
  double L[6], yh[3], x[3], sc[3], sb[3], d2[3];
  bool active = false;
  if (tid < nl) {
    const int dl = lo + tid;
#pragma unroll
    for (int a = 0; a < 6; ++a) L[a] = T.lm_L[6 * dl + a];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      yh[a] = T.lm_yhat[3 * dl + a], x[a] = (pend ? T.lm_cand : T.lm)[3 * dl + a], sc[a] = T.lm_scale[3 * dl + a];
      sb[a] = T.lm_sb[3 * dl + a], d2[a] = T.lm_D2[3 * dl + a];
    }
    active = !T.lm_const[dl];
  }
  __syncthreads();
  if (uprof) ulog[2] = wall_clock64();  // tables staged
  if ((tid >> 6) == 3 && (tid & 63) + 1 < ncp_w) relp[tid & 63] = rel_precompute(cps_c + 8 * (tid & 63), cps_c + 8 * (tid & 63) + 8);  // (read behind the next barriers)
  // partial dot products, one (landmark, control point) block per lane: the 6 x 3 block of Y-hat is 18 consecutive doubles in HBM
  const int n_task = nl * bw;
  for (int tau = tid; tau < n_task; tau += kBlock) {
    const int l = int((unsigned(tau) * bw_magic) >> 20), jb = tau - l * bw;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    if (jb < l_ncp[l]) {
      const double2* Y = reinterpret_cast<const double2*>(T.Y + l_yoff[l] + 18 * jb);
      double y[18];
#pragma unroll
      for (int e = 0; e < 9; ++e) {
        const double2 v = Y[e];
        y[2 * e] = v.x, y[2 * e + 1] = v.y;
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const double p = yp[6 * jb + r];
        t0 = fma(y[3 * r], p, t0), t1 = fma(y[3 * r + 1], p, t1), t2 = fma(y[3 * r + 2], p, t2);
      }
    }
    double* dst = part + 4 * (l * bw + jb);
    dst[0] = t0, dst[1] = t1, dst[2] = t2;
  }
  __syncthreads();
  if (uprof) ulog[3] = wall_clock64();  // Y-hat dot products done
  if (tid < nl) {  // 3 x 3 back-substitution per landmark
    const int l = tid, dl = lo + l;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0;
    for (int jb = 0; jb < l_ncp[l]; ++jb) t0 += part[4 * (l * bw + jb)], t1 += part[4 * (l * bw + jb) + 1], t2 += part[4 * (l * bw + jb) + 2];
    const double z0 = yh[0] - t0, z1 = yh[1] - t1, z2 = yh[2] - t2;  // L' y = z
    const double y2 = z2 / L[5], y1 = (z1 - L[4] * y2) / L[2], y0 = (z0 - L[1] * y1 - L[3] * y2) / L[0];
    const double s[3] = {active ? -y0 : 0.0, active ? -y1 : 0.0, active ? -y2 : 0.0};
    double xl = 0.0, sl = 0.0, gd = 0.0, dd = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double y = x[a] + sc[a] * s[a];
      T.lm_cand[3 * dl + a] = y;
      lmc[8 * l + a] = y;
      if (pend) T.lm[3 * dl + a] = x[a];
      if (active) {
        xl = fma(x[a], x[a], xl), sl = fma(x[a] - y, x[a] - y, sl);
        gd = fma(sb[a], s[a], gd);
        dd = fma(d2[a] * s[a], s[a], dd);
      }
    }
    lmc[8 * l + 4] = xl, lmc[8 * l + 5] = sl, lmc[8 * l + 6] = gd, lmc[8 * l + 7] = dd;
  }
  __syncthreads();
  if (uprof) ulog[4] = wall_clock64();  // landmarks back-substituted
  if (tid >= kBlock - 4) {  // landmark-side terms of the decision (|x|^2, |x - x+|^2, g.step, step'D^2 step), landmarks in order
    const int e = tid - (kBlock - 4);
    double v = 0.0;
    for (int l = 0; l < nl; ++l) v += lmc[8 * l + 4 + e];
    T.lm_part[4 * w + e] = v;
  }
  // value-only cost of the chunk's residual blocks at the candidate
  if (has_rec) {
    in.cam = camid < kBuildCams ? cams + 16 * camid : T.cam + kCamStride * camid;
    in.lm[0] = lmc[8 * my_l], in.lm[1] = lmc[8 * my_l + 1], in.lm[2] = lmc[8 * my_l + 2];
    costs[rt] = visual_cost_in<K>(T, cps_c - 8 * cf, in, relp - cf);
  }
  __syncthreads();
  if (uprof) ulog[5] = wall_clock64();  // candidate costs
  if (tid < 8) {  // fixed-order sum: eight strided partials, then their sum
    const int per = (nres + 7) / 8;
    double acc = 0.0;
    for (int t = tid * per; t < min(nres, (tid + 1) * per); ++t) acc += costs[t];
    cpart[tid] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    double s = 0.0;
#pragma unroll
    for (int g = 0; g < 8; ++g) s += cpart[g];
    T.cand_part[w] = s;
  }
  if (uprof) ulog[6] = wall_clock64();
  return true;
}

/// nb_pri / nb_ine: cost workgroups of the prior / inertial factors behind the norm workgroups (0: their own launches follow).
/// (Round 6 also let the workgroup that finishes LAST take the trust-region decision — every workgroup releases its partials and draws a ticket —
///  to save k_pack_decision's launch: an agent-scope release writes the XCD's L2 back, and ~150 of them made the launch 16 us longer (45 us from
///  every wave) where the one-workgroup launch behind it costs 5.5. Measured on the replays, not kept: a kernel boundary does that once.)
template <int K>
__global__ void __launch_bounds__(kBlock) k_update_visual(Tables T, int R, int Lmax, int n_vis_parts, int nb_pri = 0, int nb_ine = 0) {
  HS_DYNAMIC_LDS(smem);
  update_visual_workgroup<K>(T, R, Lmax, n_vis_parts, nb_pri, smem);
}

}  // namespace hs
