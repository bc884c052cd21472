// host_launch.hpp — the launch sequence of one LM iteration (linearise / build / factor + solve / update), the RCCL exchange and the
// one-time kernel set-up (part of capi.hip: included once, by it, behind host_tables.hpp).
#pragma once
#include "host_tables.hpp"

namespace {

int reset_state(hs_problem* p, int max_iterations, double radius, int spec = 0) {
  k_reset_state<<<1, 64, 0, p->stream>>>(p->d_state.p, max_iterations, radius, spec);
  HIP_TRY(hipGetLastError());
  return HS_OK;
}

size_t cp_lds_bytes(const hs_problem* p) { return size_t(8) * p->n_cp * sizeof(double); }
template <int K>
size_t lin_lds_bytes(const hs_problem* p) {  // control points + one record slab per wave
  return (cp_lds_bytes(p) <= 24 * 1024 ? cp_lds_bytes(p) : 0) + size_t(lin_block<K>()) * (8 + 12 * K + 2) * sizeof(double);
}

__global__ void k_noop() {}

/// The side stream of the inertial branch and its three events. Created by hs_create and used once there: a stream gets its hardware
/// queue at its first submission, which — together with the first allocations — made the first optimize() with an IMU 10 ms long.
static int ensure_side_stream(hs_problem* p) {
  if (!p->side) {
    HIP_TRY(hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&p->ev_irec, hipEventDisableTiming));
    k_noop<<<1, 64, 0, p->side>>>();
    HIP_TRY(hipEventRecord(p->ev_join, p->side));
    HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_join, 0));
    k_noop<<<1, 64, 0, p->stream>>>();
    HIP_TRY(hipStreamSynchronize(p->stream));
  }
  return HS_OK;
}

/// `inertial_on_side` (the solve loop of bordered systems): the inertial branch of an iteration — k_linearize_inertial, then the border
/// gathers k_border_pb / _bb in launch_build — only meets the visual branch (k_linearize_visual -> k_landmark -> Gram
/// kernels -> k_assemble) at the segment Gram kernel (reads the inertial records) and at k_reduce_partials, and each branch fills a
/// fraction of the chip: they run on two streams. configs[2]: 293 us of kernels back to back -> 175 us on the critical path.
/// Fused build (p->fused): the visual factors are linearised by k_build_visual inside launch_build — nothing to do for them here, unless only
/// the cost is wanted (`visual_cost_only`: hs_cost, hs_solve with zero iterations), which the value-only kernel delivers.
template <int K>
int launch_linearize(hs_problem* p, bool inertial_on_side = false, bool visual_cost_only = false) {
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  p->side_imu = inertial_on_side && T.n_ine > 0 && T.nb > 0 && !(T.debug_flags & 1048576);  // A/B switch 1048576: one stream
  if (p->side_imu) {
    const int rc = ensure_side_stream(p);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(p->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0));
  }
  if (T.n_vis && !p->fused) k_linearize_visual<K><<<p->nb_vis, lin_block<K>(), lin_lds_bytes<K>(p), s>>>(T, T.v_rec, T.v_pos, 1, T.cost_part, nullptr);
  if (T.n_vis && p->fused && visual_cost_only) k_cost_visual<K><<<p->nb_vis, kBlock, cp_lds_bytes(p), s>>>(T, T.cp, T.lm, T.cost_part);
  if (T.n_pri) k_linearize_prior<K><<<p->nb_pri, kBlock, cp_lds_bytes(p), s>>>(T, T.p_rec, T.cost_part + p->nb_vis, nullptr);
  if (T.n_ine)
    k_linearize_inertial<K, 4><<<p->nb_ine, kInertialBlock * K, cp_lds_bytes(p), p->side_imu ? p->side : s>>>(T, T.i_rec, 1, T.cost_part + p->nb_vis + p->nb_pri,
                                                                                                        nullptr);
  if (p->side_imu) HIP_TRY(hipEventRecord(p->ev_irec, p->side));
  HIP_TRY(hipGetLastError());
  return HS_OK;
}

// ---- RCCL, loaded on first use ---------------------------------------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
      api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
      api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
      api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
      api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
      api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
      api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(api.lib, "ncclCommCount"));
      if (!api.GetUniqueId || !api.CommInitRank || !api.AllReduce || !api.CommDestroy) api.lib = nullptr;
    }
  }
  return api.lib ? &api : nullptr;
}

int exchange(hs_problem* p, double* buf, int64_t count) {
  if (p->rccl_comm) {  // one in-place sum all-reduce on the library's stream, no host involvement
    const ncclResult_t r = rccl_api()->AllReduce(buf, buf, size_t(count), ncclDouble, ncclSum, static_cast<ncclComm_t>(p->rccl_comm), p->stream);
    if (r != ncclSuccess) HS_FAIL(HS_ERR_DEVICE, std::string("ncclAllReduce failed: ") + (rccl_api()->GetErrorString ? rccl_api()->GetErrorString(r) : "?"));
    return HS_OK;
  }
  if (!p->allreduce) return HS_OK;
  if (p->allreduce(p->allreduce_user, buf, count, p->stream) != 0) HS_FAIL(HS_ERR_DEVICE, "all-reduce hook reported a failure");
  return HS_OK;
}

static int border_zero_wgs(const Tables& T) { return std::min(64, (T.nb * T.nb + T.nb + kPbThreads - 1) / kPbThreads); }

/// `after_build` (stage timing of a fused build): recorded behind k_build_visual — the launch that linearises the visual factors belongs to
/// the "linearise" stage of hs_summary, what follows it (segment Gram of the prior / inertial records, assembly, finalisation) to "schur".
/// launch_factor's rule for the two-ended look-ahead factorisation (k_band_factor_la, grid 2) of a system without border unknowns.
static bool factor_two_ended_la(const hs_problem* p) {
  const Tables& T = p->T;
  const int n_blk = T.np / 6;
  return T.nb == 0 && !(T.debug_flags & 4) && la_compute_waves(T.bw) > 0 && !HS_AB(T.debug_flags, 131072) && n_blk >= 4 * T.bw && T.Sb2 && !(T.debug_flags & 2048);
}

int mfma_window_tiles(int bw);
/// launch_factor's rule for factoring from both ends at once (every case: k_band_factor_mx / _la, bordered or not; profiling builds: _mfma).
static bool factor_two_ended_any(const hs_problem* p) {
  const Tables& T = p->T;
  const int n_blk = T.np / 6;
  const bool la_ok = !(T.debug_flags & 4) && la_compute_waves(T.bw) > 0;
  const int nt = HS_AB(T.debug_flags, 131072) ? mfma_window_tiles(T.bw) : 0;
  return (la_ok || nt) && (T.nb == 0 || (!nt && !(T.debug_flags & 536870912) && (T.nb + kBorderCols - 1) / kBorderCols <= 512)) && n_blk >= 4 * T.bw && T.Sb2 &&
         !(T.debug_flags & 2048);  // (512: flag words of k_border_forward2's column groups)
}
/// launch_factor's rule for k_dense_solve_mx: one-ended solves of small systems — the sliding window's steady state (kernels_dense_mx.hpp).
/// *f0 = the decoupled block rows of the leading constant control points. launch_build asks too: the finalisation kernels then write the dense
/// copy of the system the kernel loads its tiles from (Tables::dense). A/B switch 8: the one-ended band kernels + border chain + k_band_backward.
static bool use_dense_mx(const hs_problem* p, int* f0) {
  const Tables& T = p->T;
  const int n_blk = T.np / 6;
  *f0 = (T.debug_flags & 262144) ? 0 : std::min(p->frozen_prefix, n_blk - 1);  // A/B switch 262144: eliminate every block row
  return !factor_two_ended_any(p) && !HS_AB(T.debug_flags, 131072) && dense_mx_fits(n_blk - *f0, T.nb) && !(T.debug_flags & 8);
}

/// scaling_fixed: a step of this solve has been computed (every linearisation but the first): the Jacobi scaling is fixed, and on a single
/// shard without border unknowns whose factorisation is the two-ended look-ahead kernel k_assemble writes the scaled, damped system itself
/// (direct mode) — no k_finalize_reduced launch; the iteration bookkeeping moves into the factorisation's prologue (Tables::bookkeep).
/// fold: the decision of the previous iteration was not launched behind its update (launch_update, fold_decision_into_build): workgroup 0 of
/// k_build_visual takes it, the chunk workgroups wait for its flag (Tables::fold_decision).
template <int K>
int launch_build(hs_problem* p, hipEvent_t after_build = nullptr, bool scaling_fixed = false, bool fold = false) {
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  p->bookkeep = false;
  // k_seg_gram only needs the records, k_landmark -> k_group_gram records and landmarks: the two gram kernels share one launch
  // (k_gram_pair). A/B switch 1024: the previous arrangement, k_seg_gram on a side stream next to k_landmark -> k_group_gram.
  // (for small grids only — configs[1]: ~940 workgroups, Schur stage 66 -> 62 us. The pair holds 80 KB of LDS per workgroup, two per
  //  CU, where k_group_gram alone fits three: at configs[3], ~3 750 workgroups, the two streams are faster, 0.165 vs 0.181 ms)
  const bool fused = p->fused;
  const bool pair = !fused && T.n_lm > 0 && p->n_group_wg > 0 && p->n_group_wg + p->n_seg_wg <= 2048 && !(T.debug_flags & 1024);
  const bool side_imu = p->side_imu;       // (set by launch_linearize: the side stream is busy with the inertial branch)
  const bool fork = !fused && T.n_lm > 0 && !pair && !side_imu;
  // Fused build: linearisation, landmark elimination and both Gram terms of the visual factors in one launch; what remains for the segment
  // Gram kernel are the prior / inertial records (none on visual-only windows: no launch)
  if (fused && fold) {
    Tables Tf = T;
    Tf.fold_decision = 1, Tf.fold_epoch = ++p->join_epoch;
    k_build_visual<K><<<p->nb_vis + 1, kBlock, p->build_lds, s>>>(Tf, p->build_R, p->build_L, 1);
  } else if (fused)
    k_build_visual<K><<<p->nb_vis, kBlock, p->build_lds, s>>>(T, p->build_R, p->build_L, 1);
  if (fused && T.wide_q) k_landmark_gram_wide<<<landmark_gram_wide_grid(T.sp.n_cp, T.bw), kGramWideThreads, 0, s>>>(T);  // window-wide bands: -Yh Yh' once per window
  if (fused && after_build) HIP_TRY(hipEventRecord(after_build, s));
  if (fork) {
    const int rc = ensure_side_stream(p);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(p->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(p->side, p->ev_fork, 0));
  }
  hipStream_t sb = side_imu ? p->side : s;  // stream of the border gathers
  // Single shard: the border workgroups of k_finalize_reduced pick the gathers' results up through a device flag (Tables::gather_epoch) — no
  // event between the side stream and the main stream in front of that launch. A/B switch 16384: the event.
  const bool gather_flag = side_imu && T.nb && !p->allreduce && !p->rccl_comm && p->world == 1 && !(T.debug_flags & 16384) &&
                           !(T.debug_flags & 8388608);
  const unsigned gather_epoch = gather_flag ? ++p->join_epoch : 0u;
  // (Round 6 tried the segment Gram kernel of a fused stereo-inertial window on the side stream, behind k_linearize_inertial and next to
  //  k_build_visual: 8 us off the main stream's chain, and the event k_assemble then waits for cost more — 1.080 against 1.038 ms per optimize().)
  if (side_imu && T.nb) {  // behind k_linearize_inertial on the side stream, next to k_landmark / the Gram kernels
    // (Round 6 also ran both gathers as ONE launch — zero-fill moved into k_linearize_inertial, the H_bb workgroups first: 82 us at configs[2]
    //  where the two launches take 32 + 45: the gathers do not overlap, they share whatever bounds them. Not kept.)
    k_border_pb<K><<<dim3(T.sp.n_cp + border_zero_wgs(T), p->n_split), kPbThreads, 0, sb>>>(T);  // (+ zero-fill of the border-border block)
    Tables Tg = T;
    Tg.gather_epoch = gather_epoch;
    k_border_bb<K><<<T.n_bias, kBlock, 0, sb>>>(Tg);
    HIP_TRY(hipEventRecord(p->ev_join, p->side));
  }
  bool irec_ready = !side_imu;  // the segment Gram kernel reads the inertial records: wait for the side stream's linearisation once
  auto need_irec = [&]() -> hipError_t {
    if (irec_ready) return hipSuccess;
    irec_ready = true;
    return hipStreamWaitEvent(s, p->ev_irec, 0);
  };
  if (!pair && !(side_imu && T.n_lm) && p->n_seg_wg) {
    HIP_TRY(need_irec());
    k_seg_gram<K><<<p->n_seg_wg, kBlock, kSegStage * sizeof(double), fork ? p->side : s>>>(T);
  }
  if (fork) HIP_TRY(hipEventRecord(p->ev_join, p->side));
  if (T.n_lm && !fused) {
    const int grid = (T.n_lm + kBlock / 64 - 1) / (kBlock / 64);
    if (6 * T.bw <= 128)
      k_landmark<K, 2, 2><<<grid, kBlock, 0, s>>>(T);
    else if (T.debug_flags & 4194304)  // A/B switch 4194304: one wave per landmark with four passes
      k_landmark<K, 4, 1><<<grid, kBlock, 0, s>>>(T);
    else  // long feature tracks (6 * bw <= kBlock is checked in prepare()): one workgroup per landmark, one wave per 64 rows of W
      k_landmark_rows<K, 4><<<T.n_lm, kBlock, 0, s>>>(T);
  }
  if (T.n_lm && p->n_group_wg && !fused) {
    const int ntile = T.bw * (T.bw + 1) / 2;
    const int batch = std::max(2, std::min(kGroupBatch, int(48 * 1024 / (size_t(18) * T.bw * sizeof(double)))));
    const size_t lds = std::max((size_t(batch) * 18 * T.bw + 4 * batch) * sizeof(double), size_t(128) * 42 * sizeof(double));
    const dim3 grid(p->n_group_wg);
    if (pair) {
      HIP_TRY(need_irec());
      const size_t lds2 = std::max(lds, kSegStage * sizeof(double));
      const dim3 grid2(p->n_group_wg + p->n_seg_wg);
      if (ntile <= kBlock)
        k_gram_pair<K, 1><<<grid2, kBlock, lds2, s>>>(T, batch, p->n_group_wg);
      else if (ntile <= 2 * kBlock)
        k_gram_pair<K, 2><<<grid2, kBlock, lds2, s>>>(T, batch, p->n_group_wg);
      else
        k_gram_pair<K, 4><<<grid2, kBlock, lds2, s>>>(T, batch, p->n_group_wg);
    } else if (ntile <= kBlock)
      k_group_gram<1><<<grid, kBlock, lds, s>>>(T, batch);
    else if (ntile <= 2 * kBlock)
      k_group_gram<2><<<grid, kBlock, lds, s>>>(T, batch);
    else
      k_group_gram<4><<<grid, kBlock, lds, s>>>(T, batch);
  }
  if (side_imu && T.n_lm && !pair && p->n_seg_wg) {  // (large grids with an IMU: the segment Gram kernel after the landmark chain, same stream)
    HIP_TRY(need_irec());
    k_seg_gram<K><<<p->n_seg_wg, kBlock, kSegStage * sizeof(double), s>>>(T);
  }
  if (fork) HIP_TRY(hipStreamWaitEvent(s, p->ev_join, 0));
  // (direct mode needs a factorisation that does the iteration bookkeeping: the two-ended look-ahead / matrix-core kernels, and — round 6 —
  //  k_dense_solve_mx, which then also gets its dense copy of the system from the assembly)
  Tables Ta = T;
  const bool dense_mx = use_dense_mx(p, &Ta.dense_f0);
  //  Bordered systems stay on k_finalize_reduced: with the border blocks scaled by extra workgroups of the assembly the main stream has to wait
  //  for the side stream's border gathers BEFORE the assembly instead of behind it — measured on the stereo-inertial replay: 1.09 against
  //  1.03 ms per optimize().)
  const bool direct = scaling_fixed && fused && !T.nb && !p->allreduce && !p->rccl_comm && p->world == 1 && (factor_two_ended_la(p) || dense_mx) &&
                      !(T.debug_flags & 4096);  // A/B switch 4096: k_finalize_reduced in every iteration
  if (direct && dense_mx) Ta.dense = p->d_dense_ut.p + size_t(kDenseLd) * kDenseLd;
  // window-wide bands on the fused build (more than 256 window tiles): a row collects every chunk of a short window — k_assemble_wide
  if (fused && T.bw * (T.bw + 1) / 2 > kBlock)
    k_assemble_wide<K><<<dim3(T.sp.n_cp, 6), kAsmWideThreads, 0, s>>>(Ta, direct ? 1 : 0);
  else
    k_assemble<K><<<dim3(T.sp.n_cp, 6), kAsmThreads, 0, s>>>(Ta, direct ? 1 : 0);
  if (direct) {
    p->bookkeep = true;
    HIP_TRY(hipGetLastError());
    return HS_OK;
  }
  if (T.nb && !side_imu) {
    k_border_pb<K><<<dim3(T.sp.n_cp + border_zero_wgs(T), p->n_split), kPbThreads, 0, s>>>(T);
    k_border_bb<K><<<T.n_bias, kBlock, 0, s>>>(T);
  }
  if (side_imu && !gather_flag) HIP_TRY(hipStreamWaitEvent(s, p->ev_join, 0));  // border gathers done
  // Nothing to exchange (single shard): packing + bookkeeping are an extra workgroup of k_finalize_reduced, the border blocks further
  // ones that sum the accumulation splits themselves — one launch where the exchanging path has five (~5 us each on the chain).
  // A/B switch 8388608: the five launches.
  const bool reduce_here = !p->allreduce && !p->rccl_comm && p->world == 1 && !(T.nb && (T.debug_flags & 8388608));
  const int nb_wg = T.nb ? std::min(256, ((T.np + T.nb) * T.nb + kBlock - 1) / kBlock) : 0;
  if (T.nb && !reduce_here)
    k_reduce_partials<<<std::min(1024, (T.xo_bb - T.xo_pb + kBlock - 1) / kBlock), kBlock, 0, s>>>(T, p->n_split, T.xo_pb);
  if (!reduce_here) k_pack_exchange<<<1, kBlock, 0, s>>>(T, 0);
  HIP_TRY(hipGetLastError());
  const int rc = exchange(p, T.xbuf, T.x_count1);  // one RCCL all-reduce of [S | g | diag | cost] per linearisation
  if (rc) return rc;
  Tables Td = T;
  Td.gather_epoch = gather_epoch;
  if (use_dense_mx(p, &Td.dense_f0)) Td.dense = p->d_dense_ut.p + size_t(kDenseLd) * kDenseLd;  // (second half of the scratch: the first is the factor by columns)
  k_finalize_reduced<<<T.sp.n_cp + (reduce_here ? 1 + nb_wg : 0), kBlock, 0, s>>>(Td, p->n_split);  // + 1: packing / bookkeeping workgroup, + border
  if (T.nb && !reduce_here) k_finalize_border<<<nb_wg, kBlock, 0, s>>>(Td);
  if (!reduce_here) k_cost_reduce<<<1, kBlock, 0, s>>>(T);
  HIP_TRY(hipGetLastError());
  return HS_OK;
}

/// Window size (in 16 x 16 tiles) of the MFMA factorisation for a band of bw blocks: 16 NT >= 6 bw + 12; 0: not supported.
int mfma_window_tiles(int bw) {
  for (int nt : {6, 9}) {  // (NT = 10 would cover bw <= 24: 256 VGPRs + scratch, and wrong results on gfx950 — not instantiated)
    if (6 * bw + 12 <= 16 * nt) return nt;
  }
  return 0;
}

#if HS_PROFILE_HOOKS
template <int NT, int NC>
hipError_t launch_mfma(const Tables& T, int grid, hipStream_t s) {
  static bool attr = false;
  const size_t lds = size_t(MfmaGeom<NT>::kTotal) * sizeof(double);
  if (!attr) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_mfma<NT, NC>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    attr = true;
  }
  k_band_factor_mfma<NT, NC><<<grid, 64 * (NC + 3), lds, s>>>(T);
  return hipGetLastError();
}

void launch_backward_w(const Tables& T, const BackJob& j0, const BackJob& j1, int m_mid, int grid, hipStream_t s) {
  const size_t lds = size_t(T.np) * sizeof(double);
  if (T.bw <= kBackBlocks)
    k_band_backward_w<1><<<grid, 64, lds, s>>>(T, j0, j1, m_mid);
  else if (T.bw <= 2 * kBackBlocks)
    k_band_backward_w<2><<<grid, 64, lds, s>>>(T, j0, j1, m_mid);
  else if (T.bw <= 4 * kBackBlocks)
    k_band_backward_w<4><<<grid, 64, lds, s>>>(T, j0, j1, m_mid);
  else
    k_band_backward_w<5><<<grid, 64, lds, s>>>(T, j0, j1, m_mid);
}

#endif  // HS_PROFILE_HOOKS

/// Dense Cholesky of the border Schur complement + solve for the border unknowns (one workgroup).
static hipError_t launch_border_solve(const Tables& T, hipStream_t s) {
  if (T.nb + 1 <= 128 && !(T.debug_flags & 524288)) {  // trailing matrix in registers (A/B switch 524288: the LDS version)
    const int R = std::max(3, (T.nb + 1 + 15) / 16), N = 16 * R;
    const size_t lds = (size_t(4) * N + size_t(T.nb) * (N + 1) + T.nb) * sizeof(double);
    switch (R) {
      case 3: k_border_solve_reg<3><<<1, kBlock, lds, s>>>(T); break;
      case 4: k_border_solve_reg<4><<<1, kBlock, lds, s>>>(T); break;
      case 5: k_border_solve_reg<5><<<1, kBlock, lds, s>>>(T); break;
      case 6: k_border_solve_reg<6><<<1, kBlock, lds, s>>>(T); break;
      case 7: k_border_solve_reg<7><<<1, kBlock, lds, s>>>(T); break;
      default: k_border_solve_reg<8><<<1, kBlock, lds, s>>>(T); break;
    }
  } else {
    k_border_solve<<<1, kBlock, (size_t(T.nb + 1) * (T.nb + 1) + T.nb) * sizeof(double), s>>>(T);
  }
  return hipGetLastError();
}

int launch_factor(hs_problem* p) {
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  const int ncb = 6 * T.bw;
  const size_t chol_lds = (size_t(24) * (ncb + 2) + size_t(T.np)) * sizeof(double);
  const size_t la_lds = (size_t(42) * (ncb + 2) + size_t(T.np) + 48) * sizeof(double);
  const bool legacy = T.debug_flags & 4;  // A/B switch: pre-look-ahead kernel
  // Factoring from both ends at once (visual-only systems, look-ahead kernel, window long enough to pay for the junction)
  const int n_blk = T.np / 6, w_mid = T.bw - 1;
  const bool la_ok = !legacy && la_compute_waves(T.bw) > 0;
  const int la_ncw = la_compute_waves(T.bw);
  const int nt = HS_AB(T.debug_flags, 131072) ? mfma_window_tiles(T.bw) : 0;  // A/B switch 131072 (profiling builds): k_band_factor_mfma instead of the VALU kernels
  // (bordered systems — bias splines + gravity — too: the forward sweep of the border columns follows the two-ended elimination order,
  //  k_border_forward2; A/B switch 536870912: bordered systems one-ended)
  const bool two_ended = factor_two_ended_any(p);
#if HS_PROFILE_HOOKS
  auto run_mfma = [&](const Tables& TT, int grid) -> hipError_t {
    switch (nt) {
      case 6: return launch_mfma<6, 3>(TT, grid, s);
      default: return launch_mfma<9, 3>(TT, grid, s);
    }
  };
#else
  auto run_mfma = [&](const Tables&, int) -> hipError_t { return hipErrorNotSupported; };  // (nt == 0: never reached)
#endif
  if (two_ended) {
    // The near end takes a few block rows more than the far end: the far end still has to hand its trailing window over (~7 us,
    // i.e. ~4 steps: window through HBM + agent-scope release) before the near end can pass the junction. With an even split
    // workgroup 0 waited 13 us there (tools/chol_phase_timing.py).
    // (k_band_factor_la, +2 / +3 / +4: 132.0 / 130.5 / 132.1 us; k_band_factor_mx hands its window over in 4.3 us and takes 0.96 us per block row:
    //  the far end is there 3.3 us early with +3 and 0.8 us late with +1)
    const bool use_mx = !nt && mx_fits(T.bw) && !(T.debug_flags & 64);  // A/B switch 64: the VALU look-ahead kernel
    const int m = std::min((n_blk - w_mid) / 2 + two_ended_lead(use_mx), n_blk - w_mid - w_mid), mB = n_blk - w_mid - m;
    Tables T2 = T;
    T2.fj[0] = FactorJob{T.Sb, T.g_s, T.Ub, T.Ubk, T.ybuf, p->d_win.p, m + w_mid, m};
    T2.fj[1] = FactorJob{p->d_Sb2.p, p->d_g2.p, p->d_Ub2.p, p->d_Ubk2.p, p->d_ybuf2.p, p->d_win.p, mB, -1};
    T2.mj[0] = MfmaJob{p->d_Sb2.p, T.g_s, T.Ub, T.Ubk, T.ybuf, p->d_win.p, m + w_mid, m, m + w_mid, INT_MAX, 0, p->d_Vb.p + p->vb_len};
    T2.mj[1] = MfmaJob{T.Sb, p->d_g2.p, p->d_Ub2.p, p->d_Ubk2.p, p->d_ybuf2.p, p->d_win.p, mB, -1, mB + w_mid, mB, 1, p->d_Vb.p + p->vb_len};
    T2.join_epoch = ++p->join_epoch;
    T2.bookkeep = p->bookkeep && !nt ? 1 : 0;
    // Bordered systems on k_band_factor_mx: the forward sweep of the border columns (k_border_forward2, 44 us behind the factorisation at
    // configs[2]) runs on the side stream WHILE the two ends factor and follows them row by row (MfmaJob::progress, BfJob::progress).
    // A/B switch 128: behind the factorisation on the main stream.
    // Only while the sweep's workgroups — which spin on the factorisation's progress — cannot crowd the factorisation's two workgroups out
    // of the device (they would wait for each other until the 2 s give-up): at most half of the CUs minus a reserve, counting what one CU
    // holds of them by LDS (another process may run the same pair on this device: the world-2-on-one-GPU tests).
    const size_t fwd_lds = size_t(T.np) * kBorderLd * sizeof(double);
    const int fwd_per_cu = std::max(1, int(std::min<size_t>((size_t(160) * 1024) / std::max<size_t>(fwd_lds, 1), 8)));
    const int fwd_cus = (2 * ((T.nb + kBorderCols - 1) / kBorderCols) + fwd_per_cu - 1) / fwd_per_cu;
    const bool pipe = use_mx && T.nb && p->side && fwd_cus <= p->n_cu / 2 - 8 && !(T.debug_flags & 128);
    unsigned* progress = p->d_join.p + kBfFlagBase + 512;  // near U, near W, far U, far W: kProgressStride words apart
    const unsigned progress_base = unsigned(T2.join_epoch) << 12;  // (epoch mod 2^20 | rows: the poller compares the epoch for equality)
    if (pipe) {
      T2.mj[0].progress = progress, T2.mj[1].progress = progress + 2 * kProgressStride;
      T2.mj[0].progress_base = T2.mj[1].progress_base = progress_base;
      // (no event from this stream to the side stream: k_border_forward2 reads S_pb behind the first rows the factorisation publishes —
      //  which are behind k_finalize_reduced — and everything else it touches is its own stream's or the progress protocol's)
    }
    if (nt)
      HIP_TRY(run_mfma(T2, 2));
    else if (use_mx)  // trailing window in the accumulators of the f64 matrix cores (kernels_factor_mx.hpp)
      if (mx_wide(T.bw) && pipe)
        k_band_factor_mx<true, true><<<2, kMxThreads, size_t(kMxLds) * sizeof(double), s>>>(T2);
      else if (mx_wide(T.bw))
        k_band_factor_mx<true><<<2, kMxThreads, size_t(kMxLds) * sizeof(double), s>>>(T2);
      else if (pipe)
        k_band_factor_mx<false, true><<<2, kMxThreads, size_t(kMxLds) * sizeof(double), s>>>(T2);
      else
        k_band_factor_mx<false><<<2, kMxThreads, size_t(kMxLds) * sizeof(double), s>>>(T2);
    else
      if (la_ncw == 3)
        k_band_factor_la<1, 3><<<2, la_threads(3), la_lds, s>>>(T2);
      else
        k_band_factor_la<1, 4><<<2, la_threads(4), la_lds, s>>>(T2);
    if (T.nb) {  // bordered system: Z = U^-T S_pb in the two-ended elimination order, border Schur complement and solve, y' = y - Z x_b
      Tables Tb = T2;
      Tb.ybuf2 = p->d_ybuf2.p, Tb.y_split = 6 * (m + w_mid);
      Tb.join_epoch = ++p->join_epoch;
      const int n_groups = (T.nb + kBorderCols - 1) / kBorderCols;
      HIP_TRY(p->d_bf_handover.reserve(size_t(n_groups) * 6 * w_mid * kBorderCols + 1));
      const int fwd_threads = std::max(128, 64 * ((6 * w_mid + 63) / 64));  // one lane per pending row
      const int local_rows = (!p->allreduce && !p->rccl_comm && p->world == 1) ? 1 : 0;
      // (single shard: k_border_schur picks the sweep's end up through a device flag, Tables::sweep_epoch — A/B switch 16384: an event)
      const bool sweep_flag = pipe && !p->allreduce && !p->rccl_comm && p->world == 1 && !(T.debug_flags & 16384);
      if (sweep_flag) Tb.sweep_epoch = ++p->join_epoch;
      if (pipe) {  // (+ one wave that polls the factorisation's progress)
        k_border_forward2<<<dim3(n_groups, 2), fwd_threads + 64, size_t(T.np) * kBorderLd * sizeof(double), p->side>>>(
            Tb, BfJob{T.Ub, T.Ubk, m + w_mid, 0, progress, progress_base}, BfJob{p->d_Ub2.p, p->d_Ubk2.p, mB, 1, progress + 2 * kProgressStride, progress_base}, m, 0, local_rows,
            p->d_bf_handover.p);
        HIP_TRY(hipEventRecord(p->ev_join, p->side));
        if (!sweep_flag) HIP_TRY(hipStreamWaitEvent(s, p->ev_join, 0));
      } else {
        k_border_forward2<<<dim3(n_groups, 2), fwd_threads, size_t(T.np) * kBorderLd * sizeof(double), s>>>(
            Tb, BfJob{T.Ub, T.Ubk, m + w_mid, 0, nullptr, 0}, BfJob{p->d_Ub2.p, p->d_Ubk2.p, mB, 1, nullptr, 0}, m, 0, local_rows, p->d_bf_handover.p);
      }
      // Border Schur complements of 64 .. 255 unknowns (configs[2]: 110) are factored and solved by k_dense_solve_mx in border mode (round 6:
      // ~32 us where the register Cholesky k_border_solve_reg takes 50): k_border_schur writes C | h into the dense layout as well.
      // A/B switch 8192 (with a border): k_border_solve_reg.
      const bool dense_tail = T.nb >= 64 && T.nb + 1 <= 16 * kDxTiles && !(T.debug_flags & 8192);
      if (dense_tail) {
        Tb.dense = p->d_dense_ut.p + size_t(kDenseLd) * kDenseLd, Tb.dense_border = 1, Tb.dense_f0 = T.np / 6, Tb.bookkeep = 0;
        if (p->dense_border_nb != T.nb) {  // (padding of the dense copy: once per structure, prepare() resets the mark)
          k_dense_border_init<<<64, kBlock, 0, s>>>(Tb.dense, T.nb);
          p->dense_border_nb = T.nb;
        }
      }
      const int n_tiles = (T.nb + kSchurTile - 1) / kSchurTile;
      k_border_schur<<<dim3(n_tiles, n_tiles), kBlock, 0, s>>>(Tb, 0, local_rows, m);  // (rows from the junction on are never skipped)
      if (dense_tail)
        k_dense_solve_mx<<<1, kDxThreads, kDxLdsDoubles * sizeof(double), s>>>(Tb, T.np / 6, p->d_dense_ut.p);
      else
        HIP_TRY(launch_border_solve(Tb, s));
      k_border_apply<<<(T.np + kBlock / 64 - 1) / (kBlock / 64), kBlock, 0, s>>>(Tb);
    }
    Tables T3 = T2;
    T3.join_epoch = ++p->join_epoch;
    const BackJob j0{T.Ub, T.Ubk, T.ybuf, p->d_Vb.p, p->d_yt.p, m + w_mid, 0, 0};
    const BackJob j1{p->d_Ub2.p, p->d_Ubk2.p, p->d_ybuf2.p, p->d_Vb2.p, p->d_yt2.p, mB, w_mid, 1};
    // (the two older sweeps are kept as measurement switches for visual-only systems, the shape they were measured on; they do not write
    //  the border's step outputs)
    const bool sweep_w = HS_AB(T.debug_flags, 65536) && !T.nb && p->vb_len == size_t(T.np) * (6 * T.bw),  // (its pad of zeros sits right behind np x ncb)
                sweep_rows = HS_AB(T.debug_flags, 268435456) && !T.nb;
#if HS_PROFILE_HOOKS
    if (sweep_w) k_premultiply<<<m + w_mid + mB, 128, 0, s>>>(T3, j0, j1, m + w_mid);
#endif
    if (!sweep_w) {  // (A/B switch 65536: single-wave register sweep)
      const size_t g_lds = size_t(6 * (T.bw - 1)) * (6 * (T.bw - 1) | 1) * sizeof(double);  // given-column block of the far sweep
#if HS_PROFILE_HOOKS
      if (sweep_rows)  // A/B switch 268435456: one block row per step
        k_band_backward2<<<2, kCholThreads, 2 * size_t(T.np) * sizeof(double) + g_lds, s>>>(T3, j0, j1, m);
      else
#endif
        // super-blocks of four block rows; the inverses of the diagonal super-blocks come from extra workgroups of the launch
        k_band_backward_sb<<<2 + (m + w_mid + kSb - 1) / kSb + (mB + kSb - 1) / kSb, kCholThreads,
                             std::max((2 * size_t(T.np) + 32) * sizeof(double) + g_lds + sb_phase_a_doubles(T.bw) * sizeof(double),
                                      size_t(3 * kSbN * (kSbN + 1)) * sizeof(double)), s>>>(T3, j0, j1, m, 2, 0);
    } else {
#if HS_PROFILE_HOOKS
      launch_backward_w(T3, j0, j1, m, 2, s);
#endif
    }
    (void)sweep_rows;
    HIP_TRY(hipGetLastError());
    return HS_OK;
  }
  // One-ended. Block rows of the leading constant control points are decoupled (k_factor_decoupled_rows): the dependency chain of the
  // factorisation starts behind them — the same kernels on the trailing sub-matrix (the band storage is row relative: pointer offsets).
  int f0 = 0;
  const bool dense_mx = use_dense_mx(p, &f0);
  Tables Tf = T;
  const int n_eff = n_blk - f0;
  // Small systems (the sliding window's steady state: ~33 free block rows with window-wide bands, bordered with an IMU): factorisation, border
  // and both sweeps in ONE launch, trailing matrix in the accumulators of the f64 matrix cores (kernels_dense_mx.hpp), tiles loaded from the
  // dense copy the finalisation kernels of launch_build wrote. A/B switch 8: the kernels below.
  if (dense_mx) {
    Tables Td = T;
    Td.dense = p->d_dense_ut.p + size_t(kDenseLd) * kDenseLd, Td.dense_f0 = f0;
    Td.bookkeep = p->bookkeep ? 1 : 0;
    k_dense_solve_mx<<<1, kDxThreads, size_t(kDxLdsDoubles) * sizeof(double), s>>>(Td, f0, p->d_dense_ut.p);
    HIP_TRY(hipGetLastError());
    return HS_OK;
  }
  const bool dense = !nt && !(T.debug_flags & 2097152) && T.bw > 14 && n_eff <= 2 * T.bw &&
                     dense_factor_fits(n_eff, std::min(T.bw, n_eff));  // A/B switch 2097152: banded kernels
  if (f0 > 0) {
    if (!dense) k_factor_decoupled_rows<<<f0, 64, 0, s>>>(T, f0);  // (the dense kernel writes them with extra workgroups of its own launch)
    Tf.Sb += size_t(6 * f0) * ncb, Tf.g_s += 6 * f0, Tf.Ub += size_t(6 * f0) * ncb, Tf.Ubk += size_t(24) * f0, Tf.ybuf += 6 * f0, Tf.np -= 6 * f0;
    Tf.fj[0] = FactorJob{Tf.Sb, Tf.g_s, Tf.Ub, Tf.Ubk, Tf.ybuf, nullptr, Tf.np / 6, -1};
  }
  // short systems with window-wide bands (the sliding-window replay): every band tile in a register for the whole factorisation
  if (dense) {
    k_dense_factor<<<1 + f0, kDenseThreads, (size_t(12) * (ncb + 8) + size_t(32) * n_eff) * sizeof(double), s>>>(Tf, f0);
  } else if (nt) {
    Tables T1 = Tf;
    // (the lower-band rows come from the reversed copy, whose rows are counted from the END of the matrix: no offset)
    T1.mj[0] = MfmaJob{p->d_Sb2.p, Tf.g_s, Tf.Ub, Tf.Ubk, Tf.ybuf, nullptr, n_blk - f0, -1, n_blk - f0, INT_MAX, 0, p->d_Vb.p + p->vb_len};
    T1.mj[1] = T1.mj[0];
    HIP_TRY(run_mfma(T1, 1));
  } else if (la_ok && la_ncw == 3)
    k_band_factor_la<1, 3><<<1, la_threads(3), la_lds, s>>>(Tf);
  else if (la_ok)
    k_band_factor_la<1, 4><<<1, la_threads(4), la_lds, s>>>(Tf);
  // (two tiles per lane need 168 accumulator registers: with six waves per workgroup the budget is 256 and the look-ahead
  //  kernel spills in its update loop - wider bands stay on the kernel below)
  else if (T.bw * T.bw <= kCholThreads)
    k_band_factor<1><<<1, kCholThreads + kCholIo, chol_lds, s>>>(Tf);
  else if (T.bw <= 21)  // two tiles per lane; the IO wave moves 12 x 64 entries per block row: 6 (6 bw + 1) <= 768 <=> bw <= 21
    k_band_factor<2><<<1, kCholThreads + kCholIo, chol_lds, s>>>(Tf);  // (bw = 22 dropped entries of every block row in round 1:
                                                                       //  found by the lock-step replay, tests/test_host_driver.py)
  else  // long feature tracks: trailing window in L2 instead of registers
    k_band_factor_wide<<<1, kWideThreads, size_t(12) * (ncb + 2) * sizeof(double), s>>>(Tf);
  if (T.nb) {  // bordered system (bias splines + gravity)
    const int fwd_threads = std::max(128, 64 * ((6 * (T.bw - 1) + 63) / 64));  // one lane per pending row
    // the first non-zero row of a border column follows from the inertial record table — of ALL shards: a shard of a distributed solve
    // only skips the rows of the constant control points (which every shard agrees on)
    const int local_rows = (!p->allreduce && !p->rccl_comm && p->world == 1) ? 1 : 0;
    k_border_forward<<<(T.nb + kBorderCols - 1) / kBorderCols, fwd_threads, size_t(T.np) * kBorderLd * sizeof(double), s>>>(T, f0, local_rows);
    const int nt = (T.nb + kSchurTile - 1) / kSchurTile;
    k_border_schur<<<dim3(nt, nt), kBlock, 0, s>>>(T, f0, local_rows, n_blk);
    HIP_TRY(launch_border_solve(T, s));
    k_border_apply<<<(T.np + kBlock / 64 - 1) / (kBlock / 64), kBlock, 0, s>>>(T);
  }
#if HS_PROFILE_HOOKS
  if ((T.debug_flags & 8192) && !T.nb) {  // A/B: the generalised sweep on the whole system
    const BackJob j0{T.Ub, T.Ubk, T.ybuf, nullptr, nullptr, T.np / 6, 0, 0};
    k_band_backward2<<<1, kCholThreads, 2 * size_t(T.np) * sizeof(double), s>>>(T, j0, j0, -1);
    k_step_outputs<<<1, kBlock, 0, s>>>(T);
  } else
#endif
  if (!HS_AB(T.debug_flags, 65536) || T.nb) {  // (A/B switch 65536: single-wave register sweep — visual-only systems, the shape it was measured on)
    if (6 * (T.bw - 1) <= 96 && !(T.debug_flags & 268435456)) {  // super-blocks of four block rows: one lane pair per pending row, 96 pairs
      Tables T3 = T;
      T3.join_epoch = ++p->join_epoch;
      const BackJob j0{T.Ub, T.Ubk, T.ybuf, p->d_Vb.p, p->d_yt.p, T.np / 6, 0, 0};
      k_band_backward_sb<<<1 + (T.np / 6 + kSb - 1) / kSb, kCholThreads,
                           std::max((2 * size_t(T.np) + 32) * sizeof(double), size_t(3 * kSbN * (kSbN + 1)) * sizeof(double)), s>>>(T3, j0, j0, -1, 1, f0);
    } else {  // wide bands (long feature tracks): one block row per step, one lane per pending row
      k_band_backward<<<1, kCholThreads, 2 * size_t(T.np) * sizeof(double), s>>>(T, f0);
    }
  } else {
#if HS_PROFILE_HOOKS
    const BackJob j0{T.Ub, T.Ubk, T.ybuf, p->d_Vb.p, p->d_yt.p, T.np / 6, 0, 0};
    k_premultiply<<<T.np / 6, 128, 0, s>>>(T, j0, j0, T.np / 6);
    launch_backward_w(T, j0, j0, -1, 1, s);
#endif
  }
  HIP_TRY(hipGetLastError());
  return HS_OK;
}

/// Speculative solves (visual-only windows, on one shard or on every shard of a distributed solve): the candidate is LINEARISED instead of only costed, unless this is the last
/// iteration of the solve: its records land in the record buffer that does not hold the current point and become the current
/// linearisation if the step is accepted (decide_step flips DevState::rec_sel), so that the next iteration starts at k_landmark — after an
/// accepted step and after a rejected one (the records of the unchanged current point are still there: today's path linearises again).
/// One linearise launch (16 us at configs[1]) replaces a cost launch (7.7 us) + a linearise launch per iteration.
static bool speculative_solve(const hs_problem* p) {
  const Tables& T = p->T;
  return !p->fused && T.n_vis > 0 && !T.n_pri && !T.n_ine && !T.nb && !(T.debug_flags & 1073741824);  // A/B switch (shards of a distributed solve too: the decision is replicated)
}
/// Fused build: a visual-only window keeps an accepted candidate in the candidate buffers (k_build_visual reads it from there, the next
/// k_backsub_retract copies it to x on its way): no k_commit launch per iteration. Other windows commit (their prior / inertial kernels read x).
static bool fused_visual_only(const hs_problem* p) {
  const Tables& T = p->T;
  return p->fused && !T.n_pri && !T.n_ine && !T.nb;
}

/// Fused visual-only windows on one shard with deferred landmarks (k_pack_decision(3): decide + commit the control points): the decision of an
/// iteration that another one follows is taken by workgroup 0 of that iteration's k_build_visual — nothing else reads the solver state or the
/// current point between the update and the build, so the one-workgroup kernel and its launch boundary (11 us) leave the chain.
/// A/B switch 32768: k_pack_decision behind every update.
static bool fold_decision_into_build(const hs_problem* p, bool deferred_commit) {
  return fused_visual_only(p) && deferred_commit && !p->allreduce && !p->rccl_comm && p->world == 1 && !(p->T.debug_flags & 32768);
}

/// Small problems: the decision kernel copies the accepted candidate to x itself (single shard). A/B switch 16777216: always k_commit.
static bool commit_inline(const hs_problem* p) {
  const Tables& T = p->T;
  return !p->allreduce && !p->rccl_comm && 8 * T.sp.n_cp + 3 * T.n_lm + 8 * T.n_bias <= kCommitInline && !(T.debug_flags & 16777216);
}

template <int K>
int launch_update(hs_problem* p, bool linearize_candidate = false, bool deferred_commit = false, hipEvent_t* lin_events = nullptr, bool fold_next = false) {
  const Tables& T = p->T;
  hipStream_t s = p->stream;
  const bool local_decision = !p->allreduce && !p->rccl_comm;  // single shard: decide in the packing kernel
  const bool inline_commit = commit_inline(p) && !deferred_commit;  // (a deferred commit: the control points only, decide_here = 3)
  const bool cps_here = p->fused && deferred_commit;  // fused path: the decision kernel commits the control points, the landmarks stay deferred
  const int decide_here = inline_commit ? 2 : local_decision ? (cps_here ? 3 : 1) : 0;
  // Single shard, fused path (round 6): the candidate costs of the prior / inertial factors are further workgroups of k_update_visual's launch —
  // one launch where a stereo-inertial window had three on the chain of every iteration. A/B switch 134217728: the separate launches.
  const bool merged = p->fused && local_decision && !(T.debug_flags & 134217728);
  if (p->fused) {  // candidate point, landmark back-substitution and the visual candidate cost per chunk, one launch
    const int nb_pri = merged && T.n_pri ? p->nb_pri : 0, nb_ine = merged && T.n_ine ? p->nb_ine : 0;
    const size_t lds = std::max(size_t(update_lds_doubles(T.bw, p->build_R, p->build_L)), size_t(8 * T.sp.n_cp + 8 * T.n_bias + 4)) * 8;
    k_update_visual<K><<<p->nb_vis + T.n_norm_part + nb_pri + nb_ine, kBlock, lds, s>>>(T, p->build_R, p->build_L, p->nb_vis, nb_pri, nb_ine);
    if (T.n_pri && !merged) k_cost_prior<K><<<p->nb_pri, kBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.cand_part + p->nb_vis);
    if (T.n_ine && !merged)
      k_cost_inertial<K, 4><<<p->nb_ine, kInertialBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.bias_g_cand, T.bias_a_cand, T.gravity_cand,
                                                                          T.cand_part + p->nb_vis + p->nb_pri);
  } else
    k_backsub_retract<<<T.n_lm_part + T.n_norm_part, kBlock, 0, s>>>(T);
  if (p->fused) {
  } else if (linearize_candidate) {
    if (lin_events) HIP_TRY(hipEventRecord(lin_events[0], s));  // stage timing: this launch is booked under "linearise", not "update"
    k_linearize_visual<K><<<p->nb_vis, lin_block<K>(), lin_lds_bytes<K>(p), s>>>(T, nullptr, T.v_pos, 1, T.cand_part, nullptr, T.cp_cand, T.lm_cand);
    if (lin_events) HIP_TRY(hipEventRecord(lin_events[1], s));
  } else if ((T.n_ine || T.n_pri) && !(T.debug_flags & 33554432)) {  // one launch for all factor types (A/B switch 33554432: one per type)
    k_cost_all<K, 4><<<p->nb_vis + p->nb_pri + p->nb_ine, kBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.lm_cand, T.bias_g_cand, T.bias_a_cand, T.gravity_cand,
                                                                                       T.cand_part, p->nb_vis, p->nb_pri);
  } else {
    if (T.n_vis) k_cost_visual<K><<<p->nb_vis, kBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.lm_cand, T.cand_part);
    if (T.n_pri) k_cost_prior<K><<<p->nb_pri, kBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.cand_part + p->nb_vis);
    if (T.n_ine)
      k_cost_inertial<K, 4><<<p->nb_ine, kInertialBlock, cp_lds_bytes(p), s>>>(T, T.cp_cand, T.bias_g_cand, T.bias_a_cand, T.gravity_cand,
                                                                          T.cand_part + p->nb_vis + p->nb_pri);
  }
  if (fold_next) return HS_OK;  // (fold_decision_into_build: the next iteration's k_build_visual decides, launch_build(..., fold = true))
  k_pack_decision<<<1, kBlock, 0, s>>>(T, decide_here);
  HIP_TRY(hipGetLastError());
  const int rc = exchange(p, T.xbuf + T.xo_dec, 5);  // candidate cost + norms + landmark-side model-cost terms
  if (rc) return rc;
  if (!local_decision) k_decide<<<1, kBlock, 0, s>>>(T, cps_here ? 1 : 0);
  const int nb_commit = std::max((std::max(8 * T.sp.n_cp, 3 * T.n_lm) + kBlock - 1) / kBlock, 1);  // one element per lane
  // (deferred: speculative solves of larger problems — the next iteration's k_backsub_retract copies the accepted candidate to x on its way,
  //  hs_solve launches k_commit once behind the last iteration)
  if (!inline_commit && !deferred_commit) k_commit<<<nb_commit, kBlock, 0, s>>>(T);
  HIP_TRY(hipGetLastError());
  return HS_OK;
}

static void launch_commit(hs_problem* p) {
  const Tables& T = p->T;
  const int nb_commit = std::max((std::max(8 * T.sp.n_cp, 3 * T.n_lm) + kBlock - 1) / kBlock, 1);
  k_commit<<<nb_commit, kBlock, 0, p->stream>>>(T);
}

/// First use of a kernel costs ~0.35 ms of host time (the runtime builds its kernel object lazily); a solve touches ~25 different kernels,
/// which showed up as a 9 ms hs_solve on the first optimize() of a process (HS_HOST_TIMING=2: "launches" of call 0). hs_create resolves
/// the kernels of the solve path up front, once per process and device; what remains on the first call is the allocation of the tables.
template <int K>
static void warm_kernels_of_order() {
  hipFuncAttributes fa;
  const void* kernels[] = {
      reinterpret_cast<const void*>(&k_build_visual<K>), reinterpret_cast<const void*>(&k_update_visual<K>), reinterpret_cast<const void*>(&k_linearize_visual<K>), reinterpret_cast<const void*>(&k_linearize_prior<K>),
      reinterpret_cast<const void*>(&k_linearize_inertial<K, 4>), reinterpret_cast<const void*>(&k_landmark<K, 2, 2>),
      reinterpret_cast<const void*>(&k_landmark<K, 4, 1>), reinterpret_cast<const void*>(&k_landmark_rows<K, 4>),
      reinterpret_cast<const void*>(&k_gram_pair<K, 1>), reinterpret_cast<const void*>(&k_gram_pair<K, 2>), reinterpret_cast<const void*>(&k_gram_pair<K, 4>),
      reinterpret_cast<const void*>(&k_seg_gram<K>), reinterpret_cast<const void*>(&k_assemble<K>), reinterpret_cast<const void*>(&k_assemble_wide<K>), reinterpret_cast<const void*>(&k_border_pb<K>),
      reinterpret_cast<const void*>(&k_border_bb<K>), reinterpret_cast<const void*>(&k_cost_visual<K>), reinterpret_cast<const void*>(&k_cost_prior<K>),
      reinterpret_cast<const void*>(&k_cost_inertial<K, 4>), reinterpret_cast<const void*>(&k_cost_all<K, 4>),
      reinterpret_cast<const void*>(&k_process_tracks<K>), reinterpret_cast<const void*>(&k_sample_trajectory<K>)};
  for (const void* k : kernels) (void)hipFuncGetAttributes(&fa, k);
}
static void warm_kernels(int device) {
  static std::mutex mu;
  static std::vector<int> done;
  std::lock_guard<std::mutex> lock(mu);
  if (std::find(done.begin(), done.end(), device) != done.end()) return;
  done.push_back(device);
  hipFuncAttributes fa;
  const void* kernels[] = {
      reinterpret_cast<const void*>(&k_group_gram<1>), reinterpret_cast<const void*>(&k_group_gram<2>), reinterpret_cast<const void*>(&k_group_gram<4>),
      reinterpret_cast<const void*>(&k_pack_exchange), reinterpret_cast<const void*>(&k_cost_reduce), reinterpret_cast<const void*>(&k_finalize_reduced),
      reinterpret_cast<const void*>(&k_finalize_border), reinterpret_cast<const void*>(&k_reduce_partials), reinterpret_cast<const void*>(&k_factor_decoupled_rows),
      reinterpret_cast<const void*>(&k_dense_factor), reinterpret_cast<const void*>(&k_dense_solve_mx), reinterpret_cast<const void*>(&k_band_factor_wide), reinterpret_cast<const void*>(&k_band_factor<1>),
      reinterpret_cast<const void*>(&k_band_factor<2>), reinterpret_cast<const void*>(&k_band_factor_la<1, 3>), reinterpret_cast<const void*>(&k_band_factor_la<1, 4>),
      reinterpret_cast<const void*>(&k_band_backward), reinterpret_cast<const void*>(&k_band_backward_sb), reinterpret_cast<const void*>(&k_border_forward),
      reinterpret_cast<const void*>(&k_border_forward2),
      reinterpret_cast<const void*>(&k_border_schur), reinterpret_cast<const void*>(&k_border_solve), reinterpret_cast<const void*>(&k_border_solve_reg<3>), reinterpret_cast<const void*>(&k_border_solve_reg<4>),
      reinterpret_cast<const void*>(&k_border_solve_reg<5>), reinterpret_cast<const void*>(&k_border_solve_reg<6>), reinterpret_cast<const void*>(&k_border_solve_reg<7>),
      reinterpret_cast<const void*>(&k_border_solve_reg<8>), reinterpret_cast<const void*>(&k_border_apply), reinterpret_cast<const void*>(&k_backsub_retract),
      reinterpret_cast<const void*>(&k_pack_decision), reinterpret_cast<const void*>(&k_decide), reinterpret_cast<const void*>(&k_commit),
      reinterpret_cast<const void*>(&k_reset_state), reinterpret_cast<const void*>(&k_scatter_uploads)};
  for (const void* k : kernels) (void)hipFuncGetAttributes(&fa, k);
  warm_kernels_of_order<4>();
  warm_kernels_of_order<5>();
  warm_kernels_of_order<6>();
}

int set_func_attributes(hs_problem* p) {
  // opt in to > 64 KiB dynamic LDS for the factorisation
  hipFuncAttributes fa;
  HIP_TRY(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_band_factor<2>)));
  p->chol_lds_max = 160 * 1024 - int(fa.sharedSizeBytes);
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor<1>), hipFuncAttributeMaxDynamicSharedMemorySize, p->chol_lds_max));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor<2>), hipFuncAttributeMaxDynamicSharedMemorySize, p->chol_lds_max));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_la<1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, p->chol_lds_max));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_factor_la<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, p->chol_lds_max));
  for (int k = 4; k <= 6; ++k) HS_ORDER_SWITCH(k, {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seg_gram<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_pair<K, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_pair<K, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gram_pair<K, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_visual<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_update_visual<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_linearize_visual<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  });
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_solve_mx), hipFuncAttributeMaxDynamicSharedMemorySize, int(kDxLdsDoubles * sizeof(double))));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_solve_reg<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_solve_reg<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_solve_reg<7>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_solve_reg<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_forward), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_forward2), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
#if HS_PROFILE_HOOKS
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_backward2), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
#endif
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_backward_sb), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_band_backward), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_border_solve), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  return HS_OK;
}

}  // namespace
