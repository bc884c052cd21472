// problem.hpp — HBM-resident window tables and the device-side LM state shared by kernels.hip / capi.hip.
//
// The reference keeps the window as a ceres::Problem pointer graph that is walked one residual block at a time on
// one host thread (/root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:41,278). Here the same content is a
// set of flat, sorted tables:
//   control points   n_cp x 8  [qx qy qz qw px py pz t]                       (Stamped<SE3>, stamped.hpp:35-36)
//   cameras          n x 16    [T_bs(7) | cx cy fx fy | k1 k2 p1 p2 | pad]     (sensors/camera.cpp:30-48)
//   landmarks        n x 3, device order = sorted by first control point touched
//   visual residuals landmark-major: stamp, meas[3], landmark, camera|type, first control point, record slot
//   records          one per residual block, segment-major (sorted by first control point):
//                      visual  [r(2) | J_landmark(2x3) | J_state(2 x 6k)]      = 8 + 12k doubles  (B_out of SURVEY.md §8d)
//                      prior   [r(6) | J_state(6 x 6k)]                        = 6 + 36k doubles
//   reduced system   block-banded upper storage: row rho holds S[rho][6*(rho/6) + c], c in [0, 6*BW)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hyperslam_hip.h"
#include "device_math.hpp"

namespace hs {

constexpr int kMaxIterations = 64;
constexpr int kCamStride = 16;

struct Spline {
  int k, n_cp;
  double t0, dt, inv_dt;
  int rot_const, trans_const;
};

/// LM state machine living in device memory (TrustRegionMinimizer + LevenbergMarquardtStrategy).
struct DevState {
  double radius, decrease_factor;
  double cost;            // cost at the current point
  double cand_cost;       // cost at the candidate point
  double local_cost;      // this shard's part of `cost` (speculative solves: the cost partials of the current point are not recomputed)
  double local_cand;      // this shard's part of the candidate's cost (k_pack_decision, before the exchange)
  double model_cost_change;
  unsigned long long gmax_bits;       // landmark-side max |gradient| as raw bits (atomicMax on non-negative doubles)
  unsigned long long gmax_pose_bits;  // pose-side max |gradient| (after the exchange)
  double gmax;            // gradient max norm of the current linearisation
  double x_sqnorm, step_sqnorm;
  // reduction scratch for the model cost change (scaled coordinates)
  double g_dot_step_pose, d2_step2_pose;
  double g_dot_step_far, d2_step2_far;  // the far end's share of the two sums (two-ended super-block sweep: each end reduces its own rows); 0 on every other path
  int iteration;          // index of the LM iteration being executed (1-based; 0 = initial evaluation)
  int done;               // 1 once a termination criterion fired: all later kernels early-exit
  int termination;
  int accepted;           // decision of the current iteration (consumed by k_commit)
  int step_valid;
  int invalid_streak;
  int num_successful;
  int num_iterations;
  int scaling_ready;      // Jacobi scaling computed (iteration 0 only)
  int max_iterations;
  int chol_failed;
  int spec;               // 1, 2: this solve linearises at the CANDIDATE point (visual-only windows, record path): see launch_update; 2, 4: the accepted
                          // candidate stays in the candidate buffers until the next k_backsub_retract copies it on its way (deferred commit)
  int rec_sel;            // which visual record buffer holds the linearisation of the current point (0: v_rec, 1: v_rec_alt)
  int rec_pending;        // the other buffer holds the linearisation of the candidate awaiting its decision
  hs_iteration records[kMaxIterations + 1];
};

/// Everything a kernel needs, passed by value (pointers into HBM).
/// One banded factorisation job of k_band_factor_la (two jobs when the reduced system is factored from both ends at once).
struct FactorJob {
  const double* Sb;   // band rows of the system (natural order for job 0, reversed order for job 1)
  const double* g_s;  // right-hand side
  double* Ub;         // factor rows out
  double* Ubk;        // inverted diagonal blocks out
  double* ybuf;       // forward-solved right-hand side out
  double* win;        // job 1: trailing window after n_steps (w x 6 x (ncb + 1)); job 0: the other end's window to merge
  int n_steps;        // block rows to factor
  int merge_at;       // job 0, two-ended mode: first block row of the middle part (-1: none)
};

constexpr int kProgressStride = 32;  // unsigned words between the progress words of a job (MfmaJob::progress): a cache line each

/// One job of k_band_factor_mfma (kernels_factor_mfma.hpp).
struct MfmaJob {
  const double* L;   // lower-band rows in this job's ordering: row np-1-sigma of the OTHER ordering's upper band array, columns reversed
  const double* g;   // right-hand side (own order)
  double* Ub;        // factor rows out (own order, band storage)
  double* Ubk;       // inverted diagonal blocks out
  double* ybuf;      // forward-solved right-hand side out
  double* win;       // two-ended: junction buffer dm x (dm + 1) (dm = 6 (bw - 1)), natural middle-local coordinates of job 0
  int n_steps;       // block rows to eliminate
  int merge_at;      // job 0, two-ended: first middle block row (-1: none)
  int enter_limit;   // block rows >= enter_limit never enter (other end's territory / past the matrix): zeros
  int zero_from;     // job 1, two-ended: pairs with both block rows >= zero_from enter as zeros (INT_MAX: none)
  int dump;          // job 1, two-ended: hand the trailing window over at the end
  const double* zero;  // a double 0.0 in device memory: source of entries that enter as zeros (no select on loaded values)
  // Bordered systems: k_border_forward2 sweeps the border columns WHILE this job factors (side stream). progress[0] / progress[kProgressStride] =
  // progress_base + number of leading block rows whose factor row / inverted diagonal block is complete in memory at agent scope (the rows
  // are then written with agent-scope stores). nullptr: nobody follows.
  unsigned* progress;
  unsigned progress_base;
};

struct Tables {
  Spline sp;
  hsd::BasisCoef basis;
  // control points
  double* cp;
  double* cp_cand;
  const uint8_t* cp_const;
  // cameras / plain sensors
  const double* cam;       // n_cam x 16
  int n_cam;
  const double* sensor;    // n_sensor x 8 (T_bs 7 + pad)
  // landmarks (device order)
  int n_lm;
  double* lm;
  double* lm_cand;
  const uint8_t* lm_const;
  const int* lm_ptr;       // n_lm + 1 : range of visual residuals (landmark-major)
  const int* lm_cfirst;    // first control point touched
  const int* lm_ncp;       // number of control points touched
  const int* lm_yoff;      // offset (in doubles) of Y-hat rows in `Y`
  const int* cf_ptr;       // n_cp + 1 : first device landmark with c_first >= c
  double* lm_scale;        // n_lm x 3 Jacobi scaling
  double* lm_L;            // n_lm x 6 Cholesky factor of V (l00 l10 l11 l20 l21 l22)
  double* lm_yhat;         // n_lm x 3  L^-1 (s_l o b_l)
  double* lm_sb;           // n_lm x 3  s_l o b_l
  double* lm_D2;           // n_lm x 3  LM diagonal
  double* lm_part;         // n_lm_part x 4  per-workgroup (|x|^2, |x - x+|^2, g.step, step D2 step) landmark terms of the decision
  int n_lm_part;
  double* lm_gmax;         // n_lm      per-landmark max |b_l| (gradient max norm)
  int n_obs_lm;            // observed landmarks (device order puts unobserved ones last)
  double* Y;               // concatenated Y-hat (6 n_l x 3 per landmark)
  // visual residuals (landmark-major)
  int n_vis;
  const double* v_stamp;
  const double* v_meas;    // n x 3
  const int* v_lm;
  const int* v_info;       // camera | type << 16
  const int* v_first;
  const int* v_pos;        // record slot (segment-major)
  double* v_rec;           // n x (8 + 12k)
  double* v_rec_alt;       // second buffer of the same size (speculative linearisation at the candidate point)
  const int* v_seg_ptr;    // n_seg + 1 over record slots
  // prior residuals (segment-major == record order)
  int n_pri;
  const double* p_stamp;
  const double* p_meas;    // n x 7
  const int* p_sensor;
  const int* p_first;
  double* p_rec;           // n x (6 + 36k)
  const int* p_seg_ptr;
  int n_seg;               // n_cp - k + 1
  // inertial residuals (segment-major == record order) + IMU / bias splines / gravity (SURVEY a-4)
  int n_ine;
  const double* i_stamp;
  const double* i_meas;    // n x 6
  const int* i_first;      // first state control point
  const int* i_first_bias; // first bias control point
  double* i_rec;           // n x (18 + 36k + 2kb): [r(6) | J_state(6 x 6k) | wg(kb) | wa(kb) | J_gravity(6x2)]
  const int* i_seg_ptr;
  const struct ImuParams* imu;
  hsd::BasisCoef bias_basis;
  int kb, n_bias;
  double bias_t0, bias_dt;
  double* bias_g;          // n_bias x 4 [x y z t]
  double* bias_a;
  double* bias_g_cand;
  double* bias_a_cand;
  double* gravity;         // 3
  double* gravity_cand;
  int bias_const, gravity_const;
  int inertial_literal;  // Jacobian of the inertial factor as written upstream (inertial.cpp:131-198) | 0: derivative of the prediction
  int nb;                  // border unknowns: 6 n_bias + 2 (0 without an IMU)
  // reduced system
  int bw;                  // band width in blocks
  int np;                  // 6 * n_cp
  double* scale_p;         // np
  double* Sb;              // np x (6 bw)   scaled + damped band (input of the factorisation)
  double* Ub;              // np x (6 bw)   Cholesky factor (upper, band rows)
  double* Ubk;             // n_cp x 24  inverse of the factored diagonal blocks (packed upper) for the backward sweep
  double* g_s;             // np  reduced scaled gradient
  double* g_full;          // np  scaled full gradient s_p o g_p
  double* D2p;             // np  LM diagonal (pose side)
  double* gabs;            // np + nb  |unscaled gradient| (max-reduced into the gradient tolerance test)
  double* step_p;          // np  scaled step
  double* delta_p;         // np  unscaled step (tangent update)
  double* ybuf;            // np  y = U^-T g (forward-solved right-hand side)
  double* ybuf2;           // two-ended factorisation: the far end's part of y, reversed order (kernels_border.hpp y_slot)
  int y_split;             // rows below y_split are in ybuf (np unless the factorisation runs from both ends)
  // border blocks (scaled + damped) and the bordered solve
  double* scale_b;         // nb
  double* Spb;             // np x nb
  double* Sbb;             // nb x nb
  double* gb_s;            // nb
  double* D2b;             // nb
  double* Zb;              // np x nb   U^-T S_pb
  double* Cb;              // nb x nb   S_bb - Z'Z
  double* hb;              // nb
  double* xb;              // nb        border solution
  double* delta_b;         // nb        unscaled border step
  const int* i_bias_ptr;   // n_bias + 1: first inertial record with first_bias >= f
  const int* bfwd_start;   // per workgroup of k_border_forward: first block row in which one of its border columns is non-zero
  // reductions
  double* cost_part;       // per-block cost partial sums (current point)
  double* cand_part;       // per-block cost partial sums (candidate point)
  int n_cost_part;
  double* norm_part;       // per-block (x_sqnorm, step_sqnorm) pairs of the replicated unknowns
  int n_norm_part;
  // exchange buffer (additive across residual shards): [Sraw np*6bw | g_p np | g_schur np | diag np | H_pb np*nb | H_bb nb*nb | g_b nb | cost | gmax[world] | decision 5]
  double* xbuf;
  const int* sw_ptr;  // n_seg + 1: workgroups of k_seg_gram serving segment f (splits ~ record count)
  const int* sw_seg;  // segment of workgroup w
  FactorJob fj[2];       // k_band_factor_la jobs (blockIdx.x)
  MfmaJob mj[2];         // k_band_factor_mfma jobs (blockIdx.x)
  double* Sb2;           // reversed copy of Sb / g_s (nullptr unless the two-ended factorisation will run)
  double* g2;
  double* xsol;          // np: solution of the reduced system in natural order (two-ended path)
  unsigned* join_flag;   // device word: epoch of the last finished bottom-end factorisation / published middle solution
  unsigned join_epoch;
  unsigned gather_epoch; // != 0: the border gathers on the side stream end with this value in join_flag[kGatherFlag] (k_border_bb's last workgroup), and the
                         // border workgroups of k_finalize_reduced wait for it there instead of the host enqueuing an event between the two streams
  double* gravity_part;  // n_bias x 5: gravity block partials of k_border_bb
  double* segP;   // per k_seg_gram workgroup: [J'J (6k x 6k) | J'r (6k)]
  const int* gw_ptr;  // n_cp + 1: workgroups of k_group_gram serving landmark group c (splits ~ landmark count)
  const int* gw_cf;   // group of workgroup w
  double* Qw;     // wide_q: the landmark term of the whole window, -sum_l Yh_l Yh_l', in band-row storage [np][6 bw] followed by -sum_l Yh_l yh_l [np] (k_landmark_gram_wide)
  double* Yt;     // wide_q: second copy of Y-hat for k_landmark_gram_wide, [control point][9 pairs of doubles][yt_stride landmarks][2] — consecutive landmarks in consecutive lanes
  int yt_stride;  // landmarks per row of Yt (observed landmarks rounded up to 64)
  int wide_q;     // fused build on window-wide bands: the chunk partials carry J_p'J_p only (band tiles); the landmark term is formed once per window from Y-hat
  double* grpQ;   // per k_group_gram workgroup: [upper 6x6 tiles of -sum Yh Yh' | -sum Yh yh (6 bw)]
  double* xpart;  // per-split partial copies of the H_pb part of the exchange buffer (stride x_count1); scratch for timestamps
  int xo_g, xo_gs, xo_dj, xo_pb, xo_bb, xo_gb, xo_cost, xo_gmax, xo_dec, x_count1;
  // fused build of the visual factors (kernels_build.hpp): chunk w = device landmarks [ch_ptr[w], ch_ptr[w + 1]) of one landmark group;
  // gw_ptr / gw_cf then list the chunks of a group and grpQ holds one partial [tiles | -Yh yh | J_p'r | diag J_p'J_p] per chunk
  int fused, n_chunk;
  int build_stream_lg;   // fused build: log2 of the record streams per band tile, two bits per diagonal offset (kernels_build.hpp: build_streams)
  double* ch_gmax;       // fused build: max |J_l' r| over the landmarks of chunk w (what the direct bookkeeping reads instead of 5 000 per-landmark values)
  int bookkeep;          // k_band_factor_la: the iteration bookkeeping (cost, gradient max norm, termination tests) is done in the factorisation's
                         // prologue — k_assemble wrote the scaled, damped system itself and k_finalize_reduced was not launched (launch_build)
  int fold_decision;     // k_build_visual: workgroup 0 of the launch is the trust-region decision of the PREVIOUS iteration (k_pack_decision was not
                         // launched behind its update); the chunk workgroups 1 .. wait for join_flag[kFoldFlag] >= fold_epoch before they read the
                         // solver state and the current point (launch_update / launch_build)
  unsigned fold_epoch;
  const int* ch_ptr;
  const int* ch_desc;  // n_chunk x 8: first landmark, landmarks, first control point, first residual, residuals (one 32-byte load per workgroup)
  double* dense;   // k_dense_solve_mx will solve this linearisation (launch_build): k_finalize_reduced / finalize_border_body also write the scaled,
                   // damped system as ONE dense row-major matrix (leading dimension 256, both triangles, identity on the padding to whole 16 x 16
                   // tiles) of the free block rows dense_f0 .. and the border unknowns; nullptr otherwise
  int dense_f0;
  unsigned sweep_epoch;  // != 0: k_border_forward2 (side stream) ends with this value in join_flag[kGatherFlag + 2] and k_border_schur waits for it there
                         // (no event between the streams behind the pipelined border sweep)
  int dense_border;      // k_dense_solve_mx on the border Schur complement of a two-ended bordered system (launch_factor): Tables::dense holds C | h, the
                         // kernel's only output is x_b (+ the factorisation's verdict)
  int rank, world;
  int debug_flags;  // HS_DEBUG_FLAGS env (timing experiments; 0 in production)
  DevState* st;
};

}  // namespace hs
