// kernels.hpp — gfx950 kernels of one Levenberg-Marquardt iteration (included once by capi.hip).
//
// Launch sequence per iteration (one HIP stream + one side stream, no host synchronisation; every kernel early-exits once the
// device state machine has terminated):
//   k_linearize_visual / _prior / _inertial  one residual block per lane -> segment-major records + cost partials
//   k_landmark                               one wave per landmark: H_ll, b_l, W_l -> damped 3x3 Cholesky -> Y-hat, y-hat
//   k_seg_gram / k_group_gram / k_assemble   reduced system from per-segment J'J and per-landmark-group Y-hat Y-hat' partials
//   k_border_pb / k_border_bb                border blocks of the inertial factors (bias splines, gravity)
//   k_pack_exchange -> [all-reduce] -> k_finalize_reduced / _border -> k_cost_reduce     scaling, damping, gradient test
//     (single shard without border unknowns: packing + bookkeeping are an extra workgroup of k_finalize_reduced, one launch)
//   k_band_factor_mx (two ends, bw <= 16: trailing window in f64-MFMA accumulators) | k_band_factor_la (one end; A/B) | k_band_factor
//   (bw <= 22) | k_band_factor_wide (bw <= 42) | k_dense_factor                                                             S = U'U, y
//   k_border_forward / _schur / _solve / _apply                                         bordered part of the solve
//   k_band_backward | k_band_backward_sb     U x = y (two-ended: super-blocks of four block rows, inverses built by extra workgroups), step outputs
//   k_backsub_retract                        step for landmarks, candidate point = Plus(x, delta), norm / model-cost partials
//   k_cost_visual / _prior / _inertial       cost at the candidate point
//   k_pack_decision -> [all-reduce] -> k_decide -> k_commit      trust-region logic and acceptance
#pragma once
#include "kernels_common.hpp"
#include "kernels_linearize.hpp"
#include "kernels_sensor.hpp"
#include "kernels_schur.hpp"
#include "kernels_build.hpp"
#include "kernels_border.hpp"
#include "kernels_factor.hpp"
#include "kernels_factor_mx.hpp"
#include "kernels_dense_mx.hpp"
#include "kernels_backward_sb.hpp"
#if HS_PROFILE_HOOKS  // measured alternatives (A/B switches of HS_DEBUG_FLAGS): profiling builds only, the product library carries one path per band class
#include "../../tools/ab/kernels_backward2.hpp"
#include "../../tools/ab/kernels_backward.hpp"
#include "../../tools/ab/kernels_factor_mfma.hpp"
#endif
#include "kernels_update.hpp"
#include "kernels_aux.hpp"
