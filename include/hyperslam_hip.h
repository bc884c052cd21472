/* hyperslam_hip.h — C ABI of the MI355X-native continuous-time NLLS backend (libhyperslam_hip.so).
 *
 * This is the drop-in boundary for HyperSLAM's optimisation hot path: everything that happens
 * inside `Optimizer<OptimizerSuite::CERES>::optimize()` -> `ceres::Solve`
 * (/root/reference/internal/hyper/optimizers/ceres/optimizer.cpp:276-280) and, per residual block, inside
 * `ExteroceptiveCost<CERES>::Evaluate` (/root/reference/internal/hyper/optimizers/ceres/costs/exteroceptive.cpp:101-160)
 * -> `Evaluator<Obs, SE3>::evaluate` (the four files under /root/reference/internal/hyper/optimizers/evaluators/).
 *
 * The reference walks a pointer graph (ceres::Problem) one residual at a time; this ABI takes the same content as
 * flat tables (control points, sensors, landmarks, per-type residual records), keeps them resident in HBM and runs
 * linearise -> robustify -> landmark Schur complement -> reduced solve -> retract -> accept/reject on the GPU.
 *
 * Conventions
 *   - plain C, opaque handle, every call returns int (0 = HS_OK); hs_last_error(handle) gives the message.
 *   - host buffers are caller-owned and only read/written during the call; device memory is library-owned.
 *   - a handle is used by one thread at a time (the reference calls its optimizer from the single backend thread,
 *     /root/reference/internal/hyper/system/components/backend.cpp:143-145); no internal host threads.
 *   - all arithmetic fp64 (/root/reference/include/hyper/optimizers/ceres/manifolds/variables/wrapper.hpp:22).
 *   - quaternions are (x, y, z, w) (su2.cpp:21, settings.yaml:34-36); a control point is the reference's
 *     Stamped<SE3> block [qx qy qz qw px py pz t] (stamped.hpp:35-36, se3.cpp:20-23).
 */
#ifndef HYPERSLAM_HIP_H_
#define HYPERSLAM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HS_OK 0
#define HS_ERR_INVALID 1   /* bad argument / inconsistent tables */
#define HS_ERR_DEVICE 2    /* HIP runtime error */
#define HS_ERR_STATE 3     /* call order (e.g. solve before tables are set) */
#define HS_ERR_NUMERIC 4   /* non-finite values / factorisation failure surfaced to the caller */
#define HS_ERR_KNOTS 5     /* hs_set_spline: the control-point stamps are not t0 + j*dt (a hole or a shifted knot); nothing was changed */

/* Factor types = the four in-tree evaluators (optimizer.cpp:189,212,234,253). */
#define HS_PIXEL 0     /* VisualPixelEvaluator   pixel.cpp:16     2 rows, CartesianMetric, Huber(0.5)      optimizer.cpp:226 */
#define HS_BEARING 1   /* VisualBearingEvaluator bearing.cpp:14   1 row,  AngularMetric,   Huber(1.6e-3)   optimizer.cpp:204 */
#define HS_PRIOR 2     /* ManifoldEvaluator      manifold.cpp:12  6 rows, ManifoldMetric,  no loss         optimizer.cpp:250 */
#define HS_INERTIAL 3  /* InertialEvaluator      inertial.cpp:13  6 rows, CartesianMetric, Scaled(1.6e-5)  optimizer.cpp:267 */

typedef struct hs_problem hs_problem;

/* Termination (ceres::TerminationType as used by TrustRegionMinimizer). */
#define HS_NO_CONVERGENCE 0 /* max_num_iterations reached (optimizer.cpp:40: 5) */
#define HS_CONVERGENCE 1
#define HS_FAILURE 2

typedef struct hs_iteration {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_successful;
  int32_t reserved;
  double cost;              /* cost after this iteration (0.5 * sum rho(|r|^2)) */
  double cost_change;
  double gradient_max_norm; /* max |gradient| in local coordinates after this iteration's step; for the LAST executed iteration: before it */
  double step_norm;
  double relative_decrease;
  double radius;            /* trust-region radius after this iteration */
} hs_iteration;

typedef struct hs_summary {
  double initial_cost;
  double final_cost;
  int32_t num_iterations;        /* LM iterations executed (linear solves), <= max_iterations */
  int32_t num_successful_steps;
  int32_t termination;           /* HS_NO_CONVERGENCE | HS_CONVERGENCE | HS_FAILURE */
  int32_t num_residual_blocks;   /* residual blocks evaluated per linearisation on this handle */
  double linearize_ms;           /* accumulated device time per stage over the solve (HIP events); the four stage fields are -1 unless hs_set_stage_timing is on */
  double schur_ms;
  double solve_ms;
  double update_ms;
  double total_ms;
} hs_summary;

/* Output pointers for hs_linearize (each nullable). Rows are residual blocks in table order.
 * Jacobians are Ceres *local* Jacobians (J_ambient * PlusJacobian — the quantity the reference's own tests compare,
 * tests/include/tests/optimizers/evaluators/evaluator.hpp:52), row-major per residual block. */
typedef struct hs_linearization {
  double* r;            /* n x n_res                                                           */
  double* J_state;      /* n x n_res x 6k   columns: k control points x [d_rot(3) d_trans(3)]   */
  double* J_landmark;   /* n x n_res x 3    (pixel / bearing)                                   */
  double* J_bias_g;     /* n x 6 x 3kb      (inertial)                                          */
  double* J_bias_a;     /* n x 6 x 3kb      (inertial)                                          */
  double* J_gravity;    /* n x 6 x 2        (inertial; SphereManifold<3> tangent basis of Ceres) */
  int32_t* first_cp;    /* n   index of the first of the k control points used                  */
  int32_t* first_bias;  /* n   (inertial)                                                       */
  double* cost;         /* n   0.5 * rho(|r|^2)                                                 */
  /* Sensor parameter blocks (static_sensor_idx .. dynamic_sensor_idx of exteroceptive.cpp:25-99). The solver keeps them constant
   * (camera.hpp:18, imu.hpp:18, optimizer.cpp:59-64: Ceres passes nullptr); the evaluators fill them whenever the pointer is
   * non-null (bearing.cpp:74, pixel.cpp:91-135,141, manifold.cpp:57, inertial.cpp:155-194), which is what the reference's own
   * tests exercise (tests/.../evaluators/evaluator.hpp:38-65). Produced by a kernel of their own: the solver's kernels do not
   * carry them. Ceres-local like the others (extrinsics: Product(EigenQuaternion, R3) manifold, sensors/sensor.cpp:26-29). */
  double* J_extrinsics;        /* n x n_res x 6   T_bs [d_rot(3) d_trans(3)]            (all four factors) */
  double* J_intrinsics;        /* n x 2 x 4       [cx cy fx fy]                          (pixel)            */
  double* J_distortion;        /* n x 2 x 4       radtan [k1 k2 p1 p2]                   (pixel)            */
  double* J_gyro_intrinsics;   /* n x 6 x 6       i_g [c00 c11 c22 c10 c20 c21]          (inertial)         */
  double* J_acc_intrinsics;    /* n x 6 x 6       i_a                                    (inertial)         */
  double* J_gyro_sensitivity;  /* n x 6 x 9       S_g, column-major                      (inertial)         */
  double* J_acc_offsets;       /* n x 6 x 9       X_a, column-major                      (inertial)         */
} hs_linearization;

/* Exchange hook for the multi-GPU path: called once per linearisation with a device buffer of
 * `count` doubles ([S | g | cost]) that must be summed in place across ranks (RCCL all-reduce, fp64), enqueued on
 * `stream` (the handle's HIP stream). Return 0 on success. NULL = single GPU. */
typedef int (*hs_allreduce_fn)(void* user, void* device_buffer, int64_t count, void* stream);

/* ---- lifetime ---------------------------------------------------------------------------------------------- */
/* Replaces: construction of Optimizer<CERES> in Backend::Backend (backend.cpp:37-46). `stream` is a hipStream_t
 * (NULL = the library creates its own). */
int hs_create(int device, void* stream, hs_problem** out);
int hs_destroy(hs_problem* p);
const char* hs_last_error(const hs_problem* p);
/* Library/ABI version and the gfx arch the device code was built for (e.g. "gfx950"). */
int hs_version(void);
const char* hs_arch(void);

/* ---- tables (host -> HBM) ------------------------------------------------------------------------------------ */
/* Replaces swapState/updateState (optimizer.cpp:110-128, 286-345) + setStateManifold (backend.cpp:52-55):
 * uniform spline of order k (k control points per segment; BasisInterpolator(k-1, true)), control point j at stamp
 * t0 + j*dt. cp = n_cp x 8. cp_constant[j] != 0 freezes control point j (optimizer.cpp:323-328); may be NULL.
 * A table whose stamps are not t0 + j*dt is refused with HS_ERR_KNOTS (its own code: the one refusal a caller may want to survive,
 * see the plugin's optimize()); every other bad argument is HS_ERR_INVALID. */
int hs_set_spline(hs_problem* p, int order, double t0, double dt, int n_cp, const double* cp, const uint8_t* cp_constant,
                  int rotation_constant, int translation_constant);
/* Replaces setSensorManifold for cameras (optimizer.cpp:143-155; constant blocks, camera.hpp:18).
 * T_bs n x 7, intrinsics n x 4 [cx cy fx fy] (settings.yaml:38-40), distortion n x 4 radtan [k1 k2 p1 p2] (:42-45). */
int hs_set_cameras(hs_problem* p, int n, const double* T_bs, const double* intrinsics, const double* distortion);
/* Plain sensors (extrinsics only) used by pose-prior factors (manifold.cpp:30-33). T_bs n x 7. */
int hs_set_sensors(hs_problem* p, int n, const double* T_bs);
/* Replaces addLandmark/updateLandmarks (optimizer.cpp:347-382). xyz n x 3; constant may be NULL. */
int hs_set_landmarks(hs_problem* p, int n, const double* xyz, const uint8_t* constant);
/* Replaces setSensorManifold for the IMU (optimizer.cpp:59-64) + its bias splines (imu.cpp:64-81):
 * T_bs[7], i_g[6], i_a[6] ([c00 c11 c22 c10 c20 c21], settings.yaml:87-89), S_g[9], X_a[9] (column-major, inertial.cpp:48-49);
 * bias splines: uniform R^3 splines of order bias_order, control point j at bias_t0 + j*bias_dt, n_bias x 4 [x y z t]. */
int hs_set_imu(hs_problem* p, const double* T_bs, const double* i_g, const double* i_a, const double* S_g, const double* X_a,
               int bias_order, double bias_t0, double bias_dt, int n_bias, const double* bias_g, const double* bias_a, int bias_constant);
/* Jacobian of the inertial factor. HS_INERTIAL_AS_REFERENCE (default; also HS_REFERENCE_LITERAL=1 in the environment at hs_create)
 * reproduces inertial.cpp:131-198 as written: the linear rows of the state / extrinsic-rotation columns carry I_g where the
 * prediction has I_a (:136,142,148,158), and the S_g / X_a terms of the state, extrinsic and gravity columns are absent (:134-153,
 * 155-162,198). HS_INERTIAL_EXACT (HS_REFERENCE_LITERAL=0) is the derivative of the prediction (:200-203). The two coincide for
 * I_g = I_a, S_g = 0, X_a = 0, i.e. everywhere the reference is exercised (settings.yaml:87-96). */
#define HS_INERTIAL_AS_REFERENCE 0
#define HS_INERTIAL_EXACT 1
int hs_set_inertial_jacobian(hs_problem* p, int mode);
/* Replaces swapEnvironment gravity block + setGravityConstant (optimizer.cpp:84-108, 130-141; rule abstract.cpp:57-61). */
int hs_set_gravity(hs_problem* p, const double* g, int constant);

/* Residual tables = the content of problem_.AddResidualBlock calls (optimizer.cpp:189-274). Any order is accepted;
 * the library keeps its own landmark-major permutation and reports results in table order. */
int hs_set_pixel_residuals(hs_problem* p, int n, const double* stamps, const double* pixels, const int32_t* landmark, const int32_t* camera);
int hs_set_bearing_residuals(hs_problem* p, int n, const double* stamps, const double* bearings, const int32_t* landmark, const int32_t* camera);
int hs_set_prior_residuals(hs_problem* p, int n, const double* stamps, const double* poses, const int32_t* sensor);
int hs_set_inertial_residuals(hs_problem* p, int n, const double* stamps, const double* measurements);

/* ---- delta interface: tables kept incrementally between solves --------------------------------- */
/* The reference never rebuilds its problem: CeresOptimizer::add(...) appends one residual block when AbstractOptimizer::process hands it
 * an observation (optimizer.cpp:189-274 <- abstract.cpp:246-259, 266-292), addLandmark / updateLandmarks add and remove landmark
 * parameter blocks as the window moves (optimizer.cpp:347-382; RemoveParameterBlock takes the landmark's residual blocks along,
 * :365-371), updateState adds / removes state elements (:286-345), and optimize() only calls ceres::Solve (:276-280). The functions
 * below are that interface on the flat tables: rows are appended / retired when the caller learns of them, hs_stage() sorts and
 * uploads what changed — between solves, off optimize()'s clock — and an hs_solve() that finds the tables staged uploads nothing
 * (a control-point table re-sent with new values and the same knots costs one small copy).
 * The hs_set_* functions above remain the whole-table form of the same thing; both may be mixed.
 * A delta call first brings the library's host copies of the variables (control points, landmarks, bias points, gravity) up to the
 * result of the last hs_solve, so the state persists across solves without the caller re-sending it. */
/* Rows appended to the landmark table; *first_index (nullable) receives the index of the first new row (= the landmark id the residual
 * rows refer to). Replaces addLandmark (optimizer.cpp:347-358). */
int hs_append_landmarks(hs_problem* p, int n, const double* xyz, const uint8_t* constant, int32_t* first_index);
/* Rows appended to the residual tables: same columns as hs_set_*_residuals. Replace add(...) (optimizer.cpp:189-274). */
int hs_append_pixel_residuals(hs_problem* p, int n, const double* stamps, const double* pixels, const int32_t* landmark, const int32_t* camera);
int hs_append_bearing_residuals(hs_problem* p, int n, const double* stamps, const double* bearings, const int32_t* landmark, const int32_t* camera);
int hs_append_prior_residuals(hs_problem* p, int n, const double* stamps, const double* poses, const int32_t* sensor);
int hs_append_inertial_residuals(hs_problem* p, int n, const double* stamps, const double* measurements);
/* Removes landmarks `ids` (table indices) together with every visual residual row that refers to them (updateLandmarks,
 * optimizer.cpp:360-382). The remaining landmarks keep their order and move up; remap (nullable, one entry per OLD row) receives
 * the new index of every old row, -1 for the retired ones. */
int hs_retire_landmarks(hs_problem* p, int n, const int32_t* ids, int32_t* remap);
/* Removes the rows of residual table `type` whose stamp is < stamp. (Ceres never removes prior / inertial residual blocks; the
 * plugin's retirement rule for them, include/hyper/optimizers/hip/optimizer.hpp updateLandmarks, is this call.) */
int hs_retire_residuals_before(hs_problem* p, int type, double stamp);
/* Sorts and uploads whatever changed since the tables were last staged (enqueued on the handle's stream, nothing is awaited). hs_solve
 * and the evaluation entry points stage by themselves when needed: calling this earlier only moves the work off their clock. */
int hs_stage(hs_problem* p);

/* ---- structure (bit-exact parity target) ------------------------------------------------------- */
/* Restates ExteroceptiveCost::update (exteroceptive.cpp:25-99) for residual `idx` of `type`:
 * indices[4] = {static_state_idx, static_sensor_idx, dynamic_sensor_idx, static_observation_idx},
 * sizes/offsets/block_ids have *num_blocks entries (caller provides room for 32),
 * block_ids: control-point index for state blocks, sensor id for sensor blocks, bias control-point index for bias
 * blocks, landmark id (or 0 for gravity) for the observation block. */
int hs_residual_layout(hs_problem* p, int type, int idx, int32_t* num_blocks, int32_t* indices, int32_t* sizes, int32_t* offsets,
                       int32_t* block_ids, int32_t* num_parameters, int32_t* num_residuals);

/* ---- evaluation (parity / debugging surface) ------------------------------------------------------------------ */
int hs_num_residuals(hs_problem* p, int type);
int hs_dim_pose(hs_problem* p); /* 6*n_cp (+ 6*n_bias + 2 with an IMU): size of the reduced system */
/* Batched replacement of ExteroceptiveCost::Evaluate + Ceres' local-Jacobian projection for every residual of `type`.
 * robustify != 0 additionally applies Ceres' loss corrector (sqrt(rho') scaling of r and J). */
int hs_linearize(hs_problem* p, int type, int robustify, const hs_linearization* out);
/* Ceres-compatible single-block entry: same contract as
 * `bool ExteroceptiveCost::Evaluate(double const* const* parameters, double* residuals, double** jacobians)`
 * (exteroceptive.hpp:31): parameters in update() block order, jacobians[i] row-major num_residuals x size_i, nullable
 * individually or as a whole. Evaluated on the GPU at the *given* parameter values. */
int hs_cost_function_evaluate(hs_problem* p, int type, int idx, const double* const* parameters, double* residuals, double** jacobians);
/* CostConfiguration::weights (forward.hpp:30-41; exteroceptive.cpp:109-121,129-147): an n_res x n_res matrix W per factor type
 * (row-major; pixel 2, bearing 1, prior 6, inertial 6; NULL clears) with residual = W * distance(..) and J_w = W * J_m * J_e.
 * Every production call site of the reference passes weights = nullptr (optimizer.cpp:191,214,236,255), and so does the solver here:
 * the weights are honoured by the EVALUATION entry points (hs_linearize, hs_cost_function_evaluate: device rows, weighted and
 * loss-corrected per residual block on the way out); hs_solve / hs_cost / hs_reduced_system return HS_ERR_INVALID while a weight
 * matrix is set. */
int hs_set_weights(hs_problem* p, int type, const double* weights);
/* Total cost 0.5*sum rho(|r|^2) at the current point. */
int hs_cost(hs_problem* p, double* cost);
/* Reduced (landmark-eliminated), Jacobi-scaled, LM-damped system of the first iteration at the current point:
 * S (dim x dim, row-major, symmetric) and g (dim). What one LM iteration factors; parity target for the Schur build. */
int hs_reduced_system(hs_problem* p, double radius, double* S, double* g);

/* ---- solve ------------------------------------------------------------------------------------------------------ */
/* Replaces CeresOptimizer::optimize (optimizer.cpp:276-280) with the options of optimizer.cpp:38-54 (trust-region LM,
 * Jacobi scaling, monotonic steps). iterations (nullable) receives max_iterations + 1 records (record 0 = initial point). */
int hs_solve(hs_problem* p, int max_iterations, hs_summary* summary, hs_iteration* iterations);
/* Per-stage device times in the summary: linearize_ms / schur_ms / solve_ms / update_ms. Off by default — the four HIP events per
 * iteration that bracket the stages are barrier packets on the launch stream and cost ~5.7 us each on gfx950 (7 % of a configs[1]
 * iteration). total_ms is always measured. enabled != 0 turns the stage events on for the following hs_solve calls
 * (HS_STAGE_TIMING=1 in the environment at hs_create does the same). Visual-only solves linearise each iteration's
 * CANDIDATE (its records become the next iteration's linearisation when the step is accepted): that launch is booked under
 * linearize_ms, so that linearize_ms covers max_iterations launches of the linearisation kernel as on every other path. */
int hs_set_stage_timing(hs_problem* p, int enabled);
/* Debugging mode of the library (also HS_GUARD=1 in the environment), process wide: every device table allocated while it is on has exactly
 * the size asked for, followed by a known pattern, and hs_solve / hs_cost / hs_reduced_system / hs_linearize check the patterns of all of them
 * before they return (HS_ERR_DEVICE with the table's size if a kernel wrote past the end of one). Slow; meant for test suites. */
int hs_set_guard(int enabled);
int hs_set_allreduce(hs_problem* p, hs_allreduce_fn fn, void* user);
/* RCCL on the data path without a host hook: rank 0 obtains a 128-byte unique id (ncclGetUniqueId), the caller distributes it
 * by any means (torch.distributed in bench.py), every rank then creates its communicator (ncclCommInitRank on the handle's
 * device). From then on both per-iteration exchanges are ncclAllReduce(sum, f64, in place) enqueued on the handle's stream; the
 * hook of hs_set_allreduce, if any, is ignored. librccl.so is loaded on first use (no link-time dependency). */
int hs_rccl_unique_id(char id[128]);
int hs_rccl_init(hs_problem* p, const char id[128], int rank, int world);
/* Destroys the communicator of hs_rccl_init (no-op without one): the exchanges fall back to the hs_set_allreduce hook. Every rank
 * must call it when the collective initialisation did not succeed everywhere, so that all ranks issue the same collectives. */
int hs_rccl_shutdown(hs_problem* p);
/* What the shards exchange, read back from the library (a driver can see that RCCL really runs with N ranks): rccl_ranks = ncclCommCount
 * of the communicator of hs_rccl_init (0: none, the hs_set_allreduce hook or a single shard); doubles_per_linearisation = length of the
 * all-reduce behind every linearisation ([S | g | diag | border blocks | cost | gradient norms], valid after the first hs_solve /
 * hs_reduced_system of the current tables); doubles_per_decision = length of the second all-reduce of an iteration (5). Any pointer may
 * be NULL. */
int hs_exchange_info(hs_problem* p, int32_t* rccl_ranks, int64_t* doubles_per_linearisation, int64_t* doubles_per_decision);
/* Residual-sharded operation: this handle holds shard `rank` of `world` (all observations of a landmark on one rank,
 * control points / sensors replicated). min_band_blocks = max over ranks of hs_band_blocks() so that every rank uses the
 * same band layout for the exchanged reduced system. */
int hs_set_shard(hs_problem* p, int rank, int world, int min_band_blocks);
/* Width of the block band of the reduced system in control-point blocks (max control points touched by one landmark). */
int hs_band_blocks(hs_problem* p);
/* Device-side copy / restore of the current point (control points, landmarks): lets a caller re-run optimize() from the
 * same window state without a host round trip (the reference's equivalent is re-creating the problem). */
int hs_snapshot(hs_problem* p);
int hs_restore(hs_problem* p);

/* ---- read back (the reference mutates the variables in place through raw double*, optimizer.cpp:299-305) ------- */
int hs_get_control_points(hs_problem* p, double* cp);
int hs_get_landmarks(hs_problem* p, double* xyz);
int hs_get_bias(hs_problem* p, double* bias_g, double* bias_a);
int hs_get_gravity(hs_problem* p, double* g);

/* ---- batched trajectory sampling (main.cpp:72-79) -------------------------------------------- */
/* Evaluates the spline at n stamps: pose n x 7 [qx qy qz qw px py pz]; velocity/acceleration (nullable) n x 6
 * [angular(3) body ; linear(3) world]. */
int hs_sample_trajectory(hs_problem* p, int n, const double* stamps, double* pose, double* velocity, double* acceleration);

/* AbstractOptimizer::process(VisualTracks) (internal/hyper/optimizers/abstract.cpp:186-264), the step right before the path:
 * pixel -> bearing conversion in both cameras of a stereo pair (Camera::convertPixelsToBearings, abstract.cpp:222-223; radtan
 * undistortion by 20 fixed-point iterations, unit norm) and triangulation of each pair into the world frame through the
 * current spline value at `stamp` (state evaluate abstract.cpp:197-198, Camera::Triangulate abstract.cpp:252: midpoint of the
 * two rays). Cameras 0 and 1 of hs_set_cameras, control points of hs_set_spline. pixels: n x 2; bearings: n x 3 (sensor
 * frames); positions_w: n x 3. Any output may be NULL. */
int hs_process_tracks(hs_problem* p, double stamp, int n, const double* pixels0, const double* pixels1, double* bearings0, double* bearings1,
                      double* positions_w);

/* ---- manifolds -------------------------------------------------------------------- */
/* The retractions the solve applies, exposed as the batched counterpart of ceres::Manifold::Plus / PlusJacobian as the reference
 * forwards them (include/hyper/optimizers/ceres/manifolds/variables/wrapper.hpp:32-38), and Minus / MinusJacobian (:44-50; never
 * called by Ceres' trust-region minimiser, provided for completeness of the Manifold interface). kind: */
#define HS_MANIFOLD_CONSTANT 0       /* SubsetManifold, all fixed (manifolds/variables/euclidean.hpp:35-36,45-50): tangent 0        */
#define HS_MANIFOLD_EUCLIDEAN 1      /* EuclideanManifold (euclidean.hpp:38), landmarks (optimizer.cpp:356): tangent = ambient     */
#define HS_MANIFOLD_CONTROL_POINT 2  /* Stamped<SE3> [q(4) p(3) t] (stamped.hpp:35-36, se3.cpp:20-23, su2.cpp:21): 8 -> 6, t fixed */
#define HS_MANIFOLD_SE3 3            /* [q(4) p(3)], sensor extrinsics (ceres/manifolds/sensors/sensor.cpp:26-29): 7 -> 6          */
#define HS_MANIFOLD_SPHERE3 4        /* SphereManifold<3>, gravity / bearings (variables/bearing.cpp:15, gravity.hpp:11-17): 3 -> 2 */
#define HS_MANIFOLD_BIAS_POINT 5     /* Stamped<R3> [b(3) t] bias control point (ceres/manifolds/sensors/imu.cpp:64-66): 4 -> 3    */
/* Tangent size of `kind` for an ambient size (only HS_MANIFOLD_CONSTANT / _EUCLIDEAN use `ambient`, 1..9), -1 if invalid. */
int hs_manifold_tangent_size(int kind, int ambient);
/* x: n x ambient, delta: n x tangent, x_plus_delta: n x ambient (Manifold::Plus, wrapper.hpp:32-34). */
int hs_manifold_plus(hs_problem* p, int kind, int ambient, int n, const double* x, const double* delta, double* x_plus_delta);
/* jacobian: n x (ambient x tangent) row-major (Manifold::PlusJacobian, wrapper.hpp:36-38). */
int hs_manifold_plus_jacobian(hs_problem* p, int kind, int ambient, int n, const double* x, double* jacobian);
/* y, x: n x ambient, y_minus_x: n x tangent (Manifold::Minus, wrapper.hpp:44-46): the tangent vector with Plus(x, .) = y. */
int hs_manifold_minus(hs_problem* p, int kind, int ambient, int n, const double* y, const double* x, double* y_minus_x);
/* jacobian: n x (tangent x ambient) row-major, d Minus(y, x) / dy at y = x (Manifold::MinusJacobian, wrapper.hpp:48-50). */
int hs_manifold_minus_jacobian(hs_problem* p, int kind, int ambient, int n, const double* x, double* jacobian);

#ifdef __cplusplus
}
#endif
#endif /* HYPERSLAM_HIP_H_ */
