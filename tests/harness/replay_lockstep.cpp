// replay_lockstep.cpp — TEST HARNESS: per-optimize() parity of the HIP library against the oracle on a sliding-window replay.
//
// The replay (replay_stream.hpp) is driven by the ORACLE (this binary links liboracle.so under the hso_ prefix: it is the master
// whose results are written back into the window, so the sequence of windows is the oracle's). At every optimize() the very same
// tables are also handed to the HIP library (libhyperslam_hip.so, resolved with dlopen / dlsym: the shadow) and the two are
// compared call by call from identical inputs:
//     reduced normal equations of the first iteration (hs_reduced_system, radius 1e4)          S, g
//     initial cost, the accept / reject sequence and the cost of every LM iteration            hs_solve(..., 5, ...)
//     the solver's final control points, landmarks, bias control points and gravity
// One JSON line per call; the pytest that runs this binary (tests/test_host_driver.py) applies the tolerances. A window is
// "gauge fixed" when at least k of its control points are frozen (optimizer.cpp:323-328): the solution is then unique and the
// 5-iteration trajectory is compared at 1e-6; the first windows of a replay (nothing frozen yet, stereo only) are rank
// deficient up to the LM damping and are compared in quality only.
//   usage: replay_lockstep <path/to/libhyperslam_hip.so> [seconds=3.6] [imu=0|1] [order=4] [prefix=hs_]
//   (prefix: the shadow's symbol prefix — hs_ the product, hsl_ oracle/liboracle_ld.so, hso_ oracle/liboracle.so = the harness against itself)
// HS_LOCKSTEP_DUMP=<file> also writes the raw end points of every call (shadow and master control points and landmarks, binary:
// int32 call, n_cp_values, n_lm_values, then the four double arrays), so that two runs with different shadows — the HIP library and
// oracle/liboracle_ld.so, the long-double build of the oracle — can be compared with each other (tools/lockstep_three_way.py).
// Test infrastructure only: nothing in the product path links or loads the oracle.
#include <dlfcn.h>

#include <cstring>

#include "../../hyperslam_amd/host/replay_stream.hpp"

using namespace hyper_hip;

namespace {

struct Hip {  // the product library's entry points
  void* lib = nullptr;
  int (*create)(int, void*, hs_problem**) = nullptr;
  int (*destroy)(hs_problem*) = nullptr;
  const char* (*last_error)(const hs_problem*) = nullptr;
  int (*set_spline)(hs_problem*, int, double, double, int, const double*, const uint8_t*, int, int) = nullptr;
  int (*set_cameras)(hs_problem*, int, const double*, const double*, const double*) = nullptr;
  int (*set_sensors)(hs_problem*, int, const double*) = nullptr;
  int (*set_landmarks)(hs_problem*, int, const double*, const uint8_t*) = nullptr;
  int (*set_imu)(hs_problem*, const double*, const double*, const double*, const double*, const double*, int, double, double, int, const double*, const double*,
                 int) = nullptr;
  int (*set_gravity)(hs_problem*, const double*, int) = nullptr;
  int (*set_bearing_residuals)(hs_problem*, int, const double*, const double*, const int32_t*, const int32_t*) = nullptr;
  int (*set_pixel_residuals)(hs_problem*, int, const double*, const double*, const int32_t*, const int32_t*) = nullptr;
  int (*set_prior_residuals)(hs_problem*, int, const double*, const double*, const int32_t*) = nullptr;
  int (*set_inertial_residuals)(hs_problem*, int, const double*, const double*) = nullptr;
  int (*dim_pose)(hs_problem*) = nullptr;
  int (*band_blocks)(hs_problem*) = nullptr;
  int (*reduced_system)(hs_problem*, double, double*, double*) = nullptr;
  int (*solve)(hs_problem*, int, hs_summary*, hs_iteration*) = nullptr;
  int (*get_control_points)(hs_problem*, double*) = nullptr;
  int (*get_landmarks)(hs_problem*, double*) = nullptr;
  int (*get_bias)(hs_problem*, double*, double*) = nullptr;
  int (*get_gravity)(hs_problem*, double*) = nullptr;
};

template <class F>
void resolve(void* lib, const char* name, F* fn) {
  *fn = reinterpret_cast<F>(dlsym(lib, name));
  if (!*fn) throw std::runtime_error(std::string("symbol not found: ") + name);
}

double rel_max(const std::vector<double>& a, const std::vector<double>& b) {
  double num = 0, den = 1e-300;
  for (size_t i = 0; i < a.size(); ++i) num = std::max(num, std::fabs(a[i] - b[i])), den = std::max(den, std::fabs(b[i]));
  return num / den;
}
double rel_max(const double* a, const double* b, size_t n) { return rel_max(std::vector<double>(a, a + n), std::vector<double>(b, b + n)); }

}  // namespace

extern "C" int hso_dim_pose(hs_problem*);
extern "C" int hso_reduced_system(hs_problem*, double, double*, double*);

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: replay_lockstep <libhyperslam_hip.so> [seconds] [imu] [order] [prefix]\n");
    return 2;
  }
  const double seconds = argc > 2 ? std::atof(argv[2]) : 3.6;
  const bool with_imu = argc > 3 && std::atoi(argv[3]) != 0;
  Options opt;
  opt.order = argc > 4 ? std::atoi(argv[4]) : 4;
  Hip H;
  H.lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!H.lib) {
    std::fprintf(stderr, "dlopen failed: %s\n", dlerror());
    return 2;
  }
  // prefix hso_ with liboracle.so as the shadow turns the harness into its own CPU self-test (all differences zero)
  const std::string prefix = argc > 5 ? argv[5] : std::getenv("HS_LOCKSTEP_PREFIX") ? std::getenv("HS_LOCKSTEP_PREFIX") : "hs_";
#define HS_RESOLVE(name) resolve(H.lib, (prefix + #name).c_str(), &H.name)
  HS_RESOLVE(create), HS_RESOLVE(destroy), HS_RESOLVE(last_error), HS_RESOLVE(set_spline), HS_RESOLVE(set_cameras), HS_RESOLVE(set_sensors);
  HS_RESOLVE(set_landmarks), HS_RESOLVE(set_imu), HS_RESOLVE(set_gravity), HS_RESOLVE(set_bearing_residuals), HS_RESOLVE(set_pixel_residuals);
  HS_RESOLVE(set_prior_residuals), HS_RESOLVE(set_inertial_residuals), HS_RESOLVE(dim_pose), HS_RESOLVE(band_blocks), HS_RESOLVE(reduced_system);
  HS_RESOLVE(solve), HS_RESOLVE(get_control_points), HS_RESOLVE(get_landmarks), HS_RESOLVE(get_bias), HS_RESOLVE(get_gravity);
#undef HS_RESOLVE
  hs_problem* shadow = nullptr;
  if (H.create(0, nullptr, &shadow) != HS_OK) {
    std::fprintf(stderr, "hs_create failed (no usable GPU?)\n");
    return 2;
  }
  auto check = [&](int rc, const char* what) {
    if (rc != HS_OK) throw std::runtime_error(std::string("hip ") + what + " failed: " + H.last_error(shadow));
  };

  const std::vector<Camera> cams = euroc_cameras();
  IMU imu;
  Optimizer master(opt, cams, with_imu ? &imu : nullptr);  // the oracle (HS_ABI_PREFIX = hso_)

  // results of the shadow for the call in flight
  struct {
    double S_rel = 0, g_rel = 0;
    int dim = 0, bw = 0;
    hs_summary summary{};
    std::vector<hs_iteration> iterations;
    std::vector<double> cp, lm, bg, ba;
    double gravity[3] = {0, 0, 0};
  } sh;
  int worst_fixed_call = -1;
  std::FILE* dump = std::getenv("HS_LOCKSTEP_DUMP") ? std::fopen(std::getenv("HS_LOCKSTEP_DUMP"), "wb") : nullptr;
  double worst_fixed = 0;
  int n_fixed = 0, n_free = 0;

  master.before_solve = [&](const WindowTables& t, int) {
    t.uploadWith(shadow, H.set_spline, H.set_cameras, H.set_sensors, H.set_landmarks, H.set_imu, H.set_gravity, H.set_bearing_residuals, H.set_pixel_residuals,
                 H.set_prior_residuals, H.set_inertial_residuals, check);
    const int dim = hso_dim_pose(master.handle());
    if (H.dim_pose(shadow) != dim) throw std::runtime_error("dim_pose differs");
    std::vector<double> S0(size_t(dim) * dim), g0(dim), S1(size_t(dim) * dim), g1(dim);
    if (hso_reduced_system(master.handle(), 1e4, S0.data(), g0.data()) != HS_OK) throw std::runtime_error("oracle reduced_system failed");
    check(H.reduced_system(shadow, 1e4, S1.data(), g1.data()), "reduced_system");
    sh.S_rel = rel_max(S1, S0), sh.g_rel = rel_max(g1, g0), sh.dim = dim, sh.bw = H.band_blocks(shadow);
    sh.iterations.assign(size_t(opt.max_num_iterations) + 1, hs_iteration{});
    check(H.solve(shadow, opt.max_num_iterations, &sh.summary, sh.iterations.data()), "solve");
    sh.cp.resize(t.cp.size()), sh.lm.resize(t.landmarks.size()), sh.bg.resize(t.bias_g.size()), sh.ba.resize(t.bias_a.size());
    check(H.get_control_points(shadow, sh.cp.data()), "get_control_points");
    if (!sh.lm.empty()) check(H.get_landmarks(shadow, sh.lm.data()), "get_landmarks");
    if (t.has_imu) check(H.get_bias(shadow, sh.bg.data(), sh.ba.data()), "get_bias"), check(H.get_gravity(shadow, sh.gravity), "get_gravity");
  };
  master.after_solve = [&](const WindowTables& t, const WindowTables& r, const hs_summary& s, const std::vector<hs_iteration>& it, int call) {
    const int k = t.order, frozen = t.numFrozen();
    const bool gauge_fixed = frozen >= k;
    const bool same_shape = s.num_iterations == sh.summary.num_iterations && s.num_successful_steps == sh.summary.num_successful_steps &&
                            s.termination == sh.summary.termination;
    bool same_decisions = same_shape;
    double cost_traj = 0;
    const int n_it = std::min(s.num_iterations, sh.summary.num_iterations);
    for (int i = 0; i <= n_it; ++i) {
      same_decisions = same_decisions && it[i].step_is_successful == sh.iterations[i].step_is_successful;
      cost_traj = std::max(cost_traj, std::fabs(it[i].cost - sh.iterations[i].cost) / (std::fabs(it[i].cost) + 1e-2 * s.initial_cost));
    }
    const double cost0 = std::fabs(s.initial_cost - sh.summary.initial_cost) / std::max(1e-300, s.initial_cost);
    const double costN = std::fabs(s.final_cost - sh.summary.final_cost) / std::max(1e-300, s.final_cost);
    const double cp_rel = rel_max(sh.cp, r.cp), lm_rel = sh.lm.empty() ? 0.0 : rel_max(sh.lm, r.landmarks);
    const double bias_rel = t.has_imu ? std::max(rel_max(sh.bg, r.bias_g), rel_max(sh.ba, r.bias_a)) : 0.0;
    const double grav_rel = t.has_imu ? rel_max(sh.gravity, r.gravity, 3) : 0.0;
    const double traj = std::max(std::max(cp_rel, lm_rel), std::max(bias_rel, grav_rel));
    (gauge_fixed ? n_fixed : n_free)++;
    if (dump) {
      const int32_t head[3] = {call, int32_t(sh.cp.size()), int32_t(sh.lm.size())};
      std::fwrite(head, sizeof(head), 1, dump);
      std::fwrite(sh.cp.data(), 8, sh.cp.size(), dump), std::fwrite(sh.lm.data(), 8, sh.lm.size(), dump);
      std::fwrite(r.cp.data(), 8, sh.cp.size(), dump), std::fwrite(r.landmarks.data(), 8, sh.lm.size(), dump);
    }
    if (gauge_fixed && traj > worst_fixed) worst_fixed = traj, worst_fixed_call = call;
    std::printf("{\"call\": %d, \"control_points\": %d, \"frozen\": %d, \"gauge_fixed\": %s, \"landmarks\": %zu, \"blocks\": %d, \"dim\": %d, \"band_blocks\": %d, "
                "\"window\": [%.2f, %.2f], \"gravity_constant\": %d, \"S_rel\": %.3e, \"g_rel\": %.3e, \"cost0_rel\": %.3e, \"iterations\": [%d, %d], "
                "\"successful\": [%d, %d], \"same_decisions\": %s, \"cost_traj_rel\": %.3e, \"final_cost\": [%.12g, %.12g], \"final_cost_rel\": %.3e, "
                "\"cp_rel\": %.3e, \"lm_rel\": %.3e, \"bias_rel\": %.3e, \"gravity_rel\": %.3e}\n",
                call, t.numControlPoints(), frozen, gauge_fixed ? "true" : "false", t.landmarks.size() / 3, t.numResidualBlocks(), sh.dim, sh.bw,
                master.window().lower, master.window().upper, t.gravity_constant, sh.S_rel, sh.g_rel, cost0, s.num_iterations, sh.summary.num_iterations,
                s.num_successful_steps, sh.summary.num_successful_steps, same_decisions ? "true" : "false", cost_traj, s.final_cost, sh.summary.final_cost, costN,
                cp_rel, lm_rel, bias_rel, grav_rel);
  };
  try {
    feed_stream(master, cams, seconds, with_imu, [] {});
  } catch (const std::exception& e) {
    std::fprintf(stderr, "replay_lockstep: %s\n", e.what());
    return 1;
  }
  std::printf("{\"summary\": true, \"replay_seconds\": %.2f, \"imu\": %d, \"order\": %d, \"optimizations\": %d, \"gauge_fixed_calls\": %d, \"gauge_free_calls\": %d, "
              "\"worst_gauge_fixed_trajectory_rel\": %.3e, \"worst_gauge_fixed_call\": %d, \"window\": [%.2f, %.2f]}\n",
              seconds, int(with_imu), opt.order, master.numOptimizations(), n_fixed, n_free, worst_fixed, worst_fixed_call, master.window().lower,
              master.window().upper);
  if (dump) std::fclose(dump);
  H.destroy(shadow);
  return 0;
}
