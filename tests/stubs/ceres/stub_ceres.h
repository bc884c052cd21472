// Stand-in (see tests/stubs/README.md): the Ceres names the HyperSLAM headers and tools/reference_dump.cpp mention — declarations only.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>
namespace ceres {
enum Ownership { DO_NOT_TAKE_OWNERSHIP, TAKE_OWNERSHIP };
enum LinearSolverType { DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR };
enum SparseLinearAlgebraLibraryType { SUITE_SPARSE, EIGEN_SPARSE, NO_SPARSE };
enum CallbackReturnType { SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY };
class Manifold {
 public:
  virtual ~Manifold() = default;
  virtual int AmbientSize() const = 0;
  virtual int TangentSize() const = 0;
  virtual bool Plus(const double*, const double*, double*) const = 0;
  virtual bool PlusJacobian(const double*, double*) const = 0;
  virtual bool RightMultiplyByPlusJacobian(const double*, int, const double*, double*) const;
  virtual bool Minus(const double*, const double*, double*) const = 0;
  virtual bool MinusJacobian(const double*, double*) const = 0;
};
class EuclideanManifoldBase : public Manifold {};
template <int>
class EuclideanManifold final : public Manifold {
 public:
  EuclideanManifold() = default;
  explicit EuclideanManifold(int);
  int AmbientSize() const final;
  int TangentSize() const final;
  bool Plus(const double*, const double*, double*) const final;
  bool PlusJacobian(const double*, double*) const final;
  bool Minus(const double*, const double*, double*) const final;
  bool MinusJacobian(const double*, double*) const final;
};
class SubsetManifold final : public Manifold {
 public:
  SubsetManifold(int, const std::vector<int>&);
  int AmbientSize() const final;
  int TangentSize() const final;
  bool Plus(const double*, const double*, double*) const final;
  bool PlusJacobian(const double*, double*) const final;
  bool Minus(const double*, const double*, double*) const final;
  bool MinusJacobian(const double*, double*) const final;
};
class EigenQuaternionManifold final : public Manifold {
 public:
  int AmbientSize() const final;
  int TangentSize() const final;
  bool Plus(const double*, const double*, double*) const final;
  bool PlusJacobian(const double*, double*) const final;
  bool Minus(const double*, const double*, double*) const final;
  bool MinusJacobian(const double*, double*) const final;
};
template <int>
class SphereManifold final : public Manifold {
 public:
  int AmbientSize() const final;
  int TangentSize() const final;
  bool Plus(const double*, const double*, double*) const final;
  bool PlusJacobian(const double*, double*) const final;
  bool Minus(const double*, const double*, double*) const final;
  bool MinusJacobian(const double*, double*) const final;
};
template <typename... TManifolds>
class ProductManifold final : public Manifold {
 public:
  template <typename... TArgs>
  explicit ProductManifold(TArgs&&...);
  int AmbientSize() const final;
  int TangentSize() const final;
  bool Plus(const double*, const double*, double*) const final;
  bool PlusJacobian(const double*, double*) const final;
  bool Minus(const double*, const double*, double*) const final;
  bool MinusJacobian(const double*, double*) const final;
};
class CostFunction {
 public:
  virtual ~CostFunction() = default;
  virtual bool Evaluate(double const* const*, double*, double**) const = 0;
  const std::vector<int32_t>& parameter_block_sizes() const;
  int num_residuals() const;

 protected:
  std::vector<int32_t>* mutable_parameter_block_sizes();
  void set_num_residuals(int);
};
class DynamicCostFunction : public CostFunction {
 public:
  virtual void AddParameterBlock(int);
  virtual void SetNumResiduals(int);
};
class LossFunction {
 public:
  virtual ~LossFunction() = default;
};
class HuberLoss final : public LossFunction {
 public:
  explicit HuberLoss(double);
};
class ScaledLoss final : public LossFunction {
 public:
  ScaledLoss(const LossFunction*, double, Ownership);
};
class Context;
class EvaluationCallback;
class Problem {
 public:
  struct Options {
    Ownership cost_function_ownership, loss_function_ownership, manifold_ownership;
    bool enable_fast_removal, disable_all_safety_checks;
    Context* context;
    EvaluationCallback* evaluation_callback;
  };
  Problem();
  explicit Problem(const Options&);
  void* AddResidualBlock(CostFunction*, LossFunction*, const std::vector<double*>&);
  void AddParameterBlock(double*, int, Manifold* = nullptr);
  void RemoveParameterBlock(const double*);
  bool HasParameterBlock(const double*) const;
  void SetManifold(double*, Manifold*);
  void SetParameterBlockConstant(const double*);
  void SetParameterBlockVariable(double*);
};
struct IterationSummary {
  int iteration;
  bool step_is_valid, step_is_nonmonotonic, step_is_successful;
  double cost, cost_change, gradient_max_norm, gradient_norm, step_norm, relative_decrease, trust_region_radius;
};
class IterationCallback {
 public:
  virtual ~IterationCallback() = default;
  virtual CallbackReturnType operator()(const IterationSummary&) = 0;
};
struct Solver {
  struct Options {
    int max_num_iterations, num_threads;
    LinearSolverType linear_solver_type;
    std::vector<int> residual_blocks_for_subset_preconditioner;
    SparseLinearAlgebraLibraryType sparse_linear_algebra_library_type;
    std::shared_ptr<void> linear_solver_ordering, inner_iteration_ordering;
    std::vector<int> trust_region_minimizer_iterations_to_dump;
    bool update_state_every_iteration;
    std::vector<IterationCallback*> callbacks;
  };
  struct Summary {
    std::string BriefReport() const;
  };
};
void Solve(const Solver::Options&, Problem*, Solver::Summary*);
}  // namespace ceres
