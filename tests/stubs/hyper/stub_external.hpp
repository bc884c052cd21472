// Declaration-only stand-ins for the un-vendored HyperVariables / HyperState / HyperSensors / HyperMessages headers, written by hand
// from the call sites in the reference (see tests/stubs/README.md). Every EXTERNAL header path the in-tree headers include
// resolves to a one-line file that includes this one. Nothing here is linked or run.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <set>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include <Eigen/Core>
#include <glog/logging.h>
#include <yaml-cpp/yaml.h>

namespace hyper {

// ---- hyper/definitions.hpp ----
using Scalar = double;
using Stamp = Scalar;
using Identifier = std::size_t;
using Index = std::ptrdiff_t;
template <typename T>
using Pointers = std::vector<T*>;
template <typename>
struct Traits;

// ---- hyper/range.hpp (abstract.cpp:44-46,79,139-142; optimizer.cpp:288-294,365) ----
enum class BoundaryPolicy { INCLUSIVE, LOWER_INCLUSIVE_ONLY, UPPER_INCLUSIVE_ONLY, EXCLUSIVE };
template <typename T, BoundaryPolicy>
struct Range {
  T lower, upper;
  auto lowerBound() const -> const T& { return lower; }
  auto upperBound() const -> const T& { return upper; }
  auto size() const -> T { return upper - lower; }
  auto contains(const T&) const -> bool;
  auto isSmaller(const T&) const -> bool;
  auto isGreater(const T&) const -> bool;
  template <BoundaryPolicy TOther>
  auto intersects(const Range<T, TOther>&) const -> bool;
  auto sample() const -> T;
};

// ---- hyper/variables/** (optimizer.cpp:113,150,299; exteroceptive.cpp:62-94; tests/.../inertial.cpp:72-75) ----
template <typename TScalar>
class AbstractVariable {
 public:
  virtual ~AbstractVariable() = default;
  virtual auto asVector() -> Eigen::Map<Eigen::Matrix<TScalar, Eigen::Dynamic, 1>>;
  virtual auto asVector() const -> Eigen::Map<const Eigen::Matrix<TScalar, Eigen::Dynamic, 1>>;
};
template <typename TScalar, int TSize>
class Cartesian : public Eigen::Matrix<TScalar, TSize, 1>, public AbstractVariable<TScalar> {
 public:
  using Eigen::Matrix<TScalar, TSize, 1>::Matrix;
};
template <typename TScalar>
using Position = Cartesian<TScalar, 3>;
template <typename TScalar>
using Pixel = Cartesian<TScalar, 2>;
template <typename TScalar>
using Bearing = Cartesian<TScalar, 3>;
template <typename TScalar>
class Gravity : public Cartesian<TScalar, 3> {};
template <typename TScalar>
class SU2 : public Cartesian<TScalar, 4> {};
template <typename TScalar>
class SE3 : public Cartesian<TScalar, 7> {};
template <typename TManifold>
class Tangent : public Cartesian<Scalar, 6> {};
template <typename TScalar>
class AbstractStamped : public AbstractVariable<TScalar> {
 public:
  auto stamp() const -> const Stamp&;
  auto stamp() -> Stamp&;
};
template <typename TVariable>
class Stamped final : public AbstractStamped<Scalar> {
 public:
  auto variable() const -> const TVariable&;
  auto variable() -> TVariable&;
};
template <typename TVariable>
struct Traits<Stamped<TVariable>> {
  using Stamp = Cartesian<hyper::Stamp, 1>;  // (a variable type: ceres/manifolds/variables/stamped.hpp:32-33 builds Manifold<Stamp, CERES> from it)
  static constexpr auto kNumParameters = 0;
};
template <typename TScalar>
class CompositeVariable {
 public:
  explicit CompositeVariable(std::size_t);
  auto variable(std::size_t) const -> const AbstractVariable<TScalar>&;
  auto setVariable(std::size_t, std::unique_ptr<AbstractVariable<TScalar>>&&) -> void;
};
template <typename TScalar>
class AbstractMetric {
 public:
  virtual ~AbstractMetric() = default;
};
template <typename TVariable>
class AngularMetric final : public AbstractMetric<Scalar> {};    // optimizer.cpp:192
template <typename TVariable>
class CartesianMetric final : public AbstractMetric<Scalar> {};  // optimizer.cpp:215,256
template <typename TVariable>
class ManifoldMetric final : public AbstractMetric<Scalar> {};   // optimizer.cpp:237
template <typename TScalar>
using DynamicJacobian = Eigen::Matrix<TScalar, Eigen::Dynamic, Eigen::Dynamic>;
template <typename TScalar>
using DynamicVector = Eigen::Matrix<TScalar, Eigen::Dynamic, 1>;
template <typename TOutput, typename TInput = TOutput>
using Jacobian = Eigen::Matrix<Scalar, Eigen::Dynamic, Eigen::Dynamic>;

// ---- hyper/state/** (abstract.cpp:80-96,127-136; optimizer.cpp:286-294; inertial.cpp:34; tests/.../inertial.cpp:55-75) ----
struct StateQuery {
  Stamp stamp;
  int derivative{0};
};
struct InterpolatorLayout {
  struct Block {
    int idx, size;
  };
  struct Padding {
    int left, right;
  };
  Block outer, inner;
  auto outerPadding() const -> Padding;
};
class AbstractInterpolator {
 public:
  virtual ~AbstractInterpolator() = default;
  virtual auto layout() const -> InterpolatorLayout;
};
class AbstractPolicy {
 public:
  virtual ~AbstractPolicy() = default;
};
class BasisInterpolator final : public AbstractInterpolator {  // tests/internal/tests/optimizers/evaluators/bearing.cpp:52: (degree, uniform)
 public:
  explicit BasisInterpolator(int degree = 3, bool uniform = true);
};
template <typename TVariable>
class ManifoldPolicy final : public AbstractPolicy {};  // tests/internal/tests/optimizers/evaluators/bearing.cpp:53
class AbstractState {
 public:
  struct ElementCompare {
    using is_transparent = std::true_type;
    auto operator()(const std::unique_ptr<AbstractStamped<Scalar>>&, const std::unique_ptr<AbstractStamped<Scalar>>&) const -> bool;
    auto operator()(const std::unique_ptr<AbstractStamped<Scalar>>&, const Stamp&) const -> bool;
    auto operator()(const Stamp&, const std::unique_ptr<AbstractStamped<Scalar>>&) const -> bool;
  };
  using Elements = std::set<std::unique_ptr<AbstractStamped<Scalar>>, ElementCompare>;
  using Range = hyper::Range<Stamp, BoundaryPolicy::LOWER_INCLUSIVE_ONLY>;
  AbstractState();  // tests/include/tests/state/abstract.hpp:35
  AbstractState(std::unique_ptr<AbstractInterpolator>&&, std::unique_ptr<AbstractPolicy>&&);
  auto policy() -> std::unique_ptr<AbstractPolicy>&;  // tests/internal/tests/optimizers/evaluators/bearing.cpp:53
  auto elements() const -> const Elements&;
  auto elements() -> Elements&;
  auto range() const -> Range;
  auto interpolator() const -> const std::unique_ptr<AbstractInterpolator>&;
  auto interpolator() -> std::unique_ptr<AbstractInterpolator>&;
};

// ---- hyper/sensors/** (optimizer.cpp:56-71,143-155; abstract.cpp:190-223,275-285) ----
class Sensor {
 public:
  virtual ~Sensor() = default;
  auto parameters() const -> Pointers<AbstractVariable<Scalar>>;
  auto transformation() const -> const SE3<Scalar>&;
  template <typename TSensor>
  auto as() const -> const TSensor&;
  template <typename TSensor>
  auto as() -> TSensor&;
};
template <typename TScalar>
class AbstractDistortion : public AbstractVariable<TScalar> {};
template <typename TScalar, int TOrder>
class RadialTangentialDistortion final : public AbstractDistortion<TScalar> {  // tests/include/tests/sensors/camera.hpp:31
 public:
  RadialTangentialDistortion(TScalar, TScalar, TScalar, TScalar);
  auto perturb(TScalar) -> void;
};
template <typename TScalar>
class Intrinsics final : public Cartesian<TScalar, 4> {  // tests/include/tests/sensors/camera.hpp:27
 public:
  Intrinsics(TScalar, TScalar, TScalar, TScalar);
};
class Camera final : public Sensor {
 public:
  struct SensorSize {
    int width, height;
  };
  auto sensorSize() -> SensorSize&;                                                            // tests/include/tests/sensors/camera.hpp:25
  auto intrinsics() -> Intrinsics<Scalar>&;                                                    // :27
  auto setDistortion(std::unique_ptr<AbstractDistortion<Scalar>>&&) -> void;                   // :33
};
class IMU final : public Sensor {
 public:
  auto gyroscopeBias() const -> const AbstractState&;
  auto gyroscopeBias() -> AbstractState&;
  auto accelerometerBias() const -> const AbstractState&;
  auto accelerometerBias() -> AbstractState&;
};
template <>
struct Traits<Sensor> {
  static constexpr auto kTransformationOffset = 0;
  static constexpr auto kNumParameters = kTransformationOffset + 1;
};
template <>
struct Traits<Camera> : Traits<Sensor> {
  static constexpr auto kIntrinsicsOffset = Traits<Sensor>::kNumParameters;
  static constexpr auto kDistortionOffset = kIntrinsicsOffset + 1;
  static constexpr auto kNumParameters = kDistortionOffset + 1;
};
template <>
struct Traits<IMU> : Traits<Sensor> {
  using GyroscopeBias = Stamped<Cartesian<Scalar, 3>>;
  using AccelerometerBias = Stamped<Cartesian<Scalar, 3>>;
  static constexpr auto kGyroscopeIntrinsicsOffset = Traits<Sensor>::kNumParameters;
  static constexpr auto kAccelerometerIntrinsicsOffset = kGyroscopeIntrinsicsOffset + 1;
  static constexpr auto kGyroscopeSensitivityOffset = kAccelerometerIntrinsicsOffset + 1;
  static constexpr auto kAccelerometerAxesOffsetsOffset = kGyroscopeSensitivityOffset + 1;
  static constexpr auto kNumParameters = kAccelerometerAxesOffsetsOffset + 1;
};

// ---- hyper/messages/** (abstract.cpp:150-292; observations/*.hpp) ----
class AbstractMessage {
 public:
  virtual ~AbstractMessage() = default;
  auto stamp() const -> const Stamp&;
  auto stamp() -> Stamp&;
  virtual auto sensor() const -> const Sensor&;
};
class AbstractMeasurement : public AbstractMessage {
 public:
  virtual auto variable() const -> const AbstractVariable<Scalar>&;
};
template <typename TVariable, typename TSensor = Sensor>
class VariableMeasurement : public AbstractMeasurement {
 public:
  VariableMeasurement(const Stamp&, const TSensor&, const TVariable&);
  auto sensor() const -> const TSensor& final;
  auto variable() const -> const TVariable& final;
};
using PixelMeasurement = VariableMeasurement<Pixel<Scalar>, Camera>;
using BearingMeasurement = VariableMeasurement<Bearing<Scalar>, Camera>;
template <typename TManifold>
class ManifoldMeasurement final : public VariableMeasurement<TManifold> {
 public:
  using VariableMeasurement<TManifold>::VariableMeasurement;
};
template <typename TManifold>
class InertialMeasurement final : public VariableMeasurement<Tangent<TManifold>, IMU> {
 public:
  using VariableMeasurement<Tangent<TManifold>, IMU>::VariableMeasurement;
};
class VisualTracks;

// ---- hyper/yaml/yaml.hpp (abstract.cpp:165-166; backend.cpp:52-55) ----
namespace yaml {
template <typename TValue>
auto ReadAs(const YAML::Node&, const std::string&) -> TValue;
}  // namespace yaml

}  // namespace hyper
