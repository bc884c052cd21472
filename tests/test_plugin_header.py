"""Compile check of the reference-side plugin include/hyper/optimizers/hip/optimizer.hpp.

Build container only: the header is type-checked against the reference's own in-tree headers where they lie under /root/reference
(`AbstractOptimizer` /root/reference/include/hyper/optimizers/abstract.hpp:53-139, `Environment`, the observation and landmark
classes) with the hand-written declaration-only stand-ins of tests/stubs/ for the EXTERNAL ones. Skipped where /root/reference is
absent (the GPU box). No reference text lives in this repository; nothing is linked or run.
"""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/include"
PLUGIN = "hyper/optimizers/hip/optimizer.hpp"

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is only present in the build container")


def _syntax_check(source: str):
    with tempfile.NamedTemporaryFile("w", suffix=".cpp", delete=False) as f:
        f.write(source)
    try:
        return subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", REFERENCE,
                               "-I", os.path.join(ROOT, "tests", "stubs"), f.name], capture_output=True, text=True)
    finally:
        os.unlink(f.name)


def test_plugin_header_compiles_against_the_reference_interface():
    """Every override matches a virtual of AbstractOptimizer, the class is concrete (make_hip_optimizer instantiates it), every call
    into the reference's Environment / observation / landmark / sensor / state classes type-checks, and the header is self-contained."""
    out = _syntax_check(f'''#include "{PLUGIN}"
static_assert(!std::is_abstract_v<hyper::HipOptimizer>);
static_assert(std::is_base_of_v<hyper::AbstractOptimizer, hyper::HipOptimizer>);
auto make(const YAML::Node& node, const std::vector<hyper::Sensor*>& sensors) {{ return hyper::make_hip_optimizer(node, sensors); }}
''')
    assert out.returncode == 0, out.stderr


@pytest.mark.parametrize("virtual", ["updateSensor", "updateState", "updateLandmarks", "addLandmark", "swapState", "setGravityConstant", "hasSensor"])
def test_the_check_is_not_vacuous(virtual):
    """The same translation unit with one override renamed must be rejected (`final` on a function that overrides nothing, and the class
    stays abstract): the stand-ins do not make the compile check pass by construction."""
    out = _syntax_check(f'''#include "hyper/optimizers/abstract.hpp"
#define {virtual} {virtual}Renamed
#include "{PLUGIN}"
''')
    assert out.returncode != 0
    assert "final" in out.stderr or "abstract" in out.stderr or "override" in out.stderr, out.stderr


def test_bias_splines_are_created_by_update_sensor():
    """abstract.cpp:278-289 calls updateSensor(imu, window_) when a bias spline is empty or does not contain the stamp and DCHECKs the
    containment afterwards; upstream's body is CHECK(false) (ceres/optimizer.cpp:384-386). The plugin's body must insert elements into
    both splines and must not be a no-op."""
    text = open(os.path.join(ROOT, "include", PLUGIN)).read()
    body = text[text.index("auto updateSensor(IMU& imu, const Range& range) -> void final"):]
    body = body[:body.index("\n  }\n") + 5]
    assert "imu.gyroscopeBias()" in body and "imu.accelerometerBias()" in body and "extendBias" in body
    extend = text[text.index("auto extendBias(AbstractState& bias, const Range& range) -> void"):]
    assert "elements.insert(std::move(element))" in extend and "elements.empty()" in extend


def test_control_point_table_stays_contiguous():
    """The library's basis is uniform (hs_set_spline refuses a table with a hole): updateState() may only drop a prefix / suffix of the
    parameter blocks in stamp order — never `erase_if` single elements like Ceres does (ceres/optimizer.cpp:330-341) — and optimize()
    builds the table from the contiguous run of state elements, handing an element that is not a parameter block over as a constant row."""
    text = open(os.path.join(ROOT, "include", PLUGIN)).read()
    body = text[text.index("auto updateState(const Range& range) -> void final"):]
    body = body[:body.index("\n  }\n") + 5]
    assert "erase_if(variables_" not in body and "first_kept" in body and "last_kept" in body
    opt = text[text.index("auto optimize() -> void final"):text.index("// ---- sensors:")]
    assert "elements.lower_bound(oldest_stamp)" in opt and "!variables_.contains(itr->get())" in opt
    assert "admitted(m.stamp())" in text and text.count("admitted(m.stamp())") == 4  # bearing, pixel, prior, inertial tables


def test_reference_dump_compiles_against_the_reference_interface():
    """tools/reference_dump.cpp — the reference side of the conformance kit: the reference's own evaluators (ExteroceptiveCost::update /
    Evaluate over Evaluator<Observation, SE3>), manifolds (ceres/manifolds/**) and ceres::Solve on the inputs of tests/golden/*.json, written
    out in the same schema (HS_REFERENCE_VECTORS=<dir> then points the golden tests at them, tests/util.py). It can only run where HyperSLAM
    builds; here every call into the reference's API is type-checked exactly like the plugin."""
    tool = os.path.join(ROOT, "tools", "reference_dump.cpp")
    out = subprocess.run(["g++", "-std=c++20", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", REFERENCE,
                          "-I", os.path.join(ROOT, "tests", "stubs"), tool], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    text = open(tool).read()
    for needed in ("cost.update()", "cost.Evaluate(", "RightMultiplyByPlusJacobian", "ceres::Solve(", "MinusJacobian", "factors.json", "inertial_literal.json",
                   "manifolds.json", "solve_visual.json"):
        assert needed in text, needed
