// Stand-in for the Ceres header of this name (tests/stubs/README.md).
#pragma once
#include "ceres/stub_ceres.h"
