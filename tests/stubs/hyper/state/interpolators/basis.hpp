// Stand-in for the EXTERNAL header of this name (tests/stubs/README.md).
#pragma once
#include "hyper/stub_external.hpp"
