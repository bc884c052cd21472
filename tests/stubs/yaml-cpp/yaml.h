// Stand-in (see tests/stubs/README.md).
#pragma once
namespace YAML {
class Node {
 public:
  auto IsNull() const -> bool { return true; }
};
}  // namespace YAML
