// Stand-in (see tests/stubs/README.md).
#pragma once
#include <cstddef>
#include <string>
namespace YAML {
class Node {
 public:
  class const_iterator {
   public:
    auto operator*() const -> Node;
    auto operator++() -> const_iterator&;
    auto operator!=(const const_iterator&) const -> bool;
  };
  auto IsNull() const -> bool { return true; }
  explicit operator bool() const;
  template <typename TKey>
  auto operator[](const TKey&) const -> const Node;
  template <typename TKey>
  auto operator[](const TKey&) -> Node;
  template <typename TValue>
  auto operator=(const TValue&) -> Node&;
  template <typename TValue>
  auto as() const -> TValue;
  auto size() const -> std::size_t;
  auto begin() const -> const_iterator;
  auto end() const -> const_iterator;
};
auto LoadFile(const std::string&) -> Node;
auto Clone(const Node&) -> Node;
}  // namespace YAML
