// Type-name stand-in for <boost/dynamic_bitset.hpp> (tests/stubs/README.md): the members the adapter touches.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
namespace boost
{
template <typename Block = unsigned long, typename Allocator = std::allocator<Block>>
class dynamic_bitset
{
  public:
    using size_type = std::size_t;
    class reference
    {
      public:
        reference& operator=(bool x);
        operator bool() const;
    };
    dynamic_bitset() = default;
    explicit dynamic_bitset(size_type num_bits, unsigned long value = 0);
    void resize(size_type num_bits, bool value = false);
    size_type size() const;
    size_type count() const;
    reference operator[](size_type pos);
    bool operator[](size_type pos) const;
    dynamic_bitset& set(size_type n, bool val = true);
    bool test(size_type n) const;

  private:
    std::vector<Block, Allocator> bits_;
    size_type n_ = 0;
};
} // namespace boost
