// Type-name stand-in for <boost/core/span.hpp> (tests/stubs/README.md)
#pragma once
#include <cstddef>
namespace boost
{
template <class T, std::size_t E = static_cast<std::size_t>(-1)>
class span;
}
