// Type-name stand-in for <boost/function.hpp> (tests/stubs/README.md)
#pragma once
#include <functional>
namespace boost
{
template <typename Sig>
using function = std::function<Sig>;
}
