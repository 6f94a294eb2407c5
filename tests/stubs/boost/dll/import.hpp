// Type-name stand-in for <boost/dll/import.hpp> (tests/stubs/README.md): the one call plugin_loader.hpp:21-23 makes.
#pragma once
#include <functional>
#include <string>
namespace boost
{
namespace dll
{
namespace load_mode
{
enum type
{
    default_mode = 0,
    append_decorations = 0x00800000
};
}
template <class T>
std::function<T> import_alias(const std::string& lib, const std::string& name, load_mode::type mode = load_mode::default_mode);
} // namespace dll
} // namespace boost
