// plugin_loader.hpp — mirror of the reference's plugin loader (plugin_loader.hpp:13-25).
//
// The reference calls boost::dll::import_alias<std::unique_ptr<T>()>(path, name, append_decorations):
// `name` is an exported DATA symbol (BOOST_DLL_ALIAS puts a `const void*` holding the factory's address
// into section "boostdll", test/dummy.cpp:11); import_alias dlsym()s it, dereferences once, and the
// returned callable keeps the library loaded.  Boost is not in this image, so the same contract is
// implemented on dlopen/dlsym; a library built with BOOST_DLL_ALIAS and one built with
// MSLAM_DLL_ALIAS (below) are interchangeable for either loader.
#pragma once
#include <dlfcn.h>

#include <filesystem>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>

namespace mslam
{
namespace fs = std::filesystem;

template <typename BlockType>
using BlockFactoryCreator = std::unique_ptr<BlockType>();

template <typename BlockType>
using BlockFactoryCreatorBoostFunction = std::function<BlockFactoryCreator<BlockType>>;

// library side: what BOOST_DLL_ALIAS(FunctionOrVar, AliasName) expands to on ELF platforms
#define MSLAM_DLL_ALIAS(FunctionOrVar, AliasName)                                                                     \
    extern "C" __attribute__((visibility("default"))) const void* AliasName;                                          \
    __attribute__((section("boostdll"))) const void* AliasName =                                                      \
        reinterpret_cast<const void*>(reinterpret_cast<std::intptr_t>(&FunctionOrVar));

template <typename BlockType>
BlockFactoryCreatorBoostFunction<BlockType> loadFactoryMethod(fs::path path_to_shared_library,
                                                              const std::string& factoryFunctionName)
{
    // load_mode::append_decorations: try the decorated name ("lib" prefix, ".so" suffix) first
    fs::path decorated = path_to_shared_library;
    if(decorated.extension() != ".so")
    {
        std::string fn = decorated.filename().string();
        if(fn.rfind("lib", 0) != 0)
            fn = "lib" + fn;
        decorated = decorated.parent_path() / (fn + ".so");
    }
    void* handle = dlopen(decorated.c_str(), RTLD_NOW | RTLD_LOCAL);
    if(!handle)
        handle = dlopen(path_to_shared_library.c_str(), RTLD_NOW | RTLD_LOCAL);
    if(!handle)
        throw std::runtime_error(std::string("loadFactoryMethod: cannot load library: ") + dlerror());
    std::shared_ptr<void> keep(handle, [](void* h) { dlclose(h); });
    void* alias = dlsym(handle, factoryFunctionName.c_str());
    if(!alias)
        throw std::runtime_error("loadFactoryMethod: symbol not found: " + factoryFunctionName);
    auto fn = reinterpret_cast<BlockFactoryCreator<BlockType>*>(*static_cast<void**>(alias));
    return [keep, fn]() { return fn(); };
}

} // namespace mslam
