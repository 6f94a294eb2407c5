// Stand-in for <boost/dll/alias.hpp> (tests/stubs/README.md): BOOST_DLL_ALIAS as Boost.DLL 1.84 defines it on ELF platforms.
#pragma once
#include <cstdint>
#define BOOST_DLL_ALIAS(FunctionOrVar, AliasName)                                                                     \
    extern "C" __attribute__((visibility("default"))) const void* AliasName;                                          \
    __attribute__((section("boostdll"))) const void* AliasName =                                                      \
        reinterpret_cast<const void*>(reinterpret_cast<std::intptr_t>(&FunctionOrVar));
