// TEST INFRASTRUCTURE ONLY (oracle/): stand-in for the cmake-generated system.hpp
// (template: reference src/config/system.h.in). The SIMD headers include it but use nothing from it.
#ifndef ORACLE_SHIM_SYSTEM_HPP
#define ORACLE_SHIM_SYSTEM_HPP
#define SYSTEM_PROCESSOR "x86_64"
#define SYSTEM_NAME "Linux"
#define SYSTEM_VERSION "oracle-shim"
#define COMPILER_NAME "GNU"
#define COMPILER_VERSION "oracle-shim"
#define BOOSTLIB_VERSION "none"
#define BUILD_TYPE "Release"
#endif
