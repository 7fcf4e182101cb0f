// compat_std_math.h — TEST INFRASTRUCTURE.
// libstdc++ 11 does not declare the C99 float math names inside namespace std
// (std::sqrtf, std::cosf, ...) although MSVC and libc++ — the toolchains the
// reference targets — do.  The reference headers use them
// (/root/reference/Source/Utility/Random.h:278-281,302), so pull the global
// declarations into std.  Nothing else about the reference is altered.
#pragma once
#include <cmath>
#include <math.h>
namespace std {
using ::sqrtf;
using ::cosf;
using ::sinf;
using ::powf;
using ::tanf;
using ::fabsf;
using ::atan2f;
} // namespace std
