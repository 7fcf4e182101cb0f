// test aid: stands in for the macOS SDK header HeatrayRenderer.h:18 includes on non-Windows platforms (glew.h from 3rdParty has
// already declared the GL API by then).  The toolchains the reference targets (MSVC, Apple clang + libc++ with the SDK's headers)
// make two things visible that g++ 11 / libstdc++ does not, and the application's own headers rely on both:
//   * the C99 float math names inside namespace std — std::sqrtf, std::cosf, ... (Utility/Random.h:278-302, Utility/AABB.h:49,
//     HeatrayRenderer.h:238): oracle/ref/compat_std_math.h, the shim BASELINE.md §3 already names;
//   * ::memcpy without <cstring> (Scene/PlaneMeshProvider.h:55-94, SphereMeshProvider.h:62).
// This stand-in for a PLATFORM header supplies them, as the platform would; nothing of the application or of the overlay is touched,
// and the compile uses no -include.
#pragma once

#include <cstring>

#include "../../../../oracle/ref/compat_std_math.h"
