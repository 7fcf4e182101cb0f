//
//  RLTypes.h
//  heatray_amd host layer
//
//  The viewer and the texture loader fill openrl::Texture descriptors with RL_* enum values
//  (/root/reference/Source/Utility/TextureLoader.cpp:37-153).  In the Heatray tree those come from
//  3rdParty/OpenRL/rl.h, which stays on the include path (header only, nothing is linked).
//  Standalone builds of this layer have no OpenRL checkout, so the few public API constants the
//  descriptors use are declared here with the values of the OpenRL 1.4 / OpenGL ES headers.
//

#pragma once

#if __has_include(<OpenRL/rl.h>)
#include <OpenRL/rl.h>
#else
#include <stddef.h>
typedef int RLint;
typedef RLint RLenum;
typedef struct _RLtexture* RLtexture;
#define RL_NULL_TEXTURE ((RLtexture)0)
#define RL_TRIANGLES 0x0004
#define RL_TRIANGLE_STRIP 0x0005
#define RL_UNSIGNED_BYTE 0x1401
#define RL_FLOAT 0x1406
#define RL_RGB 0x1907
#define RL_RGBA 0x1908
#define RL_LUMINANCE 0x1909
#define RL_NEAREST 0x2600
#define RL_LINEAR 0x2601
#define RL_NEAREST_MIPMAP_NEAREST 0x2700
#define RL_LINEAR_MIPMAP_LINEAR 0x2703
#define RL_REPEAT 0x2901
#define RL_CLAMP_TO_EDGE 0x812F
#endif
