//
//  ShaderLightingDefines.h
//  heatray_amd host layer
//
//  Capacity of the packed light blocks.  The block layout itself is hr_lights in include/hrcore.h, which
//  mirrors the *LightsBuffer structs of /root/reference/Source/HeatrayRenderer/Lights/ShaderLightingDefines.h:33-64.
//

#pragma once

#include <hrcore.h>

#include <stddef.h>

struct ShaderLightingDefines {
    static constexpr size_t MAX_NUM_DIRECTIONAL_LIGHTS = HR_MAX_DIRECTIONAL_LIGHTS;
    static constexpr size_t MAX_NUM_POINT_LIGHTS = HR_MAX_POINT_LIGHTS;
    static constexpr size_t MAX_NUM_SPOT_LIGHTS = HR_MAX_SPOT_LIGHTS;
};
