// Light.h (heatray_amd host layer)
// Common base of the four light kinds (/root/reference/Source/HeatrayRenderer/Lights/Light.h:17-45).  A light has no OpenRL
// program or primitive here: it is a slot of the packed hr_lights block, addressed by (type, index) in the kernels.
#pragma once

#include "ShaderLightingDefines.h"

// The application's own headers rely on Utility/Log.h arriving through this file: in the reference it came in through
// Lights/Light.h:13 -> RLWrapper/Program.h:15 (and Error.h:11), and HeatrayRenderer.h:159,177 use LOG_ERROR without including it.
#include <Utility/Log.h>

#include <memory>
#include <string>
#include <string_view>

class Light
{
public:
    enum class Type { kEnvironment, kDirectional, kPoint, kSpot };

    virtual ~Light() = default;
    explicit Light(const std::string_view name, const Type type) : m_name(name), m_type(type) {}

    Type type() const { return m_type; }
    std::string_view name() const { return m_name; }

protected:
    const std::string m_name;
    const Type m_type;
};

namespace lightunits { // photometric <-> radiometric (683 lm/W), used by every analytic light
static constexpr float WATTS_TO_LUMENS = 683.0f, LUMENS_TO_WATTS = 1.0f / 683.0f;
} // namespace lightunits
