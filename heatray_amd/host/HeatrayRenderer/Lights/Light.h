//
//  Light.h
//  heatray_amd host layer
//
//  Base class of the four light kinds (/root/reference/Source/HeatrayRenderer/Lights/Light.h:17-45).
//  A light has no OpenRL program or primitive here: it is a slot of the packed hr_lights block,
//  addressed by (type, index) in the kernels.
//

#pragma once

#include "ShaderLightingDefines.h"

#include <memory>
#include <string>
#include <string_view>

class Light
{
public:
    enum class Type {
        kEnvironment,
        kDirectional,
        kPoint,
        kSpot
    };

    explicit Light(const std::string_view name, const Type type) : m_name(name), m_type(type) {}
    virtual ~Light() = default;

    std::string_view name() const { return m_name; }
    Type type() const { return m_type; }

protected:
    const std::string m_name;
    const Type m_type;
};

namespace lightunits {
// Photometric -> radiometric conversion used by every analytic light (683 lm/W).
static constexpr float WATTS_TO_LUMENS = 683.0f;
static constexpr float LUMENS_TO_WATTS = 1.0f / 683.0f;
} // namespace lightunits
