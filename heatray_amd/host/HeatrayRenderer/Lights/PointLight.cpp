#include "PointLight.h"

#include <glm/glm/ext/scalar_constants.hpp>

#include <assert.h>

PointLight::PointLight(const std::string_view name, size_t lightIndex)
: Light(name, Light::Type::kPoint)
, m_lightIndex(lightIndex)
{
    // Defaults of the reference's constructor (PointLight.cpp:36-38): 1 W * 4π, three units above the origin.
    m_params.color = glm::vec3(1.0f);
    m_params.luminousIntensity = lightunits::WATTS_TO_LUMENS * (4.0f * glm::pi<float>());
    m_params.position = glm::vec3(0.0f, 3.0f, 0.0f);
}

// PointLight.cpp:41-50 of the reference.
void PointLight::copyToLightBuffer(hr_lights* block)
{
    assert(block && m_lightIndex < ShaderLightingDefines::MAX_NUM_POINT_LIGHTS);
    const float watts = (m_params.luminousIntensity * lightunits::LUMENS_TO_WATTS) * (4.0f * glm::pi<float>());
    const glm::vec3 radiometric = m_params.color * watts;
    for (int k = 0; k < 3; ++k) {
        block->point_positions[m_lightIndex][k] = m_params.position[k];
        block->point_colors[m_lightIndex][k] = radiometric[k];
    }
}
