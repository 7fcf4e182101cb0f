#include "SpotLight.h"

#include <glm/glm/gtx/quaternion.hpp>
#include <glm/glm/ext/scalar_constants.hpp>

#include <algorithm>
#include <assert.h>
#include <cmath>

SpotLight::SpotLight(const std::string_view name, size_t lightIndex)
: Light(name, Light::Type::kSpot)
, m_lightIndex(lightIndex)
{
    // Defaults of the reference's constructor (SpotLight.cpp:37-41).
    m_params.color = glm::vec3(1.0f);
    m_params.luminousIntensity = lightunits::WATTS_TO_LUMENS * (glm::pi<float>() * glm::pi<float>());
    m_params.position = glm::vec3(0.0f, 3.0f, 0.0f);
    m_params.outerAngle = glm::radians(40.0f);
    m_params.orientation.theta = glm::half_pi<float>();
}

// SpotLight.cpp:44-56 of the reference: cone angles are stored as cosines (x inner, y outer).
void SpotLight::copyToLightBuffer(hr_lights* block)
{
    assert(block && m_lightIndex < ShaderLightingDefines::MAX_NUM_SPOT_LIGHTS);
    const glm::vec3 direction = calculateDirection();
    const float watts = (m_params.luminousIntensity * lightunits::LUMENS_TO_WATTS) * glm::pi<float>();
    const glm::vec3 radiometric = m_params.color * watts;
    for (int k = 0; k < 3; ++k) {
        block->spot_positions[m_lightIndex][k] = m_params.position[k];
        block->spot_directions[m_lightIndex][k] = direction[k];
        block->spot_colors[m_lightIndex][k] = radiometric[k];
    }
    block->spot_angles[m_lightIndex][0] = std::cos(m_params.innerAngle);
    block->spot_angles[m_lightIndex][1] = std::cos(m_params.outerAngle);
}

// SpotLight.cpp:58-69 of the reference: keep inner < outer so that the smoothstep of the cone has a width.
void SpotLight::setParams(const Params& params)
{
    m_params = params;
    if (m_params.innerAngle > m_params.outerAngle) {
        m_params.innerAngle = std::max(0.0f, m_params.outerAngle - glm::radians(1.0f));
    }
    if ((m_params.innerAngle > 0.0f) && (m_params.innerAngle == m_params.outerAngle)) {
        m_params.innerAngle -= glm::radians(1.0f);
    }
}

// SpotLight.cpp:76-91 of the reference: direction the light shines in (FROM the light).
glm::vec3 SpotLight::calculateDirection()
{
    const glm::vec3 right = glm::vec3(1.0f, 0.0f, 0.0f);
    const glm::vec3 up = glm::vec3(0.0f, 1.0f, 0.0f);

    const glm::quat orientation = glm::angleAxis(m_params.orientation.theta, right) * glm::angleAxis(m_params.orientation.phi, up);
    const glm::mat4 viewMatrix = glm::mat4_cast(glm::inverse(orientation));

    const glm::vec3 forward = glm::vec3(viewMatrix[2][0], viewMatrix[2][1], viewMatrix[2][2]) * -1.0f;
    return glm::normalize(forward);
}
