#include "DirectionalLight.h"

#include <glm/glm/gtx/quaternion.hpp>

#include <assert.h>

DirectionalLight::DirectionalLight(const std::string_view name, size_t lightIndex)
: Light(name, Light::Type::kDirectional)
, m_lightIndex(lightIndex)
{
    // Defaults of the reference's constructor (DirectionalLight.cpp:37-40): 1 W * π straight down the -Z orientation.
    m_params.color = glm::vec3(1.0f);
    m_params.illuminance = lightunits::WATTS_TO_LUMENS * glm::pi<float>();
    m_params.orientation.phi = 0.0f;
    m_params.orientation.theta = glm::half_pi<float>();
}

// DirectionalLight.cpp:42-51 of the reference.
void DirectionalLight::copyToLightBuffer(hr_lights* block)
{
    assert(block && m_lightIndex < ShaderLightingDefines::MAX_NUM_DIRECTIONAL_LIGHTS);
    const glm::vec3 direction = calculateDirection();
    const glm::vec3 radiometric = m_params.color * (m_params.illuminance * lightunits::LUMENS_TO_WATTS);
    for (int k = 0; k < 3; ++k) {
        block->directional_directions[m_lightIndex][k] = direction[k];
        block->directional_colors[m_lightIndex][k] = radiometric[k];
    }
}

// DirectionalLight.cpp:64-78 of the reference: the light shines along the -Z axis of the
// (theta about X) * (phi about Y) orientation; the stored vector points TO the light.
glm::vec3 DirectionalLight::calculateDirection()
{
    const glm::vec3 right = glm::vec3(1.0f, 0.0f, 0.0f);
    const glm::vec3 up = glm::vec3(0.0f, 1.0f, 0.0f);

    const glm::quat orientation = glm::angleAxis(m_params.orientation.theta, right) * glm::angleAxis(m_params.orientation.phi, up);
    const glm::mat4 viewMatrix = glm::mat4_cast(glm::inverse(orientation));

    const glm::vec3 forward = glm::vec3(viewMatrix[2][0], viewMatrix[2][1], viewMatrix[2][2]) * -1.0f;
    return glm::normalize(forward) * -1.0f;
}
