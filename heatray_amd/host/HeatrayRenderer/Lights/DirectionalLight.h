//
//  DirectionalLight.h
//  heatray_amd host layer
//
//  API of /root/reference/Source/HeatrayRenderer/Lights/DirectionalLight.h:25-72.
//

#pragma once

#include "Light.h"

#include <glm/glm/glm.hpp>

class DirectionalLight final : public Light
{
public:
    explicit DirectionalLight(const std::string_view name, size_t lightIndex);
    ~DirectionalLight() = default;

    struct Params {
        glm::vec3 color = glm::vec3(1.0f);
        float illuminance = 1.0f;

        struct Orientation {
            float phi = 0.0f;   // radians [0 - 2π]
            float theta = 0.0f; // radians [-π/2 - π/2]
        } orientation;
    };

    // Write this light's slot of the packed block: direction TO the light, radiometric colour.
    void copyToLightBuffer(hr_lights* block);

    Params params() const { return m_params; }
    void setParams(const Params &params) { m_params = params; }

    void updateLightIndex(const size_t newLightIndex) { m_lightIndex = newLightIndex; }

private:
    glm::vec3 calculateDirection();

    Params m_params;
    size_t m_lightIndex = 0;
};
