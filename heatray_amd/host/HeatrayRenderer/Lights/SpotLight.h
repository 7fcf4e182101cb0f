//
//  SpotLight.h
//  heatray_amd host layer
//
//  API of /root/reference/Source/HeatrayRenderer/Lights/SpotLight.h:24-70.
//

#pragma once

#include "Light.h"

#include <glm/glm/glm.hpp>

class SpotLight final : public Light
{
public:
    explicit SpotLight(const std::string_view name, size_t lightIndex);
    ~SpotLight() = default;

    struct Params {
        glm::vec3 color = glm::vec3(1.0f);
        glm::vec3 position = glm::vec3(0.0f);
        float luminousIntensity = 1.0f;
        float innerAngle = 0.0f;
        float outerAngle = 0.0f;

        struct Orientation {
            float phi = 0.0f;   // radians [0 - 2π]
            float theta = 0.0f; // radians [-π/2 - π/2]
        } orientation;
    };

    void copyToLightBuffer(hr_lights* block);

    Params params() const { return m_params; }
    void setParams(const Params &params);

    void updateLightIndex(const size_t newLightIndex) { m_lightIndex = newLightIndex; }

private:
    glm::vec3 calculateDirection();

    Params m_params;
    size_t m_lightIndex = 0;
};
