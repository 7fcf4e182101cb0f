// DirectionalLight.h (heatray_amd host layer); Params as in /root/reference/Source/HeatrayRenderer/Lights/DirectionalLight.h
#pragma once

#include "Light.h"

#include <glm/glm/glm.hpp>

class DirectionalLight final : public Light
{
public:
    struct Params {
        glm::vec3 color = glm::vec3(1.0f);
        float illuminance = 1.0f;
        struct Orientation {
            float phi = 0.0f, theta = 0.0f; // radians: [0, 2 pi] and [-pi/2, pi/2]
        } orientation;
    };

    explicit DirectionalLight(const std::string_view name, size_t lightIndex);
    ~DirectionalLight() = default;

    Params params() const { return m_params; }
    void setParams(const Params &params) { m_params = params; }
    void updateLightIndex(const size_t newLightIndex) { m_lightIndex = newLightIndex; }
    void copyToLightBuffer(hr_lights* block); // this light's slot: direction TO the light, radiometric colour

private:
    glm::vec3 calculateDirection();

    size_t m_lightIndex = 0;
    Params m_params;
};
