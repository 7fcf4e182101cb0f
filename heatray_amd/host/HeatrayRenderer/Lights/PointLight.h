//
//  PointLight.h
//  heatray_amd host layer
//
//  API of /root/reference/Source/HeatrayRenderer/Lights/PointLight.h:24-60.
//

#pragma once

#include "Light.h"

#include <glm/glm/glm.hpp>

class PointLight final : public Light
{
public:
    explicit PointLight(const std::string_view name, size_t lightIndex);
    ~PointLight() = default;

    struct Params {
        glm::vec3 color = glm::vec3(1.0f);
        glm::vec3 position = glm::vec3(0.0f);
        float luminousIntensity = 1.0f;
    };

    void copyToLightBuffer(hr_lights* block);

    Params params() const { return m_params; }
    void setParams(const Params &params) { m_params = params; }

    void updateLightIndex(const size_t newLightIndex) { m_lightIndex = newLightIndex; }

private:
    Params m_params;
    size_t m_lightIndex = 0;
};
