// PointLight.h (heatray_amd host layer); Params as in /root/reference/Source/HeatrayRenderer/Lights/PointLight.h
#pragma once

#include "Light.h"

#include <glm/glm/glm.hpp>

class PointLight final : public Light
{
public:
    struct Params {
        glm::vec3 color = glm::vec3(1.0f), position = glm::vec3(0.0f);
        float luminousIntensity = 1.0f;
    };

    explicit PointLight(const std::string_view name, size_t lightIndex);
    ~PointLight() = default;

    Params params() const { return m_params; }
    void setParams(const Params &params) { m_params = params; }
    void updateLightIndex(const size_t newLightIndex) { m_lightIndex = newLightIndex; }
    void copyToLightBuffer(hr_lights* block);

private:
    size_t m_lightIndex = 0;
    Params m_params;
};
