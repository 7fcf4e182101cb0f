// EnvironmentLight.h (heatray_amd host layer)
// Lat/long image or solid colour surrounding the scene; public surface of
// /root/reference/Source/HeatrayRenderer/Lights/EnvironmentLight.h (plus setTexture for headless callers).
#pragma once

#include "Light.h"

#include <RLWrapper/Texture.h>

#include <glm/glm/vec3.hpp>

#include <memory>
#include <string>
#include <string_view>

class EnvironmentLight final : public Light
{
public:
    static constexpr std::string_view SOLID_COLOR = "solid color";

    explicit EnvironmentLight(const std::string_view name);
    ~EnvironmentLight() = default;

    void changeImageSource(const std::string_view path, bool builtInMap); // needs the application's util::loadTexture (see the .cpp)
    void setTexture(std::shared_ptr<openrl::Texture> texture, const std::string_view sourceName); // an already created texture
    void enableSolidColor(const glm::vec3 &color);
    void setExposure(const float exposureCompensation);
    void rotate(const float theta_radians);
    void copyToLightBuffer(hr_lights* block);

private:
    float m_thetaRotation = 0.0f; // radians
    float m_exposureCompensation = 0.0f;
    glm::vec3 m_solidColor = glm::vec3(0.5f);
    std::string m_textureSourcePath;
    std::shared_ptr<openrl::Texture> m_texture;
};
