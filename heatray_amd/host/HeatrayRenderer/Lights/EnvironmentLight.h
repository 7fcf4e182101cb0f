//
//  EnvironmentLight.h
//  heatray_amd host layer
//
//  Lat/long environment map or solid colour; API of
//  /root/reference/Source/HeatrayRenderer/Lights/EnvironmentLight.h:26-65.
//

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
    explicit EnvironmentLight(const std::string_view name);
    ~EnvironmentLight() = default;

    // Load a lat/long image (needs the application's util::loadTexture; see the .cpp).
    void changeImageSource(const std::string_view path, bool builtInMap);
    // Use an already created texture as the environment (headless callers, tests).
    void setTexture(std::shared_ptr<openrl::Texture> texture, const std::string_view sourceName);

    static constexpr std::string_view SOLID_COLOR = "solid color";
    void enableSolidColor(const glm::vec3 &color);

    void rotate(const float theta_radians);
    void setExposure(const float exposureCompensation);

    void copyToLightBuffer(hr_lights* block);

private:
    std::shared_ptr<openrl::Texture> m_texture = nullptr;
    std::string m_textureSourcePath;
    glm::vec3 m_solidColor = glm::vec3(0.5f);

    float m_exposureCompensation = 0.0f;
    float m_thetaRotation = 0.0f; // radians
};
