#include "EnvironmentLight.h"

#if __has_include(<Utility/TextureLoader.h>)
#include <Utility/TextureLoader.h> // the application's stb / FreeImage loader (untouched reference code)
#define HR_HOST_HAS_TEXTURE_LOADER 1
#endif

#include <assert.h>
#include <cmath>
#include <stdio.h>

EnvironmentLight::EnvironmentLight(const std::string_view name)
: Light(name, Light::Type::kEnvironment)
{
}

// EnvironmentLight.cpp:30-46 of the reference.
void EnvironmentLight::changeImageSource(const std::string_view path, bool builtInMap)
{
    assert(!path.empty());
    std::string fullPath = builtInMap ? std::string("Resources/Environments/") + std::string(path) : std::string(path);
    if (m_textureSourcePath != fullPath) {
#if defined(HR_HOST_HAS_TEXTURE_LOADER)
        m_texture = util::loadTexture(fullPath);
        m_textureSourcePath = fullPath;
#else
        fprintf(stderr, "EnvironmentLight: no image loader in this build, cannot load %s (use setTexture)\n", fullPath.c_str());
#endif
    }
}

void EnvironmentLight::setTexture(std::shared_ptr<openrl::Texture> texture, const std::string_view sourceName)
{
    m_texture = std::move(texture);
    m_textureSourcePath = std::string(sourceName);
}

// A solid colour is a 1x1 RGB float texture with clamped, linear sampling (EnvironmentLight.cpp:48-72 of the reference).
void EnvironmentLight::enableSolidColor(const glm::vec3& color)
{
    const bool unchanged = (m_textureSourcePath == SOLID_COLOR) && (m_solidColor == color);
    if (unchanged) return;
    openrl::Texture::Descriptor onePixel;
    onePixel.width = onePixel.height = 1;
    onePixel.format = onePixel.internalFormat = RL_RGB;
    onePixel.dataType = RL_FLOAT;
    openrl::Texture::Sampler clamped;
    clamped.minFilter = clamped.magFilter = RL_LINEAR;
    clamped.wrapS = clamped.wrapT = RL_CLAMP_TO_EDGE;
    const float texel[3] = { color.x, color.y, color.z };
    m_texture = openrl::Texture::create(texel, onePixel, clamped, false);
    m_solidColor = color;
    m_textureSourcePath = std::string(SOLID_COLOR);
}

void EnvironmentLight::setExposure(const float exposureCompensation) { m_exposureCompensation = exposureCompensation; }
void EnvironmentLight::rotate(const float theta_radians) { m_thetaRotation = theta_radians; }

// EnvironmentLight.cpp:84-98 of the reference: exposure compensation is applied as 2^stops.
void EnvironmentLight::copyToLightBuffer(hr_lights* block)
{
    assert(block);
    block->env_enabled = 1;
    block->env_texture = m_texture ? m_texture->id() : HR_TEX_NONE;
    block->env_exposure = std::pow(2.0f, m_exposureCompensation);
    block->env_theta_rotation = m_thetaRotation;
}
