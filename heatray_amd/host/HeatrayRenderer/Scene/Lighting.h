//
//  Lighting.h
//  heatray_amd host layer
//
//  The scene's lights: up to 5 directional / point / spot lights plus the environment, kept tightly packed
//  in one hr_lights block that is re-uploaded whenever a light changes.  Public API of
//  /root/reference/Source/HeatrayRenderer/Scene/Lighting.h:26-108 (the OpenRL buffer / program binding
//  methods have no counterpart: the kernels read the block directly).
//

#pragma once

#include <HeatrayRenderer/Lights/Light.h>
#include <HeatrayRenderer/Lights/ShaderLightingDefines.h>

#include <functional>
#include <memory>
#include <string_view>

class EnvironmentLight;
class DirectionalLight;
class PointLight;
class SpotLight;

class Lighting
{
public:
    Lighting();
    ~Lighting() = default;

    // Remove every light.
    void clear();
    // Remove every analytic light, keep the environment.
    void clearAllButEnvironment();

    using LightCreatedCallback = std::function<void(std::shared_ptr<Light> light)>;
    void installLightCreatedCallback(LightCreatedCallback &&callback) { m_lightCreatedCallback = std::move(callback); }

    // Re-read a light's parameters and upload the block.
    void updateLight(std::shared_ptr<Light> light);
    void removeLight(std::shared_ptr<Light> light);

    std::shared_ptr<DirectionalLight> addDirectionalLight(const std::string_view name);
    const std::shared_ptr<DirectionalLight>* directionalLights() const { return &(m_directional.lights[0]); }
    void updateDirectionalLight(std::shared_ptr<DirectionalLight> light);
    void removeDirectionalLight(std::shared_ptr<DirectionalLight> light);

    std::shared_ptr<PointLight> addPointLight(const std::string_view name);
    const std::shared_ptr<PointLight>* pointLights() const { return &(m_point.lights[0]); }
    void updatePointLight(std::shared_ptr<PointLight> light);
    void removePointLight(std::shared_ptr<PointLight> light);

    std::shared_ptr<SpotLight> addSpotLight(const std::string_view name);
    const std::shared_ptr<SpotLight>* spotLights() const { return &(m_spot.lights[0]); }
    void updateSpotLight(std::shared_ptr<SpotLight> light);
    void removeSpotLight(std::shared_ptr<SpotLight> light);

    std::shared_ptr<EnvironmentLight> addEnvironmentLight();
    void removeEnvironmentLight();
    void updateEnvironmentLight(std::shared_ptr<EnvironmentLight> light);

    // The packed block as last uploaded (tests).
    const hr_lights& block() const { return m_block; }

private:
    void upload(); // bake every live light into the block and hand it to libhrcore

    template <class L, size_t N> struct Group {
        std::shared_ptr<L> lights[N] = { nullptr };
        int count = 0;
    };

    std::shared_ptr<EnvironmentLight> m_environment = nullptr;
    Group<DirectionalLight, ShaderLightingDefines::MAX_NUM_DIRECTIONAL_LIGHTS> m_directional;
    Group<PointLight, ShaderLightingDefines::MAX_NUM_POINT_LIGHTS> m_point;
    Group<SpotLight, ShaderLightingDefines::MAX_NUM_SPOT_LIGHTS> m_spot;

    hr_lights m_block{};
    LightCreatedCallback m_lightCreatedCallback;
};
