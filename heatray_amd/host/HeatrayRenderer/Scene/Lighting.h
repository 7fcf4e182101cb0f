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

class DirectionalLight;
class EnvironmentLight;
class PointLight;
class SpotLight;

class Lighting
{
public:
    using LightCreatedCallback = std::function<void(std::shared_ptr<Light> light)>;

    Lighting();
    ~Lighting() = default;

    void installLightCreatedCallback(LightCreatedCallback &&callback) { m_lightCreatedCallback = std::move(callback); }
    void clear();                  // remove every light
    void clearAllButEnvironment(); // remove every analytic light, keep the environment
    void updateLight(std::shared_ptr<Light> light); // re-read a light's parameters and upload the block
    void removeLight(std::shared_ptr<Light> light);
    const hr_lights& block() const { return m_block; } // the packed block as last uploaded (tests)

    // one add / list / update / remove quartet per analytic light kind, as in the reference
    std::shared_ptr<DirectionalLight> addDirectionalLight(const std::string_view name);
    std::shared_ptr<PointLight> addPointLight(const std::string_view name);
    std::shared_ptr<SpotLight> addSpotLight(const std::string_view name);
    const std::shared_ptr<DirectionalLight>* directionalLights() const { return &(m_directional.lights[0]); }
    const std::shared_ptr<PointLight>* pointLights() const { return &(m_point.lights[0]); }
    const std::shared_ptr<SpotLight>* spotLights() const { return &(m_spot.lights[0]); }
    void updateDirectionalLight(std::shared_ptr<DirectionalLight> light);
    void updatePointLight(std::shared_ptr<PointLight> light);
    void updateSpotLight(std::shared_ptr<SpotLight> light);
    void removeDirectionalLight(std::shared_ptr<DirectionalLight> light);
    void removePointLight(std::shared_ptr<PointLight> light);
    void removeSpotLight(std::shared_ptr<SpotLight> light);

    std::shared_ptr<EnvironmentLight> addEnvironmentLight();
    void updateEnvironmentLight(std::shared_ptr<EnvironmentLight> light);
    void removeEnvironmentLight();

private:
    template <class L, size_t N> struct Group {
        std::shared_ptr<L> lights[N] = { nullptr };
        int count = 0;
    };

    void upload(); // bake every live light into the block and hand it to libhrcore

    Group<DirectionalLight, ShaderLightingDefines::MAX_NUM_DIRECTIONAL_LIGHTS> m_directional;
    Group<PointLight, ShaderLightingDefines::MAX_NUM_POINT_LIGHTS> m_point;
    Group<SpotLight, ShaderLightingDefines::MAX_NUM_SPOT_LIGHTS> m_spot;
    std::shared_ptr<EnvironmentLight> m_environment;
    LightCreatedCallback m_lightCreatedCallback;
    hr_lights m_block{};
};
