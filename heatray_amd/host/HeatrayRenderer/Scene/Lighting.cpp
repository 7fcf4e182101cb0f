#include "Lighting.h"

#include <HeatrayRenderer/Lights/EnvironmentLight.h>
#include <HeatrayRenderer/Lights/DirectionalLight.h>
#include <HeatrayRenderer/Lights/PointLight.h>
#include <HeatrayRenderer/Lights/SpotLight.h>

#include <RLWrapper/HrContext.h>

#include <assert.h>
#include <stdio.h>
#include <utility>

Lighting::Lighting()
{
    clear();
}

void Lighting::clear()
{
    m_environment.reset();
    clearAllButEnvironment();
}

void Lighting::clearAllButEnvironment()
{
    for (auto& l : m_directional.lights) l.reset();
    for (auto& l : m_point.lights) l.reset();
    for (auto& l : m_spot.lights) l.reset();
    m_directional.count = m_point.count = m_spot.count = 0;
    upload();
}

// The block is rebuilt from the live lights on every change, so it is tightly packed by construction
// (the reference patches slots in place and swaps the last light into a removed slot, Lighting.cpp:192-381).
void Lighting::upload()
{
    hr_lights block{};
    block.env_texture = HR_TEX_NONE;
    block.n_directional = m_directional.count;
    for (int i = 0; i < m_directional.count; ++i) m_directional.lights[i]->copyToLightBuffer(&block);
    block.n_point = m_point.count;
    for (int i = 0; i < m_point.count; ++i) m_point.lights[i]->copyToLightBuffer(&block);
    block.n_spot = m_spot.count;
    for (int i = 0; i < m_spot.count; ++i) m_spot.lights[i]->copyToLightBuffer(&block);
    if (m_environment) {
        m_environment->copyToLightBuffer(&block);
    }
    m_block = block;
    if (openrl::currentContext()) {
        HRFunc(hr_lights_set(openrl::currentContext(), &m_block));
    }
}

void Lighting::updateLight(std::shared_ptr<Light> light)
{
    (void)light; // every live light is re-baked
    upload();
}

void Lighting::removeLight(std::shared_ptr<Light> light)
{
    switch (light->type()) {
        case Light::Type::kEnvironment:
            removeEnvironmentLight();
            break;
        case Light::Type::kDirectional:
            removeDirectionalLight(std::static_pointer_cast<DirectionalLight>(light));
            break;
        case Light::Type::kPoint:
            removePointLight(std::static_pointer_cast<PointLight>(light));
            break;
        case Light::Type::kSpot:
            removeSpotLight(std::static_pointer_cast<SpotLight>(light));
            break;
    }
}

std::shared_ptr<EnvironmentLight> Lighting::addEnvironmentLight()
{
    m_environment = std::make_shared<EnvironmentLight>("Environment");
    upload();
    if (m_lightCreatedCallback) {
        m_lightCreatedCallback(m_environment);
    }
    return m_environment;
}

void Lighting::removeEnvironmentLight()
{
    m_environment.reset();
    upload();
}

void Lighting::updateEnvironmentLight(std::shared_ptr<EnvironmentLight> light)
{
    m_environment = light;
    upload();
}

namespace {

// Append a light to a group; nullptr (and a log line) when the group is full, like the reference.
template <class L, class G>
std::shared_ptr<L> addToGroup(G& group, size_t capacity, const std::string_view name, const char* what)
{
    if ((size_t)group.count >= capacity) {
        fprintf(stderr, "Attempting to add too many %s Lights!\n", what);
        return nullptr;
    }
    std::shared_ptr<L> light = std::make_shared<L>(name, (size_t)group.count);
    group.lights[group.count++] = light;
    return light;
}

// Remove a light and keep the group compact: the last light moves into the freed slot (Lighting.cpp:301-333).
template <class L, class G>
void removeFromGroup(G& group, const std::shared_ptr<L>& light)
{
    int index = 0;
    for (; index < group.count; ++index) {
        if (group.lights[index] == light) break;
    }
    assert(index < group.count);
    if (index >= group.count) return;
    const int last = group.count - 1;
    if (index != last) {
        std::swap(group.lights[index], group.lights[last]);
        group.lights[index]->updateLightIndex((size_t)index);
    }
    group.lights[last].reset();
    --group.count;
}

} // namespace

std::shared_ptr<DirectionalLight> Lighting::addDirectionalLight(const std::string_view name)
{
    auto light = addToGroup<DirectionalLight>(m_directional, ShaderLightingDefines::MAX_NUM_DIRECTIONAL_LIGHTS, name, "Directional");
    if (light) {
        upload();
        if (m_lightCreatedCallback) m_lightCreatedCallback(light);
    }
    return light;
}

void Lighting::updateDirectionalLight(std::shared_ptr<DirectionalLight> light) { updateLight(light); }

void Lighting::removeDirectionalLight(std::shared_ptr<DirectionalLight> light)
{
    removeFromGroup(m_directional, light);
    upload();
}

std::shared_ptr<PointLight> Lighting::addPointLight(const std::string_view name)
{
    auto light = addToGroup<PointLight>(m_point, ShaderLightingDefines::MAX_NUM_POINT_LIGHTS, name, "Point");
    if (light) {
        upload();
        if (m_lightCreatedCallback) m_lightCreatedCallback(light);
    }
    return light;
}

void Lighting::updatePointLight(std::shared_ptr<PointLight> light) { updateLight(light); }

void Lighting::removePointLight(std::shared_ptr<PointLight> light)
{
    removeFromGroup(m_point, light);
    upload();
}

std::shared_ptr<SpotLight> Lighting::addSpotLight(const std::string_view name)
{
    auto light = addToGroup<SpotLight>(m_spot, ShaderLightingDefines::MAX_NUM_SPOT_LIGHTS, name, "Spot");
    if (light) {
        upload();
        if (m_lightCreatedCallback) m_lightCreatedCallback(light);
    }
    return light;
}

void Lighting::updateSpotLight(std::shared_ptr<SpotLight> light) { updateLight(light); }

void Lighting::removeSpotLight(std::shared_ptr<SpotLight> light)
{
    removeFromGroup(m_spot, light);
    upload();
}
