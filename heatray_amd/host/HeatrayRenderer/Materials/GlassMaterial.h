// GlassMaterial.h (heatray_amd host layer)
// Rough dielectric with Beer-Lambert absorption; parameter names and defaults are those of
// /root/reference/Source/HeatrayRenderer/Materials/GlassMaterial.h:21-31 (the UI edits them in place).
#pragma once

#include "Material.h"

#include <RLWrapper/Texture.h>

#include <glm/glm/vec3.hpp>

class GlassMaterial final : public Material
{
public:
    struct Parameters {
        TexturePtr baseColorTexture, normalmap, metallicRoughnessTexture;
        glm::vec3 baseColor = glm::vec3(1.0f);                 // linear transmission colour
        float roughness = 1.0f, ior = 1.57f, density = 0.05f;  // [0-1]; index of refraction; absorption strength inside the medium
        bool forceEnableAllTextures = false;
    };

    GlassMaterial(const std::string_view name) : Material(name, Material::Type::Glass) {}
    virtual ~GlassMaterial() = default;

    void build() override;
    void rebuild() override;
    void modify() override;
    Parameters& parameters() { return m_params; }

    // host-side baking of the table row (no device access): exposed for the tests
    static void bake(const Parameters& params, bool vertexColors, hr_material* row);

private:
    Parameters m_params;
};
