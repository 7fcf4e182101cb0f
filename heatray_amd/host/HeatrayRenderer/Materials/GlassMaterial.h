//
//  GlassMaterial.h
//  heatray_amd host layer
//
//  Rough dielectric with Beer-Lambert absorption.  Parameters and defaults as in
//  /root/reference/Source/HeatrayRenderer/Materials/GlassMaterial.h:21-31.
//

#pragma once

#include "Material.h"

#include <RLWrapper/Texture.h>

#include <glm/glm/vec3.hpp>

class GlassMaterial final : public Material
{
public:
    GlassMaterial(const std::string_view name) : Material(name, Material::Type::Glass) {}
    virtual ~GlassMaterial() = default;

    struct Parameters {
        std::shared_ptr<openrl::Texture> baseColorTexture = nullptr;
        std::shared_ptr<openrl::Texture> normalmap = nullptr;
        std::shared_ptr<openrl::Texture> metallicRoughnessTexture = nullptr;
        glm::vec3 baseColor = glm::vec3(1.0f);  // linear transmission colour
        float roughness = 1.0f;                 // [0-1]
        float ior = 1.57f;                      // index of refraction
        float density = 0.05f;                  // absorption strength along the path inside the medium

        bool forceEnableAllTextures = false;
    };

    void build() override;
    void rebuild() override;
    void modify() override;

    Parameters& parameters() { return m_params; }

    static void bake(const Parameters& params, bool vertexColors, hr_material* row);

private:
    Parameters m_params;
};
