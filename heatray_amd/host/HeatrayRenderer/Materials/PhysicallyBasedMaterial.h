// PhysicallyBasedMaterial.h (heatray_amd host layer)
// Opaque microfacet material: diffuse + GGX specular + clear coat + multiscatter compensation.  Parameter names and defaults
// are those of /root/reference/Source/HeatrayRenderer/Materials/PhysicallyBasedMaterial.h:22-41.
#pragma once

#include "Material.h"

#include <RLWrapper/Texture.h>

#include <glm/glm/vec3.hpp>
#include <memory>

class PhysicallyBasedMaterial final : public Material
{
public:
    struct Parameters {
        TexturePtr baseColorTexture, emissiveTexture, normalmap, metallicRoughnessTexture;
        TexturePtr clearCoatTexture, clearCoatRoughnessTexture, clearCoatNormalmap;
        glm::vec3 baseColor = glm::vec3(1.0f), emissiveColor = glm::vec3(0.0f); // linear; albedo (dielectric) or specular colour (conductor)
        float roughness = 1.0f, metallic = 0.0f;              // [0-1]
        float specularF0 = 0.5f, clearCoat = 0.0f;            // [0-1], scaled to [0-0.08] and [0-0.2]
        float clearCoatRoughness = 0.0f;                      // [0-1]
        bool doubleSided = true, alphaMask = false;           // shade back faces with the flipped normal; cut out alpha < 1 texels
        bool forceEnableAllTextures = false;                  // enable every texture slot even when no texture is bound
    };

    explicit PhysicallyBasedMaterial(const std::string_view name) : Material(name, Material::Type::PBR) {}
    virtual ~PhysicallyBasedMaterial() = default;

    void build() override;
    void rebuild() override;
    void modify() override;
    Parameters& parameters() { return m_params; }

    // host-side baking of the table row (no device access): exposed for the tests
    static void bake(const Parameters& params, bool vertexColors, hr_tex_id multiscatterLut, hr_material* row);

private:
    Parameters m_params;
    TexturePtr m_multiscatterLUT;
};
