#include "PhysicallyBasedMaterial.h"
#include "MultiScatterUtil.h"

#include <assert.h>

namespace {
inline float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
inline hr_tex_id idOf(const std::shared_ptr<openrl::Texture>& t) { return t ? t->id() : HR_TEX_NONE; }
} // namespace

// Host-side conversion of the user parameters into the values the shading kernel reads, plus the
// shader-permutation flags.  Same arithmetic as PhysicallyBasedMaterial::modify and the #define
// selection of ::build in the reference
// (/root/reference/Source/HeatrayRenderer/Materials/PhysicallyBasedMaterial.cpp:127-146, 57-110).
void PhysicallyBasedMaterial::bake(const Parameters& p, bool vertexColors, hr_tex_id multiscatterLut, hr_material* row)
{
    constexpr float kMinRoughness  = 0.01f; // a perfectly smooth lobe is a Dirac delta
    constexpr float kMaxSpecularF0 = 0.08f; // Burley
    constexpr float kMaxClearcoat  = 0.2f;  // Burley

    *row = hr_material{};
    row->type = HR_MAT_PBR;
    for (int k = 0; k < 3; ++k) {
        row->base_color[k] = clamp01(p.baseColor[k]);
        row->emissive_color[k] = clamp01(p.emissiveColor[k]);
    }
    row->metallic = clamp01(p.metallic);
    row->roughness = clamp01(p.roughness) < kMinRoughness ? kMinRoughness : clamp01(p.roughness);
    row->specular_f0 = p.specularF0 * kMaxSpecularF0;
    row->roughness_alpha = row->roughness * row->roughness;
    row->clear_coat = p.clearCoat * kMaxClearcoat;
    row->clear_coat_roughness = clamp01(p.clearCoatRoughness) < kMinRoughness ? kMinRoughness : clamp01(p.clearCoatRoughness);
    row->clear_coat_roughness_alpha = row->clear_coat_roughness * row->clear_coat_roughness;

    uint32_t flags = 0;
    if (p.baseColorTexture || p.forceEnableAllTextures)          flags |= HR_MF_HAS_BASE_COLOR_TEXTURE;
    if (p.metallicRoughnessTexture || p.forceEnableAllTextures)  flags |= HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE;
    if (p.emissiveTexture)                                       flags |= HR_MF_HAS_EMISSIVE_TEXTURE;
    if (p.normalmap)                                             flags |= HR_MF_HAS_NORMALMAP;
    if (p.clearCoatTexture || p.forceEnableAllTextures)          flags |= HR_MF_HAS_CLEARCOAT_TEXTURE;
    if (p.clearCoatRoughnessTexture || p.forceEnableAllTextures) flags |= HR_MF_HAS_CLEARCOAT_ROUGHNESS_TEXTURE;
    if (p.clearCoatNormalmap)                                    flags |= HR_MF_HAS_CLEARCOAT_NORMALMAP;
    if (p.doubleSided)                                           flags |= HR_MF_DOUBLE_SIDED;
    if (p.alphaMask)                                             flags |= HR_MF_ALPHA_MASK;
    if (vertexColors)                                            flags |= HR_MF_VERTEX_COLORS;
    row->flags = flags;

    // An unbound slot reads the dummy white texel (HR_TEX_NONE samples as white in libhrcore).
    row->base_color_texture = idOf(p.baseColorTexture);
    row->metallic_roughness_texture = idOf(p.metallicRoughnessTexture);
    row->emissive_texture = idOf(p.emissiveTexture);
    row->normalmap = idOf(p.normalmap);
    row->clear_coat_texture = idOf(p.clearCoatTexture);
    row->clear_coat_roughness_texture = idOf(p.clearCoatRoughnessTexture);
    row->clear_coat_normalmap = idOf(p.clearCoatNormalmap);
    row->multiscatter_lut = multiscatterLut;
}

void PhysicallyBasedMaterial::build()
{
    m_multiscatterLUT = loadMultiscatterTexture();
    if (m_tableIndex < 0) {
        m_tableIndex = allocateTableIndex();
    }
    modify();
}

void PhysicallyBasedMaterial::rebuild()
{
    build();
}

void PhysicallyBasedMaterial::modify()
{
    assert(m_tableIndex >= 0 && "modify() before build()");
    hr_material row;
    bake(m_params, m_enableVertexColors, m_multiscatterLUT ? m_multiscatterLUT->id() : HR_TEX_NONE, &row);
    HRFunc(hr_material_set(openrl::currentContext(), m_tableIndex, &row));
}
