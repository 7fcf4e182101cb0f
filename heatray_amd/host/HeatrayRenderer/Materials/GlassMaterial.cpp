#include "GlassMaterial.h"

#include <assert.h>
#include <cmath>

namespace {
inline float clamp01(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
inline hr_tex_id idOf(const std::shared_ptr<openrl::Texture>& t) { return t ? t->id() : HR_TEX_NONE; }
} // namespace

// GlassMaterial::modify + the permutation flags of ::build
// (/root/reference/Source/HeatrayRenderer/Materials/GlassMaterial.cpp:88-107, 50-77).
void GlassMaterial::bake(const Parameters& p, bool vertexColors, hr_material* row)
{
    constexpr float kMinRoughness = 0.01f;

    *row = hr_material{};
    row->type = HR_MAT_GLASS;
    for (int k = 0; k < 3; ++k) {
        row->base_color[k] = clamp01(p.baseColor[k]);
    }
    row->roughness = clamp01(p.roughness) < kMinRoughness ? kMinRoughness : clamp01(p.roughness);
    row->roughness_alpha = row->roughness * row->roughness;
    row->density = p.density;
    row->ior = p.ior < 0.0f ? 0.0f : p.ior;
    const float f0 = std::fabs((1.0f - row->ior) / (1.0f + row->ior)); // normal-incidence reflectance of the interface
    row->specular_f0 = f0 * f0;

    uint32_t flags = 0;
    if (p.baseColorTexture || p.forceEnableAllTextures)         flags |= HR_MF_HAS_BASE_COLOR_TEXTURE;
    if (p.metallicRoughnessTexture || p.forceEnableAllTextures) flags |= HR_MF_HAS_METALLIC_ROUGHNESS_TEXTURE;
    if (p.normalmap)                                            flags |= HR_MF_HAS_NORMALMAP;
    if (vertexColors)                                           flags |= HR_MF_VERTEX_COLORS;
    row->flags = flags;

    row->base_color_texture = idOf(p.baseColorTexture);
    row->normalmap = idOf(p.normalmap);
    row->metallic_roughness_texture = idOf(p.metallicRoughnessTexture);
    row->emissive_texture = row->clear_coat_texture = row->clear_coat_roughness_texture = row->clear_coat_normalmap = HR_TEX_NONE;
    row->multiscatter_lut = HR_TEX_NONE;
}

void GlassMaterial::build()
{
    if (m_tableIndex < 0) {
        m_tableIndex = allocateTableIndex();
    }
    modify();
}

void GlassMaterial::rebuild()
{
    build();
}

void GlassMaterial::modify()
{
    assert(m_tableIndex >= 0 && "modify() before build()");
    hr_material row;
    bake(m_params, m_enableVertexColors, &row);
    HRFunc(hr_material_set(openrl::currentContext(), m_tableIndex, &row));
}
