#include "MultiScatterUtil.h"

#include <vector>

namespace {
std::weak_ptr<openrl::Texture> multiscatterTexture;
} // namespace

void generateMultiScatterTexture()
{
    // Drop the cached texture: the next loadMultiscatterTexture() integrates it again.
    multiscatterTexture.reset();
}

std::shared_ptr<openrl::Texture> loadMultiscatterTexture()
{
    std::shared_ptr<openrl::Texture> texture = multiscatterTexture.lock();
    if (!texture) {
        // Integrate on the device, read the 64 KB table back and wrap it like any other texture
        // (LINEAR + CLAMP_TO_EDGE, the sampler util::loadTexture(..., generateMips=false) produces).
        std::vector<float> lut(128 * 128);
        if (!HRFunc(hr_multiscatter_lut_generate(openrl::currentContext(), lut.data(), nullptr))) {
            return nullptr;
        }
        openrl::Texture::Descriptor desc;
        desc.dataType = RL_FLOAT;
        desc.format = RL_LUMINANCE;
        desc.internalFormat = RL_LUMINANCE;
        desc.width = desc.height = 128;
        openrl::Texture::Sampler sampler;
        sampler.magFilter = RL_LINEAR;
        sampler.minFilter = RL_LINEAR;
        sampler.wrapS = RL_CLAMP_TO_EDGE;
        sampler.wrapT = RL_CLAMP_TO_EDGE;
        texture = openrl::Texture::create(lut.data(), desc, sampler, false);
        multiscatterTexture = texture;
    }
    return texture;
}
