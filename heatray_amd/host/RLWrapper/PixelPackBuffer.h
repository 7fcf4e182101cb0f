//
//  PixelPackBuffer.h
//  heatray_amd host layer
//
//  The pixel hand-off object of PassCompleteCallback (/root/reference/Source/RLWrapper/PixelPackBuffer.h:20-91):
//  setPixelData() snapshots the accumulation buffer, mapPixelData() exposes it to the GL thread.
//  libhrcore copies the RGBA32F buffer into pinned host memory it owns (hr_readback); the pointer
//  stays valid until the next snapshot, resize or context destruction — the lifetime the viewer relies on.
//

#pragma once

#include "HrContext.h"
#include "RLTypes.h"

#include <assert.h>
#include <cmath>
#include <memory>

namespace openrl {

class PixelPackBuffer
{
public:
    ~PixelPackBuffer() { assert(!m_isMapped); }

    static std::shared_ptr<PixelPackBuffer> create(RLint sizeInBytes)
    {
        return std::shared_ptr<PixelPackBuffer>(new PixelPackBuffer(sizeInBytes));
    }

    // Snapshot the accumulation buffer (device -> pinned host copy).
    // complete = true: every pass requested so far is in the snapshot (libhrcore finishes the passes in its pipeline first).
    // complete = false (progressive display while passes accumulate): the snapshot holds the passes that are complete
    // already — alpha says how many — and the pipeline keeps running at full depth; if none is complete yet (right after a
    // reset) it falls back to a complete snapshot, so the viewer never receives an empty buffer.
    inline void setPixelData(bool complete = true)
    {
        assert(!m_isMapped);
        int32_t w = 0, h = 0;
        bool ok = false;
        if (!complete) {
            uint32_t passes = 0;
            ok = HRFunc(hr_readback_progressive(currentContext(), &m_pixels, &w, &h, &passes)) && passes > 0;
        }
        if (!ok) ok = HRFunc(hr_readback(currentContext(), &m_pixels, &w, &h));
        if (ok) {
            m_width = w;
            m_height = h;
        }
    }

    inline const float* mapPixelData() const
    {
        assert(!m_isMapped);
        m_isMapped = true;
        return m_pixels;
    }

    inline void unmapPixelData() const
    {
        assert(m_isMapped);
        m_isMapped = false;
    }

    // Extension: display-ready pixels straight from the device — displayGL.frag evaluated on the MI355X
    // (divide by the sample count, ACES / colour controls / vignette / exposure, sRGB) instead of uploading the RGBA32F
    // buffer through a PBO (HeatrayRenderer.cpp:328-344) and running the shader on the GL device.  Call it where
    // setPixelData()/mapPixelData() are called (PassGenerator's worker thread, inside the completion callback).
    // format: HR_DISPLAY_RGBA8 (4 x fewer bytes over PCIe), HR_DISPLAY_RGBA32F or HR_DISPLAY_HDR_RGBA32F (saveScreenshot).
    // The pointer stays valid until the next resolveForDisplay(), resize or context destruction.
    // complete = false: progressive, like setPixelData(false) (pixels without a complete pass yet come out black).
    // passesShown (optional): how many complete passes the returned image holds (a progressive image lags the requests).
    inline const void* resolveForDisplay(const hr_display_params& params, int32_t format = HR_DISPLAY_RGBA8, bool complete = true, uint32_t* passesShown = nullptr)
    {
        if (!complete) format |= HR_DISPLAY_PROGRESSIVE;
        const void* pixels = nullptr;
        int32_t w = 0, h = 0;
        if (HRFunc(hr_display_readback(currentContext(), &params, format, &pixels, &w, &h, passesShown))) {
            m_width = w;
            m_height = h;
        }
        return pixels;
    }

    // PostProcessingParams (HeatrayRenderer.h:104-117) -> the uniforms DisplayProgram::bind uploads (HeatrayRenderer.h:222-248)
    template <class PostProcessingParams>
    static hr_display_params displayParams(const PostProcessingParams& p)
    {
        hr_display_params d{};
        d.tonemapping_enabled = p.tonemapping_enabled ? 1 : 0;
        d.camera_exposure = std::pow(2.0f, p.exposure);
        d.brightness = p.brightness, d.contrast = p.contrast, d.hue = p.hue, d.saturation = p.saturation, d.vibrance = p.vibrance;
        d.red = p.red, d.green = p.green, d.blue = p.blue;
        d.vignette_intensity = p.vignetteIntensity, d.vignette_falloff = p.vignetteFalloff;
        return d;
    }

    inline RLint size() const { return m_sizeInBytes; }
    inline RLint width() const { return m_width; }
    inline RLint height() const { return m_height; }
    inline bool mapped() const { return m_isMapped; }

    static constexpr RLint kNumChannels = 4;

private:
    explicit PixelPackBuffer(RLint sizeInBytes) : m_sizeInBytes(sizeInBytes) {}

    const float* m_pixels = nullptr;
    RLint m_sizeInBytes = -1;
    RLint m_width = -1;
    RLint m_height = -1;
    mutable bool m_isMapped = false;
};

} // namespace openrl
