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

    // Snapshot the accumulation buffer (device -> pinned host copy; completes every pass in flight first).
    inline void setPixelData()
    {
        assert(!m_isMapped);
        int32_t w = 0, h = 0;
        if (HRFunc(hr_readback(currentContext(), &m_pixels, &w, &h))) {
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
