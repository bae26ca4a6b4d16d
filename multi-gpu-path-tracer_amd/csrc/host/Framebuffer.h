// Framebuffer.h -- RGB (3 B/px) + planar YUV420 frame, API of the reference's Framebuffer
// (Framebuffer.h:6-89).  The reference allocates both buffers with cudaMallocManaged and lets
// UVM page migration carry pixels from the GPUs to the host.  Here the host copy is plain
// memory; every DevicePathTracer renders into its own device-local buffers and copies the
// rectangles it rendered back explicitly (pt_fb_download_task) -- the explicit "gather".
#pragma once

#include <cstdint>
#include <vector>

#include "RendererConfig.h"

namespace ptamd {

class Framebuffer {
public:
    explicit Framebuffer(Resolution res) { initializePointers(res); resolution_ = res; }

    void setResolution(Resolution res)
    {
        if (res.width == resolution_.width && res.height == resolution_.height) return;
        initializePointers(res);
        resolution_ = res;
    }

    void initializePointers(Resolution res)
    {
        const size_t totalPixels = (size_t)res.width * res.height;
        rgb_.assign(totalPixels * 3, 0);
        yuv_.assign(totalPixels + 2 * (totalPixels / 4), 0);
    }

    Resolution getResolution() { return resolution_; }
    unsigned int getPixelCount() { return resolution_.width * resolution_.height; }
    uint8_t *getRGBPtr() { return rgb_.data(); }      // host-visible, valid after renderFrame()
    uint8_t *getYUVPtr() { return yuv_.data(); }

    // Framebuffer::updatePixel, Framebuffer.h:57-76 (BT.601 limited-range, 4:2:0 on even row/col)
    void updatePixel(int pixel_index, uint8_t r, uint8_t g, uint8_t b)
    {
        rgb_[3 * (size_t)pixel_index] = r;
        rgb_[3 * (size_t)pixel_index + 1] = g;
        rgb_[3 * (size_t)pixel_index + 2] = b;
        yuv_[pixel_index] = (uint8_t)(((66 * r + 129 * g + 25 * b + 128) >> 8) + 16);
        int W = (int)resolution_.width;
        int row = pixel_index / W, col = pixel_index % W;
        if (row % 2 == 0 && col % 2 == 0) {
            int total = W * (int)resolution_.height, uvSize = total / 4;
            int uvIndex = (row / 2) * (W / 2) + (col / 2);
            if (uvIndex < uvSize) {
                yuv_[total + uvIndex] = (uint8_t)(((-38 * r - 74 * g + 112 * b + 128) >> 8) + 128);
                yuv_[total + uvSize + uvIndex] = (uint8_t)(((112 * r - 94 * g - 18 * b + 128) >> 8) + 128);
            }
        }
    }

private:
    Resolution resolution_{0, 0};
    std::vector<uint8_t> rgb_, yuv_;
};

} // namespace ptamd
