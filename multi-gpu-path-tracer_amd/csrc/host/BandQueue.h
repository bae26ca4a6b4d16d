// BandQueue.h -- the intra-frame work queue of the DQRB policy ("host-side atomic counter over ~64-128
// small bands, each GPU thread pulls the next band when its stream drains").  No counterpart in the reference, whose
// StreamThread renders one pre-assigned task per frame (StreamThread.h:64-104); it sits beside TaskGenerator.
#pragma once

#include <algorithm>
#include <atomic>
#include <numeric>
#include <vector>

#include "DevicePathTracer.h"

namespace ptamd {

struct BandQueue {
    std::vector<RenderTask> bands;      // full-width row bands, band 0 = bottom rows of the image (offset_y = 0)
    std::vector<int> order;             // pull order: position k of the queue hands out band order[k]
    std::vector<int> owner;             // device that rendered band b in the current frame (-1: nobody yet)
    std::vector<int> pulls;             // how many times band b was handed out in the current frame (must end at 1)
    std::atomic<int> next{0};

    // Bands of `rows` framebuffer rows each (a multiple of 8 = the kernel's tile height, so chroma rows never split),
    // sized so that the frame has 64..128 of them where its height allows.
    static int bandRowsFor(int height)
    {
        int rows = 8 * std::max(1, (height + 1023) / 1024);       // 1080 -> 16 rows (68 bands), 2160 -> 24 (90), 400 -> 8 (50)
        return rows;
    }

    void build(int width, int height, int rows = 0)
    {
        if (rows <= 0) rows = bandRowsFor(height);
        bands.clear();
        for (int y = 0; y < height; y += rows) bands.push_back(RenderTask{width, std::min(rows, height - y), 0, y});
        order.resize(bands.size());
        std::iota(order.begin(), order.end(), 0);
        owner.assign(bands.size(), -1);
        pulls.assign(bands.size(), 0);
        next.store(0);
    }

    // Before a frame: dearest band of the previous frame first (longest-processing-time order; a queue that ends on a
    // cheap band drains evenly), natural order while nothing has been timed.  Ties keep the lower band first.
    void beginFrame()
    {
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return bands[(size_t)a].time > bands[(size_t)b].time; });
        std::fill(owner.begin(), owner.end(), -1);
        std::fill(pulls.begin(), pulls.end(), 0);
        next.store(0);
    }

    // A worker's pull: the next band of the queue, or -1 when the frame has none left.
    int pull(int device)
    {
        int k = next.fetch_add(1);
        if (k >= (int)order.size()) return -1;
        int b = order[(size_t)k];
        owner[(size_t)b] = device;       // one writer per band: k is unique
        pulls[(size_t)b]++;
        return b;
    }

    bool everyBandExactlyOnce() const
    {
        for (size_t b = 0; b < bands.size(); b++) if (pulls[b] != 1 || owner[b] < 0) return false;
        return true;
    }
};

} // namespace ptamd
