// DevicePathTracer.h -- the per-GPU engine with the reference's public surface
// (DevicePathTracer.h:167-392) as a thin C++ wrapper over the C ABI (include/pt_amd.h).
// No HIP types appear here: cudaStream_t -> pt_stream, dim3 -> Dim3.
#pragma once

#include <memory>
#include <mutex>
#include <utility>
#include <vector>

#include "../../../include/pt_amd.h"
#include "Framebuffer.h"
#include "HostScene.h"
#include "RendererConfig.h"
#include "cuda_utils.h"

namespace ptamd {

// DevicePathTracer.h:19-25
struct RenderTask {
    int width;
    int height;
    int offset_x;
    int offset_y;
    int time = 0;      // wall milliseconds of the last render of this task (StreamThread.h:87-90)
};

class DevicePathTracer {
public:
    DevicePathTracer(int device_idx, unsigned int samplesPerPixel, unsigned int recursionDepth, Dim3 threadBlockSize,
                     HostScene &hostScene, std::shared_ptr<Framebuffer> framebuffer, CameraConfig &cameraConfig)
        : device_idx_{device_idx}, threadBlockSize_{threadBlockSize}, hostScene_{hostScene},
          samplesPerPixel_{samplesPerPixel}, recursionDepth_{recursionDepth}, framebuffer_{framebuffer},
          cameraConfig_{cameraConfig}
    {
        checkPtErrors(pt_device_create(device_idx_, &dev_));
        checkPtErrors(pt_set_thread_block_size(dev_, threadBlockSize_.x, threadBlockSize_.y));
        reloadWorld();
        reloadCamera();
        setFramebuffer(framebuffer_);
    }

    DevicePathTracer(const DevicePathTracer &) = delete;
    DevicePathTracer &operator=(const DevicePathTracer &) = delete;

    // asynchronous on `stream`; a zero-width task is a no-op (reference :194-214)
    void renderTaskAsync(RenderTask &task, pt_stream stream)
    {
        if (task.width == 0) return;
        pt_task t{task.width, task.height, task.offset_x, task.offset_y};
        pt_camera c = cameraByValue();
        checkPtErrors(pt_render_task_async(dev_, &t, &c, samplesPerPixel_, recursionDepth_, stream));
    }

    void waitForRenderTask() { checkPtErrors(pt_device_sync(dev_)); }
    void synchronizeStream(pt_stream stream) { checkPtErrors(pt_stream_sync(dev_, stream)); }

    // Copies the finished task's pixels into the shared host Framebuffer (replaces UVM).
    void gatherTask(const RenderTask &task, pt_stream stream)
    {
        if (task.width == 0) return;
        pt_task t{task.width, task.height, task.offset_x, task.offset_y};
        checkPtErrors(pt_fb_download_task(dev_, &t, framebuffer_->getRGBPtr(), framebuffer_->getYUVPtr(), stream));
    }

    // ---- frames in flight (RendererConfig::framesInFlight > 1; no counterpart in the reference, whose renderFrame ends on
    // a barrier).  A launch renders into the device framebuffer bound when it was enqueued and pt_bind_framebuffer is a
    // pointer swap, so frame i goes into slot i % (F + 1) of a ring of device framebuffers while frame i - F + 1 is copied
    // out of its own slot; the library keeps one tile schedule per framebuffer, so launches into different slots overlap.
    // The workers of one GPU share the engine's binding: bind + launch and bind + copy are atomic under bindMu_.
    void setFramesInFlight(unsigned frames)
    {
        std::lock_guard<std::mutex> lk(bindMu_);
        for (size_t i = 1; i < ring_.size(); i++) pt_fb_free(dev_, ring_[i].first, ring_[i].second);
        ring_.assign(1, {d_rgb_, d_yuv_});
        Resolution r = framebuffer_->getResolution();
        for (unsigned i = 0; frames > 1 && i < frames; i++) {
            uint8_t *rgb = nullptr, *yuv = nullptr;
            checkPtErrors(pt_fb_alloc(dev_, (int)r.width, (int)r.height, &rgb, &yuv));
            ring_.push_back({rgb, yuv});
        }
    }
    size_t ringSize() const { return ring_.size(); }
    void bindSlot(size_t slot)
    {
        std::lock_guard<std::mutex> lk(bindMu_);
        bindSlotLocked(slot);
    }
    void renderTaskAsyncInto(size_t slot, RenderTask &task, pt_stream stream)
    {
        std::lock_guard<std::mutex> lk(bindMu_);
        bindSlotLocked(slot);
        renderTaskAsync(task, stream);
    }
    void gatherTaskFrom(size_t slot, const RenderTask &task, pt_stream stream)
    {
        std::lock_guard<std::mutex> lk(bindMu_);
        bindSlotLocked(slot);
        gatherTask(task, stream);
    }

    // camera parameters are read from the borrowed CameraConfig at every launch (by value, :210)
    void reloadCamera() {}

    // to be called when the scene changes (:312-340): re-upload + rebuild BVH and light list
    void reloadWorld()
    {
        std::vector<pt_material> mats(hostScene_.materials.size());
        for (size_t i = 0; i < mats.size(); i++) {
            const HostMaterial &m = hostScene_.materials[i];
            mats[i] = pt_material{{m.baseColor.x, m.baseColor.y, m.baseColor.z}, m.baseColorTextureIdx.value_or(-1),
                                  {m.emissiveFactor.x, m.emissiveFactor.y, m.emissiveFactor.z}, m.emissiveTextureIdx.value_or(-1)};
        }
        std::vector<pt_texture> texs(hostScene_.textures.size());
        for (size_t i = 0; i < texs.size(); i++) {
            const HostTexture &t = hostScene_.textures[i];
            texs[i] = pt_texture{t.width, t.height, reinterpret_cast<const float *>(t.data.data())};
        }
        // Triangle keeps the reference's 80-byte layout (8-byte-aligned float2): fields are read at their own offsets
        const pt_triangle_layout layout = triangleLayout();
        checkPtErrors(pt_upload_scene_strided(dev_, hostScene_.triangles.data(), hostScene_.triangles.size(), &layout,
                                              mats.data(), mats.size(), texs.data(), texs.size()));
    }

    void setFramebuffer(std::shared_ptr<Framebuffer> framebuffer)
    {
        framebuffer_ = framebuffer;
        if (d_rgb_) checkPtErrors(pt_fb_free(dev_, d_rgb_, d_yuv_));
        Resolution r = framebuffer_->getResolution();
        checkPtErrors(pt_fb_alloc(dev_, (int)r.width, (int)r.height, &d_rgb_, &d_yuv_));
        checkPtErrors(pt_bind_framebuffer(dev_, (int)r.width, (int)r.height, d_rgb_, d_yuv_));
        const unsigned frames = ring_.size() > 1 ? (unsigned)ring_.size() - 1 : 1;
        if (ring_.empty()) ring_.assign(1, {d_rgb_, d_yuv_});
        else { ring_[0] = {d_rgb_, d_yuv_}; setFramesInFlight(frames); }      // the ring follows the resolution
    }

    void setSamplesPerPixel(unsigned int samplesPerPixel) { samplesPerPixel_ = samplesPerPixel; }
    void setRecursionDepth(unsigned int recursionDepth) { recursionDepth_ = recursionDepth; }
    void setThreadBlockSize(Dim3 threadBlockSize)
    {
        threadBlockSize_ = threadBlockSize;
        checkPtErrors(pt_set_thread_block_size(dev_, threadBlockSize_.x, threadBlockSize_.y));
    }

    // opt-in SAH tree instead of the reference's topology (include/pt_amd.h: pt_set_bvh_mode)
    void setBvhMode(int mode) { checkPtErrors(pt_set_bvh_mode(dev_, mode)); reloadWorld(); }

    // EXTENSION (include/pt_amd.h pt_set_legacy_scene): legacy material kinds per material and
    // spheres; empty vectors switch it off.  Takes effect through reloadWorld().
    void setLegacyScene(const std::vector<pt_legacy_material> &materials, const std::vector<pt_sphere> &spheres)
    {
        checkPtErrors(pt_set_legacy_scene(dev_, materials.empty() ? nullptr : materials.data(), materials.size(),
                                          spheres.empty() ? nullptr : spheres.data(), spheres.size()));
        reloadWorld();
    }

    // four waves per tile sharing every leaf's triangles: -1 automatic (default), 0 off, 1 on
    void setCooperativeMode(int mode) { checkPtErrors(pt_set_cooperative_mode(dev_, mode)); }

    // tiles in raster order (0) or ordered by their cost in the previous launch of the same task (1, default)
    void setTileSchedule(int mode) { checkPtErrors(pt_set_tile_schedule(dev_, mode)); }

    pt_stream createStream() { pt_stream s = nullptr; checkPtErrors(pt_stream_create(dev_, &s)); return s; }
    void destroyStream(pt_stream s) { checkPtErrors(pt_stream_destroy(dev_, s)); }
    pt_counters readCounters() { pt_counters c{}; checkPtErrors(pt_counters_read(dev_, &c)); return c; }
    void resetCounters() { checkPtErrors(pt_counters_reset(dev_)); }
    pt_device *handle() { return dev_; }
    int deviceIndex() const { return device_idx_; }

    ~DevicePathTracer()
    {
        if (dev_) {
            for (size_t i = 1; i < ring_.size(); i++) pt_fb_free(dev_, ring_[i].first, ring_[i].second);
            if (d_rgb_) pt_fb_free(dev_, d_rgb_, d_yuv_);
            pt_device_destroy(dev_);
        }
    }

private:
    void bindSlotLocked(size_t slot)
    {
        const auto &b = ring_[slot % ring_.size()];
        Resolution r = framebuffer_->getResolution();
        checkPtErrors(pt_bind_framebuffer(dev_, (int)r.width, (int)r.height, b.first, b.second));
    }
    pt_camera cameraByValue() const
    {
        const CameraConfig &c = cameraConfig_;
        return pt_camera{{c.lookFrom.x, c.lookFrom.y, c.lookFrom.z}, {c.front.x, c.front.y, c.front.z}, c.vfov, c.hfov};
    }

    int device_idx_;
    pt_device *dev_ = nullptr;
    uint8_t *d_rgb_ = nullptr, *d_yuv_ = nullptr;
    std::vector<std::pair<uint8_t *, uint8_t *>> ring_{};     // device framebuffers: [0] = (d_rgb_, d_yuv_), + framesInFlight more when > 1
    std::mutex bindMu_{};
    Dim3 threadBlockSize_;
    HostScene &hostScene_;
    unsigned int samplesPerPixel_;
    unsigned int recursionDepth_;
    std::shared_ptr<Framebuffer> framebuffer_;
    CameraConfig &cameraConfig_;
};

} // namespace ptamd
