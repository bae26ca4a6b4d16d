// RenderManager.h -- frame orchestration for one process driving N GPUs with one worker
// thread per (GPU, stream), public surface of the reference's RenderManager
// (RenderManager.h:27-659): deferred parameter hot-swap, FSFL / DSFL / DSDL task policies,
// task-border overlay, and the two-barrier frame protocol (:410-431).
//
// Differences by design: the scene is uploaded once per GPU in bulk; there is no managed memory -- with ONE GPU every
// worker copies its finished rectangle into the shared host Framebuffer itself, with SEVERAL the frame is gathered
// natively after the frame barrier (pt_gather_frame: every peer GPU packs its rectangles, one grouped RCCL
// ncclSend / ncclRecv exchange brings them to GPU 0 over xGMI, one device-to-host copy of the whole frame);
// every task is rendered exactly once also with streamsPerGpu > 1.  The image does not
// depend on the task partition (per-pixel seeds), only the timing does.
#pragma once

#include <algorithm>
#include <atomic>
#include <cmath>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "DevicePathTracer.h"
#include "Framebuffer.h"
#include "HostScene.h"
#include "RendererConfig.h"
#include "StreamThread.h"
#include "TaskGenerator.h"
#include "barrier.h"

namespace ptamd {

class RenderManager {
public:
    RenderManager(RendererConfig &config, HostScene &hScene, CameraConfig &cameraConfig, SceneLoader &sceneLoader)
        : hScene_{hScene}, config_{config}, cameraConfig_{cameraConfig},
          barrier_{config.gpuNumber * config.streamsPerGpu + 1}, sceneLoader_{sceneLoader}
    {
        newConfig_ = config_;
        setup();
    }

    ~RenderManager() { reset(); }

    // rows of at most maxTasksInRow task ids (RenderManager.h:42-59)
    std::vector<std::vector<int>> getTaskLayout(unsigned int maxTasksInRow)
    {
        std::vector<std::vector<int>> layout;
        const int n = (int)(config_.gpuNumber * config_.streamsPerGpu);
        for (int task = 0; task < n;) {
            layout.emplace_back();
            for (unsigned k = 0; k < maxTasksInRow && task < n; k++) layout.back().push_back(task++);
        }
        return layout;
    }

    void reset()
    {
        if (streamThreads_.empty()) return;
        for (auto &perGpu : streamThreads_) for (auto &t : perGpu) t->finish();
        barrier_.wait();                                   // release the workers so they can exit
        for (auto &perGpu : streamThreads_) for (auto &t : perGpu) t->join();
        streamThreads_.clear();
        if (gather_) { pt_gather_destroy(gather_); gather_ = nullptr; }
        devicePathTracers_.clear();
    }

    void setup()
    {
        if (!framebuffer_) framebuffer_ = std::make_shared<Framebuffer>(config_.resolution);
        threadCount_ = (int)(config_.gpuNumber * config_.streamsPerGpu);
        barrier_.reset((size_t)threadCount_ + 1);
        taskLayout_ = getTaskLayout(config_.maxTasksInRow);
        renderTasks_ = taskGen_.generateEqualTasks(threadCount_, taskLayout_, (int)config_.resolution.width,
                                                   (int)config_.resolution.height);
        bisectionDepth_ = (int)std::log2((double)threadCount_);
        bandQueue_.build((int)config_.resolution.width, (int)config_.resolution.height);
        queueActive_.store(config_.algorithmType == DQRB);
        // frames in flight: FSFL only (the adaptive policies time a task that has the device to itself)
        fif_ = config_.algorithmType == FSFL ? std::max(1u, config_.framesInFlight) : 1u;
        // launches only overlap while each has a hardware queue of its own (include/pt_amd.h pt_hw_queue_count): more frames
        // in flight than queues are slower than fewer, so the pipeline is no deeper than the queues the process has
        if (fif_ > (unsigned)std::max(1, pt_hw_queue_count())) {
            std::cerr << "warning: framesInFlight " << fif_ << " capped at the " << pt_hw_queue_count()
                      << " hardware queues of this process (GPU_MAX_HW_QUEUES)" << std::endl;
            fif_ = (unsigned)std::max(1, pt_hw_queue_count());
        }
        frameIndex_.store(0); firstValid_.store(0); draining_.store(false); inflight_ = 0;
        for (unsigned i = 0; i < config_.gpuNumber; i++) {
            devicePathTracers_.push_back(std::make_shared<DevicePathTracer>(
                config_.allOnDevice0 ? 0 : (int)i, config_.samplesPerPixel, config_.recursionDepth, config_.threadBlockSize, hScene_, framebuffer_,
                cameraConfig_));
            if (fif_ > 1) devicePathTracers_.back()->setFramesInFlight(fif_);
            // DQRB launches 64-128 bands per frame, each its own task geometry: a cost-ordered tile schedule per band would be
            // made, sorted and evicted every launch (the library caches 16) and never reused -- and a band has fewer tiles
            // than the device has wave slots, so their order does not matter.  Bands render in raster order.
            if (config_.algorithmType == DQRB) devicePathTracers_.back()->setTileSchedule(0);
        }
        // The gather group is made BEFORE any worker thread exists: nothing is parked on the barrier if it fails.  It
        // only fails on a HIP error: without a usable librccl (or when ncclCommInitAll refuses) the group moves the same
        // strings with peer copies (pt_gather_transport says which).  If even that cannot be set up, every worker copies
        // its own rectangle to the host, as single-GPU runs do.
        bool gatherOnWorker = config_.gpuNumber == 1;
        if (config_.gpuNumber > 1) {
            std::vector<pt_device *> devs;
            for (auto &dpt : devicePathTracers_) devs.push_back(dpt->handle());
            if (pt_gather_create(devs.data(), (int)devs.size(), &gather_) != PT_OK) {
                std::cerr << "warning: native gather unavailable (" << pt_last_error_string()
                          << "): every worker copies its rectangle to the host" << std::endl;
                gather_ = nullptr;
                gatherOnWorker = true;
            } else {
                char how[256] = "";
                pt_gather_transport(gather_, how, sizeof how);
                gatherTransport_ = how;
                if (gatherTransport_.rfind("rccl", 0) != 0) std::cerr << "warning: frame gather uses " << gatherTransport_ << std::endl;
            }
        }
        for (unsigned i = 0; i < config_.gpuNumber; i++) {
            streamThreads_.emplace_back();
            for (unsigned j = 0; j < config_.streamsPerGpu; j++) {
                int taskIdx = (int)(i * config_.streamsPerGpu + j);
                StreamThread::Pipeline pipe;
                if (fif_ > 1) { pipe.frames = fif_; pipe.frameIndex = &frameIndex_; pipe.firstValid = &firstValid_; pipe.draining = &draining_; }
                streamThreads_[i].push_back(std::make_shared<StreamThread>((int)i, taskIdx, devicePathTracers_[i], renderTasks_, barrier_,
                                                                          gatherOnWorker, &bandQueue_, &queueActive_, pipe));
                streamThreads_[i][j]->start();
            }
        }
    }

    // The frame's one exchange step with several GPUs: task k was rendered by GPU k / streamsPerGpu (setup()).
    void gatherFrame(long slot = -1)
    {
        if (!gather_) return;
        // frames in flight: the exchange moves the COMPLETED frame, which sits in its own slot of every engine's ring (the
        // workers are parked between frames, nobody else binds meanwhile)
        if (slot >= 0) for (auto &dpt : devicePathTracers_) dpt->bindSlot((size_t)slot);
        std::vector<pt_task> tasks;
        std::vector<int> owner;
        if (queueActive_.load()) {                         // DQRB: band b was rendered by whoever pulled it this frame
            for (size_t b = 0; b < bandQueue_.bands.size(); b++) {
                const RenderTask &t = bandQueue_.bands[b];
                tasks.push_back(pt_task{t.width, t.height, t.offset_x, t.offset_y});
                owner.push_back(bandQueue_.owner[b]);
            }
        } else
        for (size_t k = 0; k < renderTasks_.size(); k++) {
            const RenderTask &t = renderTasks_[k];
            tasks.push_back(pt_task{t.width, t.height, t.offset_x, t.offset_y});
            owner.push_back((int)(k / config_.streamsPerGpu));
        }
        checkPtErrors(pt_gather_frame(gather_, tasks.data(), owner.data(), (int)tasks.size(), framebuffer_->getRGBPtr(),
                                      framebuffer_->getYUVPtr()));
    }

    // ---- deferred setters (RenderManager.h:114-248): applied at the top of the next frame
    void setKParameter(int val) { newConfig_.kParam = val; dirty_ = true; }
    void setGpuNumber(int gpuNumber)
    {
        if (config_.algorithmType == DSDL) {               // bisection needs a power of two (:188-195)
            int p = 1;
            while (p * 2 <= gpuNumber) p *= 2;
            gpuNumber = p;
        }
        newConfig_.gpuNumber = (unsigned)gpuNumber;
        dirty_ = true;
    }
    void setStreamsPerGpu(int streamsPerGpu) { newConfig_.streamsPerGpu = (unsigned)streamsPerGpu; dirty_ = true; }
    void setGpuAndStreamNumber(int gpuNumber, int streamsPerGpu) { setGpuNumber(gpuNumber); setStreamsPerGpu(streamsPerGpu); }
    void setResolution(Resolution res) { newConfig_.resolution = res; dirty_ = true; }
    void setSamplesPerPixel(unsigned int samples) { newConfig_.samplesPerPixel = samples; dirty_ = true; }
    void setRecursionDepth(unsigned int depth) { newConfig_.recursionDepth = depth; dirty_ = true; }
    void setThreadBlockSize(Dim3 threadBlockSize) { newConfig_.threadBlockSize = threadBlockSize; dirty_ = true; }
    void setSchedulingAlgorithm(SchedulingAlgorithmType alg) { newConfig_.algorithmType = alg; dirty_ = true; }
    void setShowTasks(bool val) { newConfig_.showTasks = val; dirty_ = true; }

    void setFramesInFlight(unsigned frames) { newConfig_.framesInFlight = std::max(1u, frames); dirty_ = true; }

    void updatePathTracingParamsIfNeeded()
    {
        if (!dirty_) return;
        dirty_ = false;
        flush();                                       // parameters change between frames: nothing stays in flight across them
        const unsigned wantFif = newConfig_.algorithmType == FSFL ? std::max(1u, newConfig_.framesInFlight) : 1u;
        if (wantFif != fif_) {                         // the workers' streams and the engines' rings follow the pipeline depth
            reset();
            config_ = newConfig_;
            framebuffer_->setResolution(config_.resolution);
            setup();
            return;
        }
        config_.framesInFlight = newConfig_.framesInFlight;
        config_.algorithmType = newConfig_.algorithmType;
        config_.showTasks = newConfig_.showTasks;
        config_.kParam = newConfig_.kParam;
        const bool resChanged = config_.resolution.width != newConfig_.resolution.width ||
                                config_.resolution.height != newConfig_.resolution.height;
        if (resChanged) {
            config_.resolution = newConfig_.resolution;
            framebuffer_->setResolution(config_.resolution);
        }
        if (config_.gpuNumber != newConfig_.gpuNumber || config_.streamsPerGpu != newConfig_.streamsPerGpu) {
            reset();
            config_.gpuNumber = newConfig_.gpuNumber;
            config_.streamsPerGpu = newConfig_.streamsPerGpu;
            config_.samplesPerPixel = newConfig_.samplesPerPixel;
            config_.recursionDepth = newConfig_.recursionDepth;
            config_.threadBlockSize = newConfig_.threadBlockSize;
            setup();
            return;
        }
        if (queueActive_.load() != (config_.algorithmType == DQRB))      // (see setup(): no per-band tile schedules under DQRB)
            for (auto &dpt : devicePathTracers_) dpt->setTileSchedule(config_.algorithmType == DQRB ? 0 : 1);
        queueActive_.store(config_.algorithmType == DQRB);       // workers read it after the frame's first barrier
        if (resChanged) {
            for (auto &dpt : devicePathTracers_) dpt->setFramebuffer(framebuffer_);
            bandQueue_.build((int)config_.resolution.width, (int)config_.resolution.height);
            renderTasks_ = taskGen_.generateEqualTasks(threadCount_, taskLayout_, (int)config_.resolution.width,
                                                       (int)config_.resolution.height);
        }
        if (config_.samplesPerPixel != newConfig_.samplesPerPixel) {
            config_.samplesPerPixel = newConfig_.samplesPerPixel;
            for (auto &dpt : devicePathTracers_) dpt->setSamplesPerPixel(config_.samplesPerPixel);
        }
        if (config_.recursionDepth != newConfig_.recursionDepth) {
            config_.recursionDepth = newConfig_.recursionDepth;
            for (auto &dpt : devicePathTracers_) dpt->setRecursionDepth(config_.recursionDepth);
        }
        if (config_.threadBlockSize.x != newConfig_.threadBlockSize.x || config_.threadBlockSize.y != newConfig_.threadBlockSize.y) {
            config_.threadBlockSize = newConfig_.threadBlockSize;
            for (auto &dpt : devicePathTracers_) dpt->setThreadBlockSize(config_.threadBlockSize);
        }
    }

    void reloadWorldIfNeeded()
    {
        if (!shouldReloadWorld_) return;
        shouldReloadWorld_ = false;
        for (auto &dpt : devicePathTracers_) dpt->reloadWorld();
    }

    // ---- DSFL (RenderManager.h:334-408, :546-639): layout fixed, each divider moves toward the
    // equal-time point of its row by at most one block per frame; row heights likewise.
    void adjustTasksDSFL()
    {
        const int W = (int)config_.resolution.width, H = (int)config_.resolution.height;
        const int bx = (int)config_.threadBlockSize.x, by = (int)config_.threadBlockSize.y;
        for (auto &row : taskLayout_) {
            std::vector<int> times, lens;
            for (int id : row) { times.push_back(renderTasks_[(size_t)id].time); lens.push_back(renderTasks_[(size_t)id].width); }
            std::vector<int> target = equalTimeDividers(times, lens, bx, W);
            int x = 0;
            for (size_t k = 0; k + 1 < row.size(); k++) {
                RenderTask &t = renderTasks_[(size_t)row[k]];
                t.offset_x = x;
                int cur = t.offset_x + t.width;
                int want = std::clamp(target[k], cur - bx, cur + bx);
                int remaining = (int)(row.size() - 1 - k);
                want = std::clamp(want, x + 1, W - remaining);
                t.width = want - x;
                x = want;
            }
            RenderTask &last = renderTasks_[(size_t)row.back()];
            last.offset_x = x;
            last.width = W - x;
        }
        std::vector<int> rowTimes, rowLens;
        for (auto &row : taskLayout_) {
            int sum = 0;
            for (int id : row) sum += renderTasks_[(size_t)id].time;
            rowTimes.push_back(sum);
            rowLens.push_back(renderTasks_[(size_t)row[0]].height);
        }
        std::vector<int> target = equalTimeDividers(rowTimes, rowLens, by, H);
        int y = 0;
        for (size_t r = 0; r < taskLayout_.size(); r++) {
            int h;
            if (r + 1 < taskLayout_.size()) {
                int cur = y + renderTasks_[(size_t)taskLayout_[r][0]].height;
                int want = std::clamp(target[r], cur - by, cur + by);
                int remaining = (int)(taskLayout_.size() - 1 - r);
                want = std::clamp(want, y + 1, H - remaining);
                h = want - y;
            } else {
                h = H - y;
            }
            for (int id : taskLayout_[r]) { renderTasks_[(size_t)id].offset_y = y; renderTasks_[(size_t)id].height = h; }
            y += h;
        }
    }

    // ---- DSDL (RenderManager.h:264-331): spread last frame's task times over a block grid and
    // cut it by recursive equal-time bisection, alternating the cut direction.
    void adjustTasksDSDL()
    {
        const int bx = (int)config_.threadBlockSize.x, by = (int)config_.threadBlockSize.y;
        const int W = (int)config_.resolution.width, H = (int)config_.resolution.height;
        const int gw = (W + bx - 1) / bx, gh = (H + by - 1) / by;
        if ((1 << bisectionDepth_) != threadCount_) return;             // power-of-two thread counts only
        std::vector<float> grid((size_t)gw * gh, 0.0f);
        for (const RenderTask &t : renderTasks_) {
            int x0 = t.offset_x / bx, x1 = std::max(x0 + 1, (t.offset_x + t.width + bx - 1) / bx);
            int y0 = t.offset_y / by, y1 = std::max(y0 + 1, (t.offset_y + t.height + by - 1) / by);
            float per = (float)std::max(t.time, 1) / (float)((x1 - x0) * (y1 - y0));
            for (int y = y0; y < std::min(y1, gh); y++)
                for (int x = x0; x < std::min(x1, gw); x++) grid[(size_t)y * gw + x] = per;
        }
        std::vector<RenderTask> out;
        bisect(grid, gw, 0, 0, gw, gh, 0, true, out);
        for (RenderTask &t : out) {                                      // block units -> pixels, clipped to the frame
            t.offset_x *= bx; t.offset_y *= by; t.width *= bx; t.height *= by;
            t.width = std::min(t.width, W - t.offset_x);
            t.height = std::min(t.height, H - t.offset_y);
        }
        if ((int)out.size() == threadCount_) {
            for (size_t i = 0; i < out.size(); i++) { out[i].time = renderTasks_[i].time; renderTasks_[i] = out[i]; }
        }
    }

    // RenderManager::renderFrame, RenderManager.h:410-431
    void renderFrame()
    {
        updatePathTracingParamsIfNeeded();
        reloadWorldIfNeeded();
        if (config_.algorithmType == DSFL) adjustTasksDSFL();
        else if (config_.algorithmType == DSDL) adjustTasksDSDL();
        else if (config_.algorithmType == DQRB) bandQueue_.beginFrame();
        if (fif_ > 1) { pipelineRound(false); return; }
        barrier_.wait();     // release the workers
        barrier_.wait();     // all tasks rendered (and, with one GPU, gathered by their workers)
        if (config_.algorithmType == DQRB && !bandQueue_.everyBandExactlyOnce())
            throw std::runtime_error("DQRB: a band of the frame was not rendered exactly once");
        gatherFrame();       // several GPUs: one RCCL exchange to GPU 0 + one copy to the host
        if (config_.showTasks) markTasks();
    }

    // Frames in flight: complete every launched frame, oldest first; afterwards the Framebuffer holds the last frame launched.
    void flush()
    {
        while (fif_ > 1 && inflight_ > 0) pipelineRound(true);
        if (fif_ > 1) firstValid_.store(frameIndex_.load());
    }
    unsigned framesInFlight() const { return fif_; }
    // true when the Framebuffer holds a frame completed by the latest renderFrame() / flush()
    bool frameCompleted() const { return completedLast_; }

    // per-task milliseconds and imbalance = max / mean (RenderManager.h:433-447)
    float imbalance() const
    {
        float sum = 0.0f, mx = 0.0f;
        if (queueActive_.load()) {                         // DQRB: what a worker is busy = the bands it pulled
            std::vector<float> busy(config_.gpuNumber, 0.0f);
            for (size_t b = 0; b < bandQueue_.bands.size(); b++)
                if (bandQueue_.owner[b] >= 0) busy[(size_t)bandQueue_.owner[b]] += (float)bandQueue_.bands[b].time;
            for (float v : busy) { sum += v; mx = std::max(mx, v); }
            return sum > 0.0f ? mx / (sum / (float)busy.size()) : 1.0f;
        }
        for (const RenderTask &t : renderTasks_) { sum += (float)t.time; mx = std::max(mx, (float)t.time); }
        return sum > 0.0f ? mx / (sum / (float)renderTasks_.size()) : 1.0f;
    }
    const std::vector<RenderTask> &tasks() const { return queueActive_.load() ? bandQueue_.bands : renderTasks_; }
    const BandQueue &bandQueue() const { return bandQueue_; }
    const std::string &gatherTransport() const { return gatherTransport_; }

    // black task borders drawn into the frame (RenderManager.h:449-507)
    void markTasks()
    {
        const int W = (int)framebuffer_->getResolution().width, H = (int)framebuffer_->getResolution().height;
        const int bold = H / 300;
        auto hline = [&](int x0, int x1, int y) {
            for (int yy = y; yy <= y + bold && yy < H; yy++)
                for (int x = std::max(x0, 0); x < std::min(x1, W); x++) framebuffer_->updatePixel(yy * W + x, 0, 0, 0);
        };
        auto vline = [&](int y0, int y1, int x) {
            for (int xx = x; xx <= x + bold && xx < W; xx++)
                for (int y = std::max(y0, 0); y < std::min(y1, H); y++) framebuffer_->updatePixel(y * W + xx, 0, 0, 0);
        };
        for (const RenderTask &t : tasks()) {
            if (t.offset_y != 0) hline(t.offset_x, t.offset_x + t.width, t.offset_y);
            hline(t.offset_x, t.offset_x + t.width, t.offset_y + t.height);
            if (t.offset_x != 0) vline(t.offset_y, t.offset_y + t.height, t.offset_x);
            vline(t.offset_y, t.offset_y + t.height, t.offset_x + t.width);
        }
    }

    std::shared_ptr<Framebuffer> &getFramebuffer() { return framebuffer_; }
    uint8_t *getCurrentFrame() { return framebuffer_->getRGBPtr(); }
    uint8_t *getYUVFrame() { return framebuffer_->getYUVPtr(); }
    unsigned int getCurrentFrameWidth() { return framebuffer_->getResolution().width; }
    unsigned int getCurrentFrameHeight() { return framebuffer_->getResolution().height; }

    void reloadScene()                                                    // RenderManager.h:534-539
    {
        std::string path = "../files/f" + config_.jobId + ".glb";
        hScene_ = sceneLoader_.load(path);
        shouldReloadWorld_ = true;
    }
    void updatePrimitives() { shouldReloadWorld_ = true; }
    std::vector<std::shared_ptr<DevicePathTracer>> &tracers() { return devicePathTracers_; }

private:
    // One round of the frames-in-flight protocol: the workers launch frame i (unless draining) and complete frame
    // i - F + 1 (StreamThread::threadMain); with several GPUs the completed frame is then exchanged out of its ring slot.
    void pipelineRound(bool draining)
    {
        const long F = (long)fif_, i = frameIndex_.load(), j = i - (F - 1);
        draining_.store(draining);
        barrier_.wait();
        barrier_.wait();
        if (!draining) inflight_++;
        completedLast_ = j >= 0 && j >= firstValid_.load();
        if (completedLast_) {
            inflight_--;
            gatherFrame(j % (F + 1));
            if (config_.showTasks) markTasks();
        }
        draining_.store(false);
        frameIndex_.store(i + 1);
    }

    // Equal-time division points of a row of tasks (in pixels, multiples of `block`), assuming
    // each task's time is spread evenly over its blocks (RenderManager.h:546-603).
    static std::vector<int> equalTimeDividers(const std::vector<int> &times, const std::vector<int> &lens, int block, int total)
    {
        std::vector<int> out;
        if (times.size() <= 1) return out;
        double sum = 0.0;
        for (int t : times) sum += std::max(t, 0);
        const double target = sum / (double)times.size();
        double acc = 0.0;
        int pos = 0;
        const size_t need = times.size() - 1;
        for (size_t i = 0; i < times.size() && out.size() < need; i++) {
            int blocks = std::max(1, lens[i] / block);
            double perBlock = (double)std::max(times[i], 0) / (double)blocks;
            for (int k = 0; k < blocks && out.size() < need; k++) {
                if (acc + perBlock > target && pos > (out.empty() ? 0 : out.back())) { out.push_back(pos); acc = 0.0; }
                acc += perBlock;
                pos += block;
            }
        }
        while (out.size() < need) { pos = std::min(pos + block, total - 1); out.push_back(pos); }
        return out;
    }

    void bisect(const std::vector<float> &grid, int gw, int x, int y, int w, int h, int level, bool horizontalCut,
                std::vector<RenderTask> &out)
    {
        if (level == bisectionDepth_) { out.push_back(RenderTask{w, h, x, y}); return; }
        double total = 0.0;
        for (int yy = y; yy < y + h; yy++) for (int xx = x; xx < x + w; xx++) total += grid[(size_t)yy * gw + xx];
        const double half = total / 2.0;
        double acc = 0.0;
        if (h > 1 && (horizontalCut || w <= 1)) {          // the asked-for direction, or the only one a 1-wide region has
            int cut = 1;
            for (int yy = y; yy < y + h - 1; yy++) {
                for (int xx = x; xx < x + w; xx++) acc += grid[(size_t)yy * gw + xx];
                cut = yy - y + 1;
                if (acc >= half) break;
            }
            bisect(grid, gw, x, y, w, cut, level + 1, false, out);
            bisect(grid, gw, x, y + cut, w, h - cut, level + 1, false, out);
        } else if (w > 1) {
            int cut = 1;
            for (int xx = x; xx < x + w - 1; xx++) {
                for (int yy = y; yy < y + h; yy++) acc += grid[(size_t)yy * gw + xx];
                cut = xx - x + 1;
                if (acc >= half) break;
            }
            bisect(grid, gw, x, y, cut, h, level + 1, true, out);
            bisect(grid, gw, x + cut, y, w - cut, h, level + 1, true, out);
        } else {
            // a 1 x 1 region cannot be cut: it stays whole and the other half of this subtree is empty tasks, so that the
            // subtree still yields its 2^(depth - level) tasks and the layout keeps threadCount_ entries
            out.push_back(RenderTask{w, h, x, y});
            for (int k = 1; k < (1 << (bisectionDepth_ - level)); k++) out.push_back(RenderTask{0, 0, x, y});
        }
    }

    std::vector<std::shared_ptr<DevicePathTracer>> devicePathTracers_{};
    TaskGenerator taskGen_{};
    std::shared_ptr<Framebuffer> framebuffer_;
    HostScene &hScene_;
    std::vector<RenderTask> renderTasks_{};
    std::vector<std::vector<std::shared_ptr<StreamThread>>> streamThreads_{};
    RendererConfig &config_;
    RendererConfig newConfig_{};
    bool dirty_ = false;
    bool shouldReloadWorld_ = false;
    CameraConfig &cameraConfig_;
    Barrier barrier_;
    std::vector<std::vector<int>> taskLayout_;
    int threadCount_ = 1;
    SceneLoader &sceneLoader_;
    int bisectionDepth_ = 0;
    pt_gather *gather_ = nullptr;
    std::string gatherTransport_{};
    BandQueue bandQueue_{};
    std::atomic_bool queueActive_{false};
    // frames in flight (RendererConfig::framesInFlight, FSFL only)
    unsigned fif_ = 1;
    std::atomic<long> frameIndex_{0};       // the round the workers run next
    std::atomic<long> firstValid_{0};       // rounds before it were flushed: nothing of theirs is left to complete
    std::atomic_bool draining_{false};
    long inflight_ = 0;                     // frames launched and not yet completed
    bool completedLast_ = true;
};

} // namespace ptamd
