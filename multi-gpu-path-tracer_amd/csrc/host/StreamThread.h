// StreamThread.h -- one worker per (GPU, stream), as StreamThread.h:16-115: between the two
// frame barriers it launches its task on its stream, waits for it, records the wall time in
// RenderTask::time and (new, single-GPU runs) copies the task's pixels into the shared host framebuffer; with several
// GPUs the RenderManager gathers all rectangles over RCCL after the frame barrier instead.
// Unlike the reference (tasks_[deviceIdx], StreamThread.h:83) every worker owns task number
// `taskIdx`, so streamsPerGpu > 1 renders every task exactly once.
#pragma once

#include <atomic>
#include <chrono>
#include <memory>
#include <thread>
#include <vector>

#include "BandQueue.h"
#include "DevicePathTracer.h"
#include "barrier.h"

namespace ptamd {

// Frames in flight (RendererConfig::framesInFlight > 1): `frames` of them per GPU, the round counter, the first round that
// still holds a frame and the "only complete, launch nothing" flag of RenderManager::flush(), all owned by the
// RenderManager (null pointers: one frame at a time).
struct StreamPipeline {
    unsigned frames = 1;
    const std::atomic<long> *frameIndex = nullptr;
    const std::atomic<long> *firstValid = nullptr;     // rounds before this one hold no frame to complete (everything was flushed)
    const std::atomic_bool *draining = nullptr;
};

class StreamThread {
public:
    // gatherOnWorker: this worker copies its finished rectangle to the host itself (single GPU); otherwise the
    // RenderManager gathers all devices' rectangles in one RCCL exchange after the frame barrier.
    // queue: the DQRB policy's band queue (nullptr or inactive = the reference's one pre-assigned task per frame).
    typedef StreamPipeline Pipeline;

    StreamThread(int device_idx, int task_idx, std::shared_ptr<DevicePathTracer> devicePathTracer,
                 std::vector<RenderTask> &tasks, Barrier &barrier, bool gatherOnWorker = true, BandQueue *queue = nullptr,
                 const std::atomic_bool *queueActive = nullptr, Pipeline pipeline = Pipeline())
        : deviceIdx{device_idx}, taskIdx{task_idx}, devicePathTracer{devicePathTracer}, tasks_{tasks}, barrier_{barrier},
          gatherOnWorker_{gatherOnWorker}, queue_{queue}, queueActive_{queueActive}, pipeline_{pipeline}
    {
        stream = devicePathTracer->createStream();
        for (unsigned k = 1; k < pipeline_.frames; k++) moreStreams_.push_back(devicePathTracer->createStream());
        launchedAt_.resize(pipeline_.frames);
    }

    ~StreamThread()
    {
        shouldTerminate = true;
        if (thread_.joinable()) thread_.join();
        devicePathTracer->destroyStream(stream);
        for (pt_stream s : moreStreams_) devicePathTracer->destroyStream(s);
    }

    void start() { thread_ = std::thread(&StreamThread::threadMain, this); }
    void finish() { shouldTerminate = true; }
    void join() { if (thread_.joinable()) thread_.join(); }

    void threadMain()
    {
        while (true) {
            barrier_.wait();                       // frame start (RenderManager::renderFrame)
            if (shouldTerminate) return;
            if (pipeline_.frames > 1 && pipeline_.frameIndex && (size_t)taskIdx < tasks_.size()) {
                // Frames in flight: launch frame i into ring slot i % (F + 1) on stream i % F, then complete frame i - F + 1
                // (wait for its stream, copy its rectangle out of ITS slot).  While flushing nothing is launched.
                const long F = (long)pipeline_.frames, i = pipeline_.frameIndex->load();
                const bool draining = pipeline_.draining && pipeline_.draining->load();
                RenderTask &task = tasks_[(size_t)taskIdx];
                if (!draining) {
                    launchedAt_[(size_t)(i % F)] = std::chrono::high_resolution_clock::now();
                    devicePathTracer->renderTaskAsyncInto((size_t)(i % (F + 1)), task, streamOf(i % F));
                }
                const long j = i - (F - 1);
                if (j >= 0 && (!pipeline_.firstValid || j >= pipeline_.firstValid->load())) {
                    devicePathTracer->synchronizeStream(streamOf(j % F));
                    auto stop = std::chrono::high_resolution_clock::now();
                    task.time = (int)std::chrono::duration_cast<std::chrono::milliseconds>(stop - launchedAt_[(size_t)(j % F)]).count();
                    if (gatherOnWorker_) devicePathTracer->gatherTaskFrom((size_t)(j % (F + 1)), task, streamOf(j % F));
                }
            } else if (queue_ && queueActive_ && queueActive_->load()) {
                // DQRB: pull bands until the frame's queue is empty; this worker's stream never idles while work is left
                for (int b = queue_->pull(deviceIdx); b >= 0; b = queue_->pull(deviceIdx)) {
                    RenderTask &band = queue_->bands[(size_t)b];
                    auto start = std::chrono::high_resolution_clock::now();
                    devicePathTracer->renderTaskAsync(band, stream);
                    devicePathTracer->synchronizeStream(stream);
                    auto stop = std::chrono::high_resolution_clock::now();
                    band.time = (int)std::chrono::duration_cast<std::chrono::microseconds>(stop - start).count();   // microseconds: bands are short
                    if (gatherOnWorker_) devicePathTracer->gatherTask(band, stream);
                }
            } else if ((size_t)taskIdx < tasks_.size()) {
                RenderTask &task = tasks_[(size_t)taskIdx];
                auto start = std::chrono::high_resolution_clock::now();
                devicePathTracer->renderTaskAsync(task, stream);
                devicePathTracer->synchronizeStream(stream);
                auto stop = std::chrono::high_resolution_clock::now();
                task.time = (int)std::chrono::duration_cast<std::chrono::milliseconds>(stop - start).count();
                if (gatherOnWorker_) devicePathTracer->gatherTask(task, stream);
            }
            barrier_.wait();                       // frame end
#ifdef PTAMD_TEST_AFTER_FRAME_END
            // test seam (tests/native/stream_thread_teardown_check.cpp): a statement run by the worker right after the frame-end
            // barrier, where the teardown race lived; never defined in a product build
            PTAMD_TEST_AFTER_FRAME_END
#endif
            // (No termination check here: RenderManager::reset() raises the flag and then joins the frame-START barrier with
            // every worker.  A worker that left after the frame-end barrier -- the flag can already be up when it wakes, if
            // the frame was short -- would never arrive there, and the ones that did would wait for it forever.)
        }
    }

private:
    pt_stream streamOf(long k) const { return k == 0 ? stream : moreStreams_[(size_t)k - 1]; }

    int deviceIdx;
    int taskIdx;
    std::atomic_bool shouldTerminate{false};
    std::shared_ptr<DevicePathTracer> devicePathTracer;
    pt_stream stream = nullptr;
    std::thread thread_{};
    std::vector<RenderTask> &tasks_;
    Barrier &barrier_;
    bool gatherOnWorker_ = true;
    BandQueue *queue_ = nullptr;
    const std::atomic_bool *queueActive_ = nullptr;
    Pipeline pipeline_{};
    std::vector<pt_stream> moreStreams_{};                                        // streams 1 .. F - 1 of the pipeline
    std::vector<std::chrono::high_resolution_clock::time_point> launchedAt_{};    // when the frame on stream k was launched
};

} // namespace ptamd
