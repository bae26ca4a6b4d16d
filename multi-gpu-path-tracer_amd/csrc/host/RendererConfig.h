// RendererConfig.h -- configuration PODs of the render path, same names and defaults as the
// reference's RendererConfig.h:6-37 / CameraConfig.h:5-17, without CUDA vector types.
#pragma once

#include <string>

namespace ptamd {

struct Float3 { float x = 0.0f, y = 0.0f, z = 0.0f; };
struct Dim3 { unsigned x = 1, y = 1, z = 1; };            // stands in for CUDA's dim3

struct Resolution {
    unsigned int width;
    unsigned int height;
};

enum SchedulingAlgorithmType {
    FSFL,   // fixed-size tasks, fixed layout
    DSFL,   // task sizes adapt frame to frame, layout fixed
    DSDL,   // layout adapts too (recursive bisection; power-of-two GPU counts)
    // Not in the reference (its three policies all fix the frame's split BEFORE the frame starts): an intra-frame queue
    // of 64-128 full-width row bands; every (GPU, stream) worker pulls the next band the moment its stream drains
    // (BASELINE.json configs[4] "dynamic tile load-balancer").  Balances the FIRST frame after any scene /
    // camera / resolution change, which frame-to-frame feedback cannot.
    DQRB
};

struct RendererConfig {
    std::string jobId = "0";
    unsigned int samplesPerPixel = 10;
    unsigned int recursionDepth = 3;
    std::string modelPath{};
    unsigned int gpuNumber = 1;
    unsigned int streamsPerGpu = 1;
    Resolution resolution{400, 400};
    SchedulingAlgorithmType algorithmType = FSFL;
    Dim3 threadBlockSize{8, 8, 1};
    float vfov = 45.0f;
    float hfov = 45.0f;
    Float3 cameraLookFromVec{0.0f, 0.0f, 0.0f};
    Float3 cameraFrontVec{1.0f, 0.0f, 0.0f};
    unsigned int maxTasksInRow = 2;
    bool showTasks = true;
    int kParam = 1;
    // Not in the reference (its renderFrame ends on a barrier, RenderManager.h:410-431): frames rendered concurrently per
    // GPU, each in its own device framebuffer on its own stream.  With F > 1 renderFrame() launches frame i and hands out
    // frame i - F + 1 (RenderManager::flush() completes the rest); FSFL only -- the adaptive policies time a task alone.
    unsigned int framesInFlight = 1;
    // Rehearsal on a box with ONE GPU: engine i still gets its own pt_device, streams, framebuffers and share of the frame,
    // but all of them on device 0 (the gather then runs its peer-copy transport within the device).  Checks the N-engine
    // path, not its speed.
    bool allOnDevice0 = false;
};

struct CameraConfig {
    CameraConfig(Float3 lookFrom_, Float3 front_, float vfov_ = 45.0f, float hfov_ = 45.0f)
        : front{front_}, lookFrom{lookFrom_}, vfov{vfov_}, hfov{hfov_} {}
    Float3 front;
    Float3 lookFrom;
    float vfov = 45.0f;
    float hfov = 45.0f;
    float pitch = 0.0f;     // UI state, not read by the render path
    float yaw = 0.0f;
};

} // namespace ptamd
