// main.cpp -- `pt_render`: the .glb -> PPM driver on top of RenderManager, replacing the
// reference's main.cu:33-95 (whose only sinks are a websocket stream or a GL window).
//   pt_render <jobId> <model.glb|.obj> [options]      (argv[1], argv[2] as ArgumentLoader.h:10-13)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

#include "RenderManager.h"

using namespace ptamd;

static bool parse3(const char *s, Float3 &out) { return sscanf(s, "%f,%f,%f", &out.x, &out.y, &out.z) == 3; }

int main(int argc, char **argv)
{
    // (hardware queues for frames in flight: libptamd.so asks for 8 when it is loaded, before main(); pt_hw_queue_count())
    RendererConfig config;
    config.jobId = (argc > 1 && argv[1][0] != '\0') ? argv[1] : "0";
    config.modelPath = (argc > 2 && argv[2][0] != '\0') ? argv[2] : "models/cornell_duck.glb";
    config.showTasks = false;
    CameraConfig cameraConfig(Float3{0, 0, 0.5f}, Float3{0, 0, -0.5f});      // main.cu:40
    std::string out = "out.ppm", yuvOut;
    int frames = 1, warmup = 0;
    bool cornell = false, sah = false, json = false;
    for (int i = 3; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--width") config.resolution.width = (unsigned)atoi(next());
        else if (a == "--height") config.resolution.height = (unsigned)atoi(next());
        else if (a == "--spp") config.samplesPerPixel = (unsigned)atoi(next());
        else if (a == "--depth") config.recursionDepth = (unsigned)atoi(next());
        else if (a == "--gpus") config.gpuNumber = (unsigned)atoi(next());
        else if (a == "--streams") config.streamsPerGpu = (unsigned)atoi(next());
        else if (a == "--max-tasks-in-row") config.maxTasksInRow = (unsigned)atoi(next());
        else if (a == "--frames") frames = atoi(next());
        else if (a == "--frames-in-flight") config.framesInFlight = (unsigned)std::max(1, atoi(next()));
        else if (a == "--warmup") warmup = std::max(0, atoi(next()));
        else if (a == "--json") json = true;                      // bench.py --host native: one summary line of the timed frames
        else if (a == "--same-device") config.allOnDevice0 = true;
        else if (a == "--out") out = next();
        else if (a == "--yuv") yuvOut = next();
        else if (a == "--cornell") cornell = true;
        else if (a == "--bvh") sah = std::string(next()) == "sah";
        else if (a == "--show-tasks") config.showTasks = true;
        else if (a == "--vfov") cameraConfig.vfov = (float)atof(next());
        else if (a == "--hfov") cameraConfig.hfov = (float)atof(next());
        else if (a == "--lookfrom") { if (!parse3(next(), cameraConfig.lookFrom)) { std::cerr << "bad --lookfrom\n"; return 2; } }
        else if (a == "--front") { if (!parse3(next(), cameraConfig.front)) { std::cerr << "bad --front\n"; return 2; } }
        else if (a == "--algo") {
            std::string v = next();
            config.algorithmType = v == "dsfl" ? DSFL : (v == "dsdl" ? DSDL : (v == "dqrb" ? DQRB : FSFL));
        } else { std::cerr << "unknown option " << a << "\n"; return 2; }
    }
    if (cornell) { cameraConfig.lookFrom = Float3{-255.5f, 252.5f, 700.0f}; cameraConfig.front = Float3{0, 0, -1}; }

    SceneLoader sceneLoader;
    HostScene hScene;
    try {
        hScene = sceneLoader.load(config.modelPath, cornell);
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    std::cout << "Number of triangles: " << hScene.triangles.size() << std::endl;

    auto t0 = std::chrono::high_resolution_clock::now();
    RenderManager manager(config, hScene, cameraConfig, sceneLoader);
    auto t1 = std::chrono::high_resolution_clock::now();
    std::cout << "initializing in: " << std::chrono::duration_cast<std::chrono::milliseconds>(t1 - t0).count() << "ms" << std::endl;

    if (sah) for (auto &dpt : manager.tracers()) dpt->setBvhMode(1);
    if (json) {
        // The benchmark leg (bench.py --host native): `warmup` untimed frames, then `frames` timed ones -- launched, rendered,
        // gathered (pt_gather_frame) and in the host Framebuffer, the pipeline empty at both ends -- and ONE line with the
        // engines' own counters.  The reference's loop (main.cu:60-86) times the same call, RenderManager::renderFrame.
        for (int f = 0; f < warmup; f++) manager.renderFrame();
        manager.flush();
        for (auto &dpt : manager.tracers()) { dpt->waitForRenderTask(); dpt->resetCounters(); }
        auto start = std::chrono::high_resolution_clock::now();
        for (int f = 0; f < frames; f++) manager.renderFrame();
        manager.flush();
        auto stop = std::chrono::high_resolution_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(stop - start).count();
        pt_counters tot{};
        for (auto &dpt : manager.tracers()) {
            const pt_counters c = dpt->readCounters();
            tot.paths += c.paths; tot.segments += c.segments; tot.box_tests += c.box_tests; tot.tri_tests += c.tri_tests; tot.light_tests += c.light_tests;
        }
        static const char *algos[] = {"FSFL", "DSFL", "DSDL", "DQRB"};
        printf("{\"frames\": %d, \"warmup\": %d, \"ms_total\": %.3f, \"devices\": %zu, \"workers\": %zu, \"frames_in_flight\": %u, "
               "\"gather_transport\": \"%s\", \"algorithm\": \"%s\", \"paths\": %llu, \"segments\": %llu, \"box_tests\": %llu, "
               "\"tri_tests\": %llu, \"light_tests\": %llu, \"hw_queues\": %d}\n",
               frames, warmup, ms, manager.tracers().size(), (size_t)(config.gpuNumber * config.streamsPerGpu), manager.framesInFlight(),
               manager.gatherTransport().c_str(), algos[(int)config.algorithmType], (unsigned long long)tot.paths, (unsigned long long)tot.segments,
               (unsigned long long)tot.box_tests, (unsigned long long)tot.tri_tests, (unsigned long long)tot.light_tests, pt_hw_queue_count());
        frames = 0;
    } else
    if (manager.framesInFlight() > 1) {
        // Frames in flight: renderFrame() launches frame f and hands out frame f - F + 1; the counters of overlapping frames
        // cannot be told apart, so the run is reported as a whole (pipeline empty at both ends).
        for (auto &dpt : manager.tracers()) dpt->resetCounters();
        auto start = std::chrono::high_resolution_clock::now();
        for (int f = 0; f < frames; f++) manager.renderFrame();
        manager.flush();
        auto stop = std::chrono::high_resolution_clock::now();
        double ms = std::chrono::duration<double, std::milli>(stop - start).count();
        std::cout << "Path Tracing took: " << (long)ms << "ms" << std::endl;
        unsigned long long seg = 0;
        for (auto &dpt : manager.tracers()) seg += dpt->readCounters().segments;
        printf("{\"frames\": %d, \"frames_in_flight\": %u, \"ms\": %.3f, \"ms_per_frame\": %.3f, \"segments\": %llu, \"segments_per_s\": %.4g, \"tasks\": %zu}\n",
               frames, manager.framesInFlight(), ms, ms / std::max(frames, 1), seg, (double)seg / (ms * 1e-3), manager.tasks().size());
        frames = 0;
    }
    for (int f = 0; f < frames; f++) {
        for (auto &dpt : manager.tracers()) dpt->resetCounters();
        auto start = std::chrono::high_resolution_clock::now();
        manager.renderFrame();
        auto stop = std::chrono::high_resolution_clock::now();
        double ms = std::chrono::duration<double, std::milli>(stop - start).count();
        std::cout << "Path Tracing took: " << (long)ms << "ms" << std::endl;
        unsigned long long seg = 0;
        for (auto &dpt : manager.tracers()) seg += dpt->readCounters().segments;
        printf("{\"frame\": %d, \"ms\": %.3f, \"segments\": %llu, \"segments_per_s\": %.4g, \"imbalance\": %.3f, \"tasks\": %zu}\n", f, ms, seg,
               (double)seg / (ms * 1e-3), manager.imbalance(), manager.tasks().size());
    }
    if (out.empty()) return 0;                                   // (--out "": no picture wanted)
    if (pt_write_ppm(out.c_str(), (int)manager.getCurrentFrameWidth(), (int)manager.getCurrentFrameHeight(), manager.getCurrentFrame()) != PT_OK) {
        std::cerr << pt_last_error_string() << std::endl;
        return 1;
    }
    if (!yuvOut.empty()) {
        FILE *fp = fopen(yuvOut.c_str(), "wb");
        size_t total = (size_t)manager.getCurrentFrameWidth() * manager.getCurrentFrameHeight();
        if (fp) { fwrite(manager.getYUVFrame(), 1, total + 2 * (total / 4), fp); fclose(fp); }
    }
    std::cout << "wrote " << out << std::endl;
    return 0;
}
