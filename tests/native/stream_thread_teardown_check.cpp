// tests/native/stream_thread_teardown_check.cpp -- deterministic CPU test of the worker teardown.
//
// The race fixed in 6d7caea: StreamThread::threadMain used to test `shouldTerminate` after the frame-END barrier, as the
// reference's does (/root/reference/src/StreamThread.h:94-101).  A worker that wakes from that barrier late -- after
// RenderManager::reset() has already raised the flag -- then left, reset()'s release barrier (which counts every worker)
// never filled, and the workers that had gone on to the frame-START barrier waited with it forever.
//
// This program compiles the product's own host headers (RenderManager.h, StreamThread.h, barrier.h, DevicePathTracer.h)
// against a stub of the C ABI (no HIP, no GPU), and holds worker 0 for 300 ms right after the frame-end barrier through
// the header's test seam while the main thread runs ~RenderManager() -> reset(): exactly the interleaving that hung.
//   default build                     : must terminate (exit 0)
//   -DOLD_CHECK_AFTER_FRAME_END       : re-creates the pre-fix check inside the seam; must NOT terminate (the Python test
//                                       runs it under a timeout and expects the timeout) -- the test can fail.
// Also run under -fsanitize=thread by tests/test_host_logic.py.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

static std::atomic<int> g_after_frame_end{0};
#ifdef OLD_CHECK_AFTER_FRAME_END
#define PTAMD_TEST_AFTER_FRAME_END                                                                  \
    { if (taskIdx == 0) { g_after_frame_end++; std::this_thread::sleep_for(std::chrono::milliseconds(300)); } \
      if (shouldTerminate) return; }
#else
#define PTAMD_TEST_AFTER_FRAME_END                                                                  \
    { if (taskIdx == 0) { g_after_frame_end++; std::this_thread::sleep_for(std::chrono::milliseconds(300)); } }
#endif

#include "host/RenderManager.h"

// ---- stub C ABI: just enough of include/pt_amd.h for the host headers to drive (the task arithmetic and the error string are the
// product's own pt_common.cpp + pt_bvh.cpp, plain C++, linked in) ----------------------------------------------------------
struct pt_device { int index; };
static std::atomic<int> g_renders{0};
extern "C" {
pt_status pt_device_create(int i, pt_device **out) { *out = new pt_device{i}; return PT_OK; }
pt_status pt_device_destroy(pt_device *d) { delete d; return PT_OK; }
pt_status pt_set_thread_block_size(pt_device *, unsigned, unsigned) { return PT_OK; }
pt_status pt_upload_scene_strided(pt_device *, const void *, size_t, const pt_triangle_layout *, const pt_material *, size_t, const pt_texture *, size_t) { return PT_OK; }
pt_status pt_fb_alloc(pt_device *, int w, int h, uint8_t **rgb, uint8_t **yuv) { *rgb = new uint8_t[(size_t)w * h * 3](); *yuv = new uint8_t[(size_t)w * h * 3 / 2](); return PT_OK; }
pt_status pt_fb_free(pt_device *, uint8_t *rgb, uint8_t *yuv) { delete[] rgb; delete[] yuv; return PT_OK; }
pt_status pt_bind_framebuffer(pt_device *, int, int, uint8_t *, uint8_t *) { return PT_OK; }
pt_status pt_render_task_async(pt_device *, const pt_task *, const pt_camera *, unsigned, unsigned, pt_stream) { g_renders++; return PT_OK; }
pt_status pt_stream_sync(pt_device *, pt_stream) { return PT_OK; }
pt_status pt_device_sync(pt_device *) { return PT_OK; }
pt_status pt_fb_download_task(pt_device *, const pt_task *, uint8_t *, uint8_t *, pt_stream) { return PT_OK; }
pt_status pt_stream_create(pt_device *, pt_stream *out) { *out = (pt_stream) new int(0); return PT_OK; }
pt_status pt_stream_destroy(pt_device *, pt_stream s) { delete (int *)s; return PT_OK; }
pt_status pt_counters_read(pt_device *, pt_counters *c) { memset(c, 0, sizeof *c); return PT_OK; }
pt_status pt_counters_reset(pt_device *) { return PT_OK; }
pt_status pt_set_bvh_mode(pt_device *, int) { return PT_OK; }
pt_status pt_set_legacy_scene(pt_device *, const pt_legacy_material *, size_t, const pt_sphere *, size_t) { return PT_OK; }
pt_status pt_set_cooperative_mode(pt_device *, int) { return PT_OK; }
pt_status pt_set_tile_schedule(pt_device *, int) { return PT_OK; }
pt_status pt_gather_create(pt_device *const *, int, pt_gather **) { return PT_ERR_UNSUPPORTED; }
pt_status pt_gather_destroy(pt_gather *) { return PT_OK; }
pt_status pt_gather_transport(const pt_gather *, char *out, size_t cap) { if (cap) out[0] = 0; return PT_OK; }
pt_status pt_gather_frame(pt_gather *, const pt_task *, const int *, int, uint8_t *, uint8_t *) { return PT_OK; }
pt_status pt_scene_load_glb(const char *, pt_host_scene **) { return PT_ERR_IO; }
pt_status pt_scene_load_obj(const char *, pt_host_scene **) { return PT_ERR_IO; }
pt_status pt_scene_apply_cornell_convention(pt_host_scene *) { return PT_OK; }
void pt_scene_free(pt_host_scene *) {}
size_t pt_scene_triangle_count(const pt_host_scene *) { return 0; }
size_t pt_scene_material_count(const pt_host_scene *) { return 0; }
size_t pt_scene_texture_count(const pt_host_scene *) { return 0; }
const pt_triangle *pt_scene_triangles(const pt_host_scene *) { return nullptr; }
const pt_material *pt_scene_materials(const pt_host_scene *) { return nullptr; }
const pt_texture *pt_scene_textures(const pt_host_scene *) { return nullptr; }
const char *pt_scene_material_name(const pt_host_scene *, size_t) { return ""; }
}

int main(int argc, char **argv)
{
    using namespace ptamd;
    const unsigned workers = argc > 1 ? (unsigned)atoi(argv[1]) : 8;
    const int frames = argc > 2 ? atoi(argv[2]) : 3;
    HostScene scene;
    scene.triangles.resize(1);
    scene.materials.resize(1);
    CameraConfig cam(Float3{0.0f, 0.0f, 1.0f}, Float3{0.0f, 0.0f, -1.0f});
    SceneLoader loader;
    RendererConfig cfg;
    cfg.gpuNumber = 1;                       // one engine, `workers` streams: the thin-frame DSDL test's shape (8 workers, 16 x 8 pixels)
    cfg.streamsPerGpu = workers;
    cfg.resolution = Resolution{16, 8};
    {
        RenderManager mgr(cfg, scene, cam, loader);
        for (int f = 0; f < frames; f++) mgr.renderFrame();
        // ~RenderManager() -> reset(): raises every worker's flag, joins the frame-START barrier, joins the threads --
        // while worker 0 is still asleep behind the frame-END barrier of the last frame
    }
    printf("teardown ok: %u workers, %d frames, %d renders, worker 0 was held %d times after a frame-end barrier\n", workers, frames,
           g_renders.load(), g_after_frame_end.load());
    return g_after_frame_end.load() == frames && g_renders.load() == (int)workers * frames ? 0 : 1;
}
