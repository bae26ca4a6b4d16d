#!/bin/bash
# Diagnostic (GPU box): how long the dearest tiles of a config take ALONE on the device (one 8x8 task each) against their
# duration inside the full frame.  usage: bash tools/lone_tile.sh [config=c4] [n=4]
set -e
cd $GRAFT_REPO_ROOT; CFG=${1:-c4}; N=${2:-4}
# the variant is built BESIDE the product library and loaded through PTAMD_LIBRARY: nothing is overwritten, nothing to restore
mkdir -p build/variants
make -C multi-gpu-path-tracer_amd/csrc -B EXTRA=-DPT_TIMELINE OUT=../../build/variants/libptamd_timeline.so ../../build/variants/libptamd_timeline.so > /dev/null 2>&1
PTAMD_LIBRARY=$PWD/build/variants/libptamd_timeline.so python3 - $CFG $N <<'PY'
import os, sys, importlib, ctypes as C, numpy as np
sys.path.insert(0, ".")
import bench
pt = importlib.import_module("multi-gpu-path-tracer_amd")
cfg, n_top = sys.argv[1], int(sys.argv[2])
kind, W, H, spp, depth, desc = bench.CONFIGS[cfg]
scene, camera = bench.load_scene(pt, kind)
rc = pt.RendererConfig(samplesPerPixel=spp, recursionDepth=depth, gpuNumber=1, resolution=pt.Resolution(W, H), showTasks=False,
                       algorithmType=pt.FSFL)
mgr = pt.RenderManager(rc, scene, pt.CameraConfig(camera[0], camera[1]), device_idx=0, band_rows=8, bvh_mode=1 if cfg == "c4" else 0)
tr = mgr.tracer
for _ in range(3):
    tr.resetKernelTime()
    mgr.renderFrame()
print("frame kernel ms %.2f" % (tr.kernelTime()[0] / tr.kernelTime()[1]))
slots = ((W + 7) // 8) * ((H + 7) // 8)
buf = np.zeros(4 * slots, np.uint32)
fn = pt.capi.lib().pt_debug_timeline_read
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert fn(tr._h, buf.ctypes.data, buf.size) == 0
a = buf.reshape(-1, 4).copy()
tiles_x = (W + 7) // 8
for k in np.argsort(-a[:, 3].astype(np.int64))[:n_top]:
    tile, cost = int(a[k, 2]) & 0x7fffffff, int(a[k, 3])
    in_frame_ms = (int(a[k, 1]) - int(a[k, 0])) / 1e5
    tx, ty = tile % tiles_x, tile // tiles_x
    # a full-frame task starts at (0, 0), so tile (tx, ty) of the frame is the 8x8 task at (8 tx, 8 ty)
    best = 1e9
    for _ in range(3):
        tr.resetKernelTime()
        tr.renderTaskAsync(pt.RenderTask(8, 8, tx * 8, ty * 8)); tr.waitForRenderTask()
        ms, n = tr.kernelTime(); best = min(best, ms / n)
    print("tile %d (%d,%d) cost %d: %.2f ms in the frame, %.2f ms alone" % (tile, tx, ty, cost, in_frame_ms, best))
    for w, h in (((8, 4), (8, 2), (8, 1), (4, 1), (1, 1)) if os.environ.get("PARTS") else ()):      # parts of the tile: how the chain shortens with fewer pixels per wave
        worst = 0.0
        for oy in range(0, 8, h):
            for ox in range(0, 8, w):
                tr.resetKernelTime()
                tr.renderTaskAsync(pt.RenderTask(w, h, tx * 8 + ox, ty * 8 + oy)); tr.waitForRenderTask()
                ms, n = tr.kernelTime(); worst = max(worst, ms / n)
        print("    as %dx%d parts: slowest part %.2f ms" % (w, h, worst))
PY
