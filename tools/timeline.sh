#!/bin/bash
# Diagnostic (GPU box): build with -DPT_TIMELINE, render frames of a bench.py config, dump per-tile {start, end, hw id, cost}
# (100 MHz counter) of the last launch, in launch order, to gpurun_out/timeline_<tag>.npy, restore the normal build.
# usage: bash tools/timeline.sh <tag> [config=c2] [sah]   (c4 always uses the SAH tree)
set -e
cd $GRAFT_REPO_ROOT; TAG=${1:-c2}; CFG=${2:-c2}; SAH=${3:-}
# the variant is built BESIDE the product library and loaded through PTAMD_LIBRARY: nothing is overwritten, nothing to restore
mkdir -p build/variants
make -C multi-gpu-path-tracer_amd/csrc -B EXTRA=-DPT_TIMELINE OUT=../../build/variants/libptamd_timeline.so ../../build/variants/libptamd_timeline.so > /dev/null 2>&1
PTAMD_LIBRARY=$PWD/build/variants/libptamd_timeline.so python3 - $TAG $CFG $SAH <<'PY'
import sys, importlib, ctypes as C, numpy as np
sys.path.insert(0, ".")
import bench
pt = importlib.import_module("multi-gpu-path-tracer_amd")
tag, cfg = sys.argv[1], sys.argv[2]
kind, W, H, spp, depth, desc = bench.CONFIGS[cfg]
scene, camera = bench.load_scene(pt, kind)
rc = pt.RendererConfig(samplesPerPixel=spp, recursionDepth=depth, gpuNumber=1, resolution=pt.Resolution(W, H), showTasks=False,
                       algorithmType=pt.FSFL)
mgr = pt.RenderManager(rc, scene, pt.CameraConfig(camera[0], camera[1]), device_idx=0, band_rows=8, bvh_mode=1 if (cfg == "c4" or len(sys.argv) > 3) else 0)
tr = mgr.tracer
for _ in range(3):
    tr.resetKernelTime()
    mgr.renderFrame()
ms, n = tr.kernelTime()
slots = ((W + 7) // 8) * ((H + 7) // 8)
buf = np.zeros(4 * slots, np.uint32)
fn = pt.capi.lib().pt_debug_timeline_read
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert fn(tr._h, buf.ctypes.data, buf.size) == 0
np.save("gpurun_out/timeline_%s.npy" % tag, buf.reshape(-1, 4))
print("timeline", tag, cfg, tr.kernelName(), "kernel ms", ms / n, "tiles", slots)
PY
