#!/bin/bash
# Diagnostic (GPU box): build with -DPT_TIMELINE, render C2 frames, dump per-wave {start, end, hw id, cost} of the last
# launch to gpurun_out/timeline_<tag>.npy, restore the normal build.  usage: bash tools/timeline.sh <tag> [height]
set -e
cd $GRAFT_REPO_ROOT; TAG=${1:-c2}; HEIGHT=${2:-1080}
cp multi-gpu-path-tracer_amd/libptamd.so /tmp/lib_keep.so
make -C multi-gpu-path-tracer_amd/csrc -B EXTRA=-DPT_TIMELINE ../libptamd.so > /dev/null 2>&1
python3 - $TAG $HEIGHT <<'PY'
import sys, importlib, ctypes as C, numpy as np
sys.path.insert(0, ".")
pt = importlib.import_module("multi-gpu-path-tracer_amd")
tag, H = sys.argv[1], int(sys.argv[2]); W = 1920
sc = pt.SceneLoader().load("models/cornell_box.glb").applyCornellConvention()
fb = pt.Framebuffer(pt.Resolution(W, H))
tr = pt.DevicePathTracer(0, 256, 8, (8, 8), sc, fb, pt.CameraConfig((-255.5, 252.5, 700.0), (0.0, 0.0, -1.0)))
for _ in range(3):
    tr.resetKernelTime()
    tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
ms, n = tr.kernelTime()
slots = (W // 8) * ((H + 7) // 8)
buf = np.zeros(4 * slots, np.uint32)
fn = pt.capi.lib().pt_debug_timeline_read
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
assert fn(tr._h, buf.ctypes.data, buf.size) == 0
np.save("gpurun_out/timeline_%s.npy" % tag, buf.reshape(-1, 4))
print("timeline", tag, "kernel ms", ms / n, "waves", slots)
tr.close()
PY
cp /tmp/lib_keep.so multi-gpu-path-tracer_amd/libptamd.so
