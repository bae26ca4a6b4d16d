#!/bin/bash
# Diagnostic: build a stamped copy of the library (PT_STAMPS), run a tiny image so every wave has a
# SIMD to itself, print where a lone wave's cycles go.  The stamped copy is built beside the product library
# (build/variants/, loaded through PTAMD_LIBRARY): nothing is overwritten.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p build/variants
make -C multi-gpu-path-tracer_amd/csrc -B EXTRA=-DPT_STAMPS OUT=../../build/variants/libptamd_stamps.so ../../build/variants/libptamd_stamps.so > /dev/null 2>&1
PTAMD_LIBRARY=$PWD/build/variants/libptamd_stamps.so python3 - <<'PY'
import sys, importlib, json
sys.path.insert(0, ".")
pt = importlib.import_module("multi-gpu-path-tracer_amd")
sc = pt.SceneLoader().load("models/cornell_box.glb").applyCornellConvention()
out = {}
# 64x32 at the centre of the box: 32 tiles, every wave has a SIMD to itself; then the whole frame (7 waves per SIMD)
for (W, H, label) in [(64, 32, "lone_waves_64x32"), (1920, 1080, "full_frame_1920x1080")]:
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, 256, 8, (8, 8), sc, fb, pt.CameraConfig((-255.5, 252.5, 700.0), (0.0, 0.0, -1.0)))
    for _ in range(2):
        tr.resetCounters(); tr.resetKernelTime()
        tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
    ms, n = tr.kernelTime()
    c = tr.readCounters()
    raw = tr.readStamps()["raw"]
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    trips = c["segments"] / 64.0                      # a lower bound on wave trips (64 tracing lanes per trip)
    out[label] = {"kernel": tr.kernelName(), "kernel_ms": ms / n, "segments": c["segments"], "tiles": tiles,
                  "cycles_path_start": raw[5], "cycles_walk": raw[6], "cycles_shading": raw[7],
                  "share_path_start": raw[5] / max(raw[5] + raw[6] + raw[7], 1), "share_walk": raw[6] / max(raw[5] + raw[6] + raw[7], 1),
                  "share_shading": raw[7] / max(raw[5] + raw[6] + raw[7], 1),
                  "cycles_per_64_segments": {"path_start": raw[5] / trips, "walk": raw[6] / trips, "shading": raw[7] / trips}}
    tr.close()
print(json.dumps(out, indent=1))
open("gpurun_out/stamps_dealt.json", "w").write(json.dumps(out, indent=1))
PY
