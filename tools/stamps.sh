#!/bin/bash
# Diagnostic: build a stamped copy of the library (PT_STAMPS), run a tiny image so every wave has a
# SIMD to itself, print where a lone wave's cycles go.  The stamped copy is built beside the product library
# (build/variants/, loaded through PTAMD_LIBRARY): nothing is overwritten.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p build/variants
make -C multi-gpu-path-tracer_amd/csrc -B EXTRA=-DPT_STAMPS OUT=../../build/variants/libptamd_stamps.so ../../build/variants/libptamd_stamps.so > /dev/null 2>&1
PTAMD_LIBRARY=$PWD/build/variants/libptamd_stamps.so python3 - <<'PY'
import sys, importlib, ctypes as C
sys.path.insert(0, ".")
pt = importlib.import_module("multi-gpu-path-tracer_amd")
sc = pt.SceneLoader().load("models/cornell_box.glb").applyCornellConvention()
for (W, H) in [(64, 32), (1920, 1080)]:
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, 256, 8, (8, 8), sc, fb, pt.CameraConfig((-255.5, 252.5, 700.0), (0.0, 0.0, -1.0)))
    tr.resetCounters(); tr.resetKernelTime()
    tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
    import torch
    raw = (C.c_uint64 * 8)()
    # counters[5..7] are read through the raw device buffer: reuse pt_counters_read's layout by hipMemcpy via torch
    ms, n = tr.kernelTime()
    c = tr.readCounters()
    print(W, H, "kernel ms", ms, "segments", c["segments"], "stamps:", getattr(tr, "readStamps", lambda: None)())
    tr.close()
PY
