#!/bin/bash
# Diagnostic (GPU box): share of the SAH walk's node visits that only find both children beyond the closest hit so far
# (entries pushed before the hit was found; -DPT_STALE build under build/variants/libptamd_stale.so, made here with
# `make -C multi-gpu-path-tracer_amd/csrc -B EXTRA=-DPT_STALE OUT=../../build/variants/libptamd_stale.so`).
set -e
cd $GRAFT_REPO_ROOT
PTAMD_LIBRARY=$PWD/build/variants/libptamd_stale.so python3 - "$@" <<'PY'
import sys, importlib
sys.path.insert(0, ".")
import bench
pt = importlib.import_module("multi-gpu-path-tracer_amd")
for cfg in sys.argv[1:] or ["c4", "c3"]:
    kind, W, H, spp, depth, desc = bench.CONFIGS[cfg]
    scene, camera = bench.load_scene(pt, kind)
    rc = pt.RendererConfig(samplesPerPixel=spp, recursionDepth=depth, gpuNumber=1, resolution=pt.Resolution(W, H), showTasks=False, algorithmType=pt.FSFL)
    mgr = pt.RenderManager(rc, scene, pt.CameraConfig(camera[0], camera[1]), device_idx=0, band_rows=8, bvh_mode=1)
    tr = mgr.tracer
    tr.resetCounters()
    mgr.renderFrame()
    st = tr.readStamps(); c = tr.readCounters()
    stale, visits = st["longest_wave_cycles"], st["sum_traversal_cycles"]
    print("%s: node visits %d (box tests / 2 = %d), stale %d = %.1f %%; per segment %.1f visits, %.1f triangle tests" %
          (cfg, visits, c["box_tests"] // 2, stale, 100.0 * stale / max(visits, 1), visits / c["segments"], c["tri_tests"] / c["segments"]))
    mgr.reset() if hasattr(mgr, "reset") else None
PY
