#!/bin/bash
# Diagnostic (GPU box): where the SAH kernel's lane slots go -- per micro-step trip the lanes that ran the step, the vote's
# losers, the lanes waiting for the shading pass and the lanes without a pixel; per shading pass the lanes it served.
# -DPT_OCC build beside the product library (build/variants/, loaded through PTAMD_LIBRARY).  usage: bash tools/occupancy_census.sh [configs...]
set -e
cd $GRAFT_REPO_ROOT; mkdir -p build/variants gpurun_out
make -C multi-gpu-path-tracer_amd/csrc -B EXTRA=-DPT_OCC OUT=../../build/variants/libptamd_occ.so ../../build/variants/libptamd_occ.so > /dev/null 2>&1
PTAMD_LIBRARY=$PWD/build/variants/libptamd_occ.so python3 - "$@" <<'PY'
import sys, importlib, json, ctypes as C
sys.path.insert(0, ".")
import bench
pt = importlib.import_module("multi-gpu-path-tracer_amd")
fn = pt.capi.lib().pt_debug_occ_read
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
out = {}
for cfg in sys.argv[1:] or ["c4", "c3"]:
    kind, W, H, spp, depth, desc = bench.CONFIGS[cfg]
    scene, camera = bench.load_scene(pt, kind)
    rc = pt.RendererConfig(samplesPerPixel=spp, recursionDepth=depth, gpuNumber=1, resolution=pt.Resolution(W, H), showTasks=False, algorithmType=pt.FSFL)
    mgr = pt.RenderManager(rc, scene, pt.CameraConfig(camera[0], camera[1]), device_idx=0, band_rows=8, bvh_mode=1)
    tr = mgr.tracer
    raw = (C.c_uint64 * 8)()
    for frame in range(3):                       # the third frame runs in cost order with its dearest tiles split
        assert fn(tr._h, raw, 1) == 0
        tr.resetCounters(); tr.resetKernelTime()
        mgr.renderFrame()
    assert fn(tr._h, raw, 0) == 0
    o = [int(v) for v in raw]
    ms, n = tr.kernelTime()
    c = tr.readCounters()
    slots = 64.0 * max(o[0], 1)
    out[cfg] = {"kernel": tr.kernelName(), "kernel_ms_instrumented": ms / n, "segments": c["segments"],
                "micro_step_trips": o[0], "outer_trips": o[7],
                "lane_slots_of_micro_steps": {"ran_the_step": o[1] / slots, "wanted_the_other_kind": o[2] / slots,
                                              "ray_finished_waiting_for_shading": o[3] / slots, "no_pixel_to_work_on": o[4] / slots},
                "lanes_per_shading_pass": o[6] / max(o[5], 1), "micro_steps_per_segment_lane": o[1] / max(c["segments"], 1),
                "micro_step_trips_per_outer_trip": o[0] / max(o[7], 1)}
    mgr.reset() if hasattr(mgr, "reset") else None
print(json.dumps(out, indent=1))
open("gpurun_out/occupancy_census.json", "w").write(json.dumps(out, indent=1))
PY
