#!/usr/bin/env python3
"""tools/scene_sweep.py -- one sweep BETWEEN the five BASELINE configs (VERDICT r3 item 7 / missing 7): the synthetic terrain at
grids 16 .. 2048 (512 .. 8.4 M triangles) under both trees, 1920x1080, 16 spp, 8 bounces, one frame at a time; per scene the
kernel the library chose, ms per frame (second frame: cost-ordered tiles), segments and tests per segment.  launch_render's
thresholds (<= 64 inner nodes: dealt mask walk; <= 4 096: dealt stack walk; beyond: per-lane walk; SAH: 128-byte nodes
up to 64 MB, 64-byte nodes beyond) were tuned on C1-C5 only.  Reference trees get dearer with every triangle (unpruned walk):
a config whose 1-spp probe predicts more than BUDGET_S seconds is measured at 1 spp and marked so, beyond 10 x that skipped.

  GPU box:  python tools/scene_sweep.py > gpurun_out/r04_scene_sweep.json
"""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pt = importlib.import_module("multi-gpu-path-tracer_amd")

W, H, SPP, DEPTH = 1920, 1080, 16, 8
CAMERA = ((0.0, 220.0, 620.0), (0.0, -0.35, -1.0))
BUDGET_S = float(os.environ.get("SWEEP_BUDGET_S", "20"))
grids = [int(g) for g in os.environ.get("SWEEP_GRIDS", "16 32 48 64 96 128 192 256 384 512 768 1024 1536 2048").split()]
rows = []
for grid in grids:
    scene = pt.make_terrain_scene(grid)
    for tree in ("reference", "sah"):
        row = {"grid": grid, "triangles": len(scene.triangles), "tree": tree}
        try:
            fb = pt.Framebuffer(pt.Resolution(W, H))
            t0 = time.perf_counter()
            tr = pt.DevicePathTracer(0, 1, DEPTH, (8, 8), scene, fb, pt.CameraConfig(*CAMERA), bvhMode=1 if tree == "sah" else 0)
            row["upload_s"] = round(time.perf_counter() - t0, 3)
            info = tr.bvhInfo()
            row["reference_tree"] = {k: info[k] for k in ("nodes", "leaves", "max_leaf", "depth")}

            def frame(spp):
                tr.setSamplesPerPixel(spp)
                tr.resetCounters(); tr.resetKernelTime()
                tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
                ms, n = tr.kernelTime()
                return ms / max(n, 1), tr.readCounters()
            probe_ms, _ = frame(1)
            predicted_s = probe_ms * SPP * 1e-3
            row["probe_1spp_ms"] = round(probe_ms, 3)
            if predicted_s > 10 * BUDGET_S:
                row["skipped"] = "1-spp probe predicts %.0f s per 16-spp frame" % predicted_s
            else:
                spp = SPP if predicted_s <= BUDGET_S else 1
                frame(spp)                                   # first frame of this spp: raster order
                ms, cnt = frame(spp)                         # second: cost-ordered tiles
                row.update({"spp": spp, "ms_per_frame": round(ms, 3), "ms_per_16spp_frame": round(ms * SPP / spp, 3),
                            "extrapolated_from_1spp": spp != SPP, "kernel": tr.kernelName(), "segments": cnt["segments"],
                            "box_tests_per_segment": round(cnt["box_tests"] / max(cnt["segments"], 1), 2),
                            "tri_tests_per_segment": round(cnt["tri_tests"] / max(cnt["segments"], 1), 2),
                            "segments_per_s": cnt["segments"] / (ms * 1e-3)})
            tr.close()
        except Exception as e:                               # a refusal (LDS, stack depth) is a result too
            row["error"] = str(e)[:300]
        rows.append(row)
        sys.stderr.write(json.dumps(row) + "\n")
        sys.stderr.flush()
print(json.dumps({"config": {"width": W, "height": H, "spp": SPP, "bounces": DEPTH, "camera": CAMERA, "frames_in_flight": 1,
                             "note": "kernel time of the second frame at the measured spp (HIP events, pt_kernel_time_read)"},
                  "rows": rows}, indent=1))
