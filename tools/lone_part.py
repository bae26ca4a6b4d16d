#!/usr/bin/env python3
"""GPU box: how long the dearest C2 tile's pixels take ALONE on the device as parts of w x h pixels, under each walk
schedule -- what bounds a rank's share of a multi-GPU frame is the chain of its dearest pixels, i.e. the latency of a
sparse wave's trip.  usage: python tools/lone_part.py [config=c2]"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pt = importlib.import_module("multi-gpu-path-tracer_amd")
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
kind, W, H, spp, depth, desc = bench.CONFIGS[cfg]
scene, camera = bench.load_scene(pt, kind)
fb = pt.Framebuffer(pt.Resolution(W, H))
tr = pt.DevicePathTracer(0, spp, depth, (8, 8), scene, fb, pt.CameraConfig(camera[0], camera[1]))
# the dearest tile of the box frame (round 2's timelines): tile 7495 = (tx 55, ty 31) of 240 x 135
tx, ty = 7495 % 240, 7495 // 240
out = {}
for mode, name in ((4, "dealt mask walk"), (2, "mask walk"), (1, "per-lane walk")):
    tr.setTraversalMode(mode)
    for (w, h) in ((8, 8), (8, 2), (4, 1), (1, 1)):
        worst = 0.0
        for oy in range(0, 8, h):
            for ox in range(0, 8, w):
                best = 1e9
                for _ in range(2):
                    tr.resetKernelTime()
                    tr.renderTaskAsync(pt.RenderTask(w, h, tx * 8 + ox, ty * 8 + oy)); tr.waitForRenderTask()
                    ms, n = tr.kernelTime(); best = min(best, ms / n)
                worst = max(worst, best)
        out["%s %dx%d" % (name, w, h)] = round(worst, 2)
        print("%-16s parts of %dx%d: slowest part %.2f ms  (%s)" % (name, w, h, worst, tr.kernelName()), flush=True)
json.dump(out, open("gpurun_out/lone_part_%s.json" % cfg, "w"), indent=1)
