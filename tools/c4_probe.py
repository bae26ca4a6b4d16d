import sys, time, importlib, json
sys.path.insert(0, ".")
pt = importlib.import_module("multi-gpu-path-tracer_amd")
t0=time.time(); sc = pt.make_terrain_scene(2048); print("gen", time.time()-t0, flush=True)
cam = pt.CameraConfig((0.0,220.0,620.0),(0.0,-0.35,-1.0))
for (W,H,spp) in [(64,36,2),(128,72,4)]:
    fb = pt.Framebuffer(pt.Resolution(W,H))
    t0=time.time(); tr = pt.DevicePathTracer(0, spp, 8, (8,8), sc, fb, cam); print("upload+bvh", time.time()-t0, tr.bvhInfo(), flush=True)
    tr.resetCounters(); tr.resetKernelTime()
    t0=time.time(); tr.renderTaskAsync(pt.RenderTask(W,H,0,0)); tr.waitForRenderTask(); dt=time.time()-t0
    c = tr.readCounters()
    print(json.dumps({"W":W,"H":H,"spp":spp,"sec":dt,"counters":c,"tri_per_seg":c["tri_tests"]/c["segments"],"box_per_seg":c["box_tests"]/c["segments"],"tri_tests_per_s":c["tri_tests"]/dt}), flush=True)
    tr.close()
