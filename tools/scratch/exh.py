import sys, os; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import importlib, time
import numpy as np
pt = importlib.import_module("multi-gpu-path-tracer_amd")
sc = pt.SceneLoader().load("models/cornell_box.glb").applyCornellConvention()
fb = pt.Framebuffer(pt.Resolution(8, 8))
d = pt.DevicePathTracer(0, 1, 1, (8, 8), sc, fb, pt.CameraConfig((0,0,5),(0,0,-1)))
for op in (0, 1, 2):
    bad, ex = d.katExhaustive(op, 0, 1 << 32, 8)
    print("op", op, "bad", bad, ["%08x" % b for b in ex])
