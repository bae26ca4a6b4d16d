#!/usr/bin/env python3
"""On the GPU box: per-exponent failure map of the kernels' short division / sqrt forms against the IEEE
expressions (pt_kat_exhaustive).  op 0: 1/x guarded, 1: x/pi, 2: 1/x Moller-Trumbore form, 3: unguarded
reciprocal core (fails for denormal x and |x| > 2^126), 4: sqrt.  usage: python tools/exhaustive_probe.py [op...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pt = importlib.import_module("multi-gpu-path-tracer_amd")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sc = pt.SceneLoader().load(os.path.join(root, "models", "cornell_box.glb")).applyCornellConvention()
d = pt.DevicePathTracer(0, 1, 1, (8, 8), sc, pt.Framebuffer(pt.Resolution(8, 8)), pt.CameraConfig((0, 0, 5), (0, 0, -1)))
for op in [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4]:
    per = []
    for e in range(512):                      # sign bit + 8 exponent bits
        bad, ex = d.katExhaustive(op, e << 23, 1 << 23, 2)
        if bad:
            per.append((e, bad, ["%08x" % b for b in ex]))
    print("op", op, "inputs that differ:", sum(p[1] for p in per), "in", len(per), "sign/exponent classes", per[:8])
