#!/usr/bin/env python3
"""tools/c4_gate_diff.py [n_rays] [rays.npz] -- GPU box.  The C4-scale closest-hit gate (tests/test_gpu_c4.py) with the differing
rays SHOWN: for every ray on which the SAH walk and the reference-topology walk disagree, both answers, the CPU brute force
over all triangles with the oracle's triangle test, and the ray itself; the rays are saved to gpurun_out/c4_gate_diff.npz.
Given such a file, only its rays are evaluated (e.g. under another build through PTAMD_LIBRARY)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib                                                     # noqa: E402

pt = importlib.import_module("multi-gpu-path-tracer_amd")
from oracle import binding as ob                                     # noqa: E402  (the checker, as in the tests)
import test_gpu_c4 as g                                              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
terrain = pt.make_terrain_scene(2048)
if len(sys.argv) > 2:
    z = np.load(sys.argv[2]); org, dirs = z["org"], z["dirs"]
else:
    org, dirs = g._gate_rays(terrain, n)
fb = pt.Framebuffer(pt.Resolution(8, 8))


def walk(mode, o, d):
    tr = pt.DevicePathTracer(0, 1, 1, (8, 8), terrain, fb, pt.CameraConfig(*g.TERRAIN_CAMERA), bvhMode=mode)
    tr.renderTaskAsync(pt.RenderTask(8, 8, 0, 0)); tr.waitForRenderTask()
    t, f = np.empty(len(o), np.float32), np.empty(len(o), np.int32)
    for a in range(0, len(o), 200_000):
        t[a:a + 200_000], f[a:a + 200_000] = tr.katClosestHit(o[a:a + 200_000], d[a:a + 200_000])
    name = tr.kernelName()
    tr.close()
    return t, f, name


t_s, f_s, name = walk(1, org, dirs)
t_r, f_r, _ = walk(0, org, dirs)
bad = np.nonzero((t_s.view(np.uint32) != t_r.view(np.uint32)) | (f_s != f_r))[0]
print("%s: %d of %d rays differ" % (name, len(bad), len(org)))
if len(bad):
    t_b, f_b, ties = ob.brute_closest(terrain.triangles, org[bad], dirs[bad], threads=min(16, os.cpu_count() or 1))
    for k, i in enumerate(bad):
        print("ray %d: o = %r d = %r" % (i, org[i].tolist(), dirs[i].tolist()))
        print("   SAH walk        t = %.9g (0x%08x) face %d" % (t_s[i], t_s[i:i + 1].view(np.uint32)[0], f_s[i]))
        print("   reference walk  t = %.9g (0x%08x) face %d" % (t_r[i], t_r[i:i + 1].view(np.uint32)[0], f_r[i]))
        print("   brute force     t = %.9g (0x%08x) face %d, %d tied" % (t_b[k], t_b[k:k + 1].view(np.uint32)[0], f_b[k], ties[k]))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez(os.path.join(ROOT, "gpurun_out", "c4_gate_diff.npz"), org=org[bad], dirs=dirs[bad])
