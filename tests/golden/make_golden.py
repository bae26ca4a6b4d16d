#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/pt_oracle.c).

The reference ships no golden vectors and cannot be built or run here (no CUDA; see
DESIGN.md), so these fixtures are outputs of the ORACLE, committed so that (a) the oracle
itself is regression-pinned and (b) the GPU box can check the HIP path against the same
bytes without re-deriving them.  Inputs are the reference's own models/*.glb (data files,
copied to models/).  Run from the repo root:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pt = importlib.import_module("multi-gpu-path-tracer_amd")
from oracle import binding as ob  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
BOX_CAMERA = ((-255.5, 252.5, 700.0), (0.0, 0.0, -1.0), 45.0, 45.0)
DUCK_CAMERA = ((0.0, 0.0, 0.5), (0.0, 0.0, -0.5), 45.0, 45.0)

CASES = [
    # name, scene, camera, W, H, spp, depth
    ("box_64x64_4spp_4b", "box", BOX_CAMERA, 64, 64, 4, 4),
    ("box_160x90_8spp_8b", "box", BOX_CAMERA, 160, 90, 8, 8),
    ("duck_64x36_2spp_4b", "duck", DUCK_CAMERA, 64, 36, 2, 4),
    ("duck_96x54_3spp_8b", "duck", DUCK_CAMERA, 96, 54, 3, 8),
    # BASELINE.json configs[0] (C1, the CPU-runnable case) and the survey's larger duck fixture
    ("box_400x400_16spp_4b", "box", BOX_CAMERA, 400, 400, 16, 4),
    ("duck_160x90_4spp_8b", "duck", DUCK_CAMERA, 160, 90, 4, 8),
]


def main():
    scenes = {
        "box": pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_box.glb")).applyCornellConvention(),
        "duck": pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_duck.glb")),
    }
    oracles = {k: ob.OracleScene(s.triangles, s.materials, s.textures) for k, s in scenes.items()}
    for name, sc, cam, W, H, spp, depth in CASES:
        rgb, yuv, cnt = oracles[sc].render(cam, W, H, spp, depth, threads=8)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), rgb=rgb, yuv=yuv,
                            counters=np.array([cnt[k] for k in ("paths", "segments", "box_tests", "tri_tests",
                                                                "light_tests", "rng_draws", "max_stack")], np.uint64),
                            params=np.array([W, H, spp, depth], np.int64))
        print(name, cnt)
    # BVH known answers (reference topology): full node arrays + triangle permutation + lights
    for sc in ("box", "duck"):
        b, l, i, lights = oracles[sc].bvh()
        np.savez_compressed(os.path.join(OUT, "bvh_%s.npz" % sc), bounds=b, links=l, indices=i, lights=lights)
    # HostScene dumps of the two models as produced by the GLB loader (loader regression pin)
    for sc in ("box", "duck"):
        s = scenes[sc]
        np.savez_compressed(os.path.join(OUT, "hostscene_%s.npz" % sc), triangles=s.triangles, materials=s.materials)
    # XORWOW: first 8 raw outputs + uniforms for three seeds
    seeds = [1984, 1985, 1984 + 1920 * 1080 - 1]
    raw = np.stack([ob.xorwow_uniforms(s, 8)[0] for s in seeds])
    uni = np.stack([ob.xorwow_uniforms(s, 8)[1] for s in seeds])
    np.savez_compressed(os.path.join(OUT, "xorwow.npz"), seeds=np.array(seeds, np.uint64), raw=raw, uniforms=uni)
    # sin/cos contract samples
    x = np.linspace(0.0, 2 * np.pi, 1025).astype(np.float32)
    sc_ = np.array([ob.sincos(float(v)) for v in x], np.float32)
    np.savez_compressed(os.path.join(OUT, "sincos.npz"), x=x, sin=sc_[:, 0], cos=sc_[:, 1])


if __name__ == "__main__":
    main()
