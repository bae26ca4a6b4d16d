#!/usr/bin/env python3
"""Full-size oracle fixtures: the frames BASELINE.json's configs are quoted on, rendered ONCE by the CPU oracle
(oracle/pt_oracle.c) and kept as sha256(RGB || YUV) + the exact work counters + one sha256 per band of 8 framebuffer
rows (so that a mismatch on the GPU box names the rows).  ~3 min (C2 box), ~8 min (duck) and ~40 min (C5 box) on 8 threads; the GPU suite
compares the HIP path's frame with these at the size bench.py runs (tests/test_gpu_pipeline.py).

  python tests/golden/make_fullsize.py [case ...]
"""
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pt = importlib.import_module("multi-gpu-path-tracer_amd")
from oracle import binding as ob  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "fullsize.json")
BOX_CAMERA = ((-255.5, 252.5, 700.0), (0.0, 0.0, -1.0), 45.0, 45.0)
DUCK_CAMERA = ((0.0, 0.0, 0.5), (0.0, 0.0, -0.5), 45.0, 45.0)
CASES = {
    # BASELINE.json configs[1] (C2, the configuration the metric is quoted on)
    "box_1920x1080_256spp_8b": ("box", BOX_CAMERA, 1920, 1080, 256, 8),
    # configs[2]'s scene and frame size at the sample count the oracle finishes in minutes (reference tree: 2 120
    # triangle tests per segment)
    "duck_1920x1080_16spp_8b": ("duck", DUCK_CAMERA, 1920, 1080, 16, 8),
    # configs[2] (C3) at its full sample count: 1.33e9 segments x 2 120 triangle tests, ~2 h on 8 threads
    "duck_1920x1080_256spp_8b": ("duck", DUCK_CAMERA, 1920, 1080, 256, 8),
    # BASELINE.json configs[4] (C5) at its full size: 2.1e10 segments, ~40 min on 8 threads
    "box_3840x2160_1024spp_16b": ("box", BOX_CAMERA, 3840, 2160, 1024, 16),
}
KEYS = ("paths", "segments", "box_tests", "tri_tests", "light_tests")


def band_hashes(rgb, yuv, W, H, rows=8):
    """sha256 over the RGB rows, the Y rows and the U / V rows of each band of `rows` framebuffer rows."""
    out = []
    rgb = rgb.reshape(H, W * 3)
    y = yuv[:W * H].reshape(H, W)
    u = yuv[W * H:W * H + W * H // 4].reshape(H // 2, W // 2)
    v = yuv[W * H + W * H // 4:].reshape(H // 2, W // 2)
    for r0 in range(0, H, rows):
        r1 = min(H, r0 + rows)
        h = hashlib.sha256()
        for part in (rgb[r0:r1], y[r0:r1], u[(r0 + 1) // 2:(r1 + 1) // 2], v[(r0 + 1) // 2:(r1 + 1) // 2]):
            h.update(part.tobytes())
        out.append(h.hexdigest()[:16])
    return out


def main():
    want = sys.argv[1:] or list(CASES)
    scenes = {
        "box": lambda: pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_box.glb")).applyCornellConvention(),
        "duck": lambda: pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_duck.glb")),
    }
    doc = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in want:
        sc, cam, W, H, spp, depth = CASES[name]
        s = scenes[sc]()
        osc = ob.OracleScene(s.triangles, s.materials, s.textures)
        t0 = time.perf_counter()
        rgb, yuv, cnt = osc.render(cam, W, H, spp, depth, threads=os.cpu_count() or 8)
        dt = time.perf_counter() - t0
        doc[name] = {
            "params": [W, H, spp, depth], "scene": sc,
            "sha256_rgb_yuv": hashlib.sha256(rgb.tobytes() + yuv.tobytes()).hexdigest(),
            "counters": {k: int(cnt[k]) for k in KEYS},
            "band_sha16": band_hashes(rgb.reshape(-1), yuv.reshape(-1), W, H),
            "oracle_seconds": round(dt, 1), "oracle_threads": os.cpu_count() or 8,
        }
        print(name, doc[name]["sha256_rgb_yuv"], doc[name]["counters"], "%.1f s" % dt, flush=True)
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
