#!/usr/bin/env python3
"""On the GPU box: randomized parity campaign, HIP path vs the CPU oracle, bit-exact (bytes + counters).
Random triangle soups (tests/test_gpu_parity.py::_random_scene), random even frame sizes, spp, depth,
walk schedule (the dealt walk with random dealing thresholds / list capacities included), tile schedule, and a random
way of covering the frame: one task, a grid of rectangular tasks with ragged edges, single-launch row bands of a random
number of ranks, or the band lists a cost-fed dealer hands out.
--legacy adds random legacy material kinds and spheres (the pt_set_legacy_scene extension) to half the cases,
--sah also renders every case with the SAH tree and reports how many images differ (a statistic, not a gate).
usage: python tools/fuzz_parity.py [first_seed] [count] [--legacy] [--sah]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pt = importlib.import_module("multi-gpu-path-tracer_amd")
from oracle import binding as ob  # noqa: E402  (test infrastructure: this is a checker, not the product)
import test_gpu_parity as tgp  # noqa: E402


LEGACY = "--legacy" in sys.argv
if LEGACY:
    sys.argv.remove("--legacy")
SAH_STATS = {"cases": 0, "cases_differ": 0, "bytes": 0, "bytes_differ": 0} if "--sah" in sys.argv else None
if SAH_STATS is not None:
    sys.argv.remove("--sah")


def one(seed):
    rng = np.random.default_rng(10_000 + seed)
    tris, mats, texs, cam = tgp._random_scene(pt, seed)
    W, H = 2 * int(rng.integers(4, 49)), 2 * int(rng.integers(4, 33))
    spp, depth = int(rng.integers(1, 7)), int(rng.integers(0, 10))
    scene = pt.HostScene.from_arrays(tris, mats, texs)
    osc = ob.OracleScene(tris, mats, texs)
    legacy = spheres = None
    if LEGACY and rng.random() < 0.5:
        # extension: random legacy kinds on the non-emissive materials, a few random spheres
        legacy = np.zeros(len(mats), pt.capi.LEGACY_MATERIAL_DTYPE)
        for i in range(len(mats)):
            if mats["emissiveFactor"][i].max() > 0:
                continue
            legacy[i]["kind"] = int(rng.integers(0, 5))
            legacy[i]["albedo"] = rng.uniform(0.05, 1.0, 3) * (4.0 if legacy[i]["kind"] == 4 else 1.0)
            legacy[i]["param"] = float(rng.choice([0.0, 0.3, 1.7, 1.5, 1.0, 0.6]))
        spheres = np.zeros(int(rng.integers(0, 4)), pt.capi.SPHERE_DTYPE)
        for sp in spheres:
            sp["center"] = rng.uniform(-8, 8, 3); sp["radius"] = float(rng.choice([-1, 1])) * float(rng.uniform(0.3, 4.0))
            sp["materialIdx"] = int(rng.integers(0, len(mats)))
        osc.set_legacy(legacy, spheres)
    ref = osc.render(cam, W, H, spp, depth, threads=8)
    info = pt.host_bvh_build(tris)[0]
    small = info["nodes"] - info["leaves"] <= 64
    modes = [0, 1, 3, 5, 5] + ([2, 4, 4, 4] if small else [])       # 4 / 5 = mask / stack walk with dealt sparse leaves (the box's and the duck's default walks)
    mode = 0 if legacy is not None else int(rng.choice(modes))
    # the dealt walk's split between dealt and in-place leaves, and its list capacity, are read at pt_device_create
    os.environ["PTAMD_DEAL_MIN"] = str(int(rng.choice([0, 8, 24, 48, 56, 65])))
    os.environ["PTAMD_DEAL_CAP"] = str(int(rng.choice([64, 128, 768])))
    # so are: from how many workgroups on a launch gets a cost-ordered schedule (default 64: never on frames this small),
    # and the SAH kernel's split of its dearest tiles (10 % of an even launch's cost: always some)
    os.environ["PTAMD_SCHED_MIN_BLOCKS"] = str(int(rng.choice([1, 2, 64])))
    os.environ["PTAMD_SPLIT_PCT"] = str(int(rng.choice([10, 90])))
    os.environ["PTAMD_SAH_SPLIT"] = str(int(rng.choice([0, 4, 16, 64])))
    os.environ["PTAMD_DEAL_SPLIT"] = str(int(rng.choice([4, 4, 8, 2])))     # dealt kernels: waves per split tile ...
    os.environ["PTAMD_PART_LANES"] = str(int(rng.choice([1, 1, 0])))        # ... whose parts walk by lanes where the tree allows it
    os.environ["PTAMD_SAH_WIDE"] = str(int(rng.choice([0, 1, 1])))          # the SAH tree's device form: pairs or 4-wide nodes
    os.environ["PTAMD_SAH_QUANT"] = str(int(rng.choice([0, 1])))            # ... in 128-byte or in 64-byte (quantised) records
    os.environ["PTAMD_SAH_STACK"] = str(int(rng.choice([0, 11, 16])))       # ... with the whole stack in LDS or a short one + overflow area
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), scene, fb, pt.CameraConfig(cam[0], cam[1], cam[2], cam[3]),
                             legacyMaterials=legacy, spheres=spheres)
    tr.setTraversalMode(mode)
    tr.setTileSchedule(int(rng.integers(0, 2)))
    cover = int(rng.integers(0, 4))
    frames = int(rng.integers(1, 4))                      # later frames run in cost order
    for _ in range(frames):
        fb.rgb.zero_(); fb.yuv.zero_()
        tr.resetCounters()
        if cover == 0:
            tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0))
        elif cover == 1:
            xs = sorted(set([0, W] + [int(v) for v in rng.integers(1, W, int(rng.integers(0, 4)))]))
            ys = sorted(set([0, H] + [int(v) for v in rng.integers(1, H, int(rng.integers(0, 4)))]))
            for y0, y1 in zip(ys[:-1], ys[1:]):
                for x0, x1 in zip(xs[:-1], xs[1:]):
                    tr.renderTaskAsync(pt.RenderTask(x1 - x0, y1 - y0, x0, y0))
        elif cover == 2:
            band, world = 2 * int(rng.integers(1, 9)), int(rng.integers(1, 6))
            for r in range(world):
                tr.renderBandsAsync(band, world, r)
        else:                                                 # the dynamic dealer's launches: arbitrary band lists
            band, world = 2 * int(rng.integers(1, 9)), int(rng.integers(1, 6))
            n_bands = (H + band - 1) // band
            for l in pt.lpt_assignment([int(v) for v in rng.integers(1, 100, n_bands)], world):
                tr.renderBandListAsync(band, l)
        tr.waitForRenderTask()
        got = (fb.rgb_host().copy(), fb.yuv_host().copy(), tr.readCounters())
        tgp._assert_same(got, ref)
    tr.close()
    # statistic, not a gate: the opt-in SAH tree is not the reference's traversal
    if SAH_STATS is not None and legacy is None:
        fb2 = pt.Framebuffer(pt.Resolution(W, H))
        tr2 = pt.DevicePathTracer(0, spp, depth, (8, 8), scene, fb2, pt.CameraConfig(cam[0], cam[1], cam[2], cam[3]))
        tr2.setBvhMode(1)
        tr2.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr2.waitForRenderTask()
        one_rgb, one_yuv, one_cnt = fb2.rgb_host().copy(), fb2.yuv_host().copy(), None
        for _ in range(3):        # later launches run in cost order with the dearest tiles split over several waves:
            tr2.resetCounters()   # a gate -- whatever wave renders a pixel, the frame and the counters stay the same
            tr2.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr2.waitForRenderTask()
            cnt = tr2.readCounters()
            assert np.array_equal(fb2.rgb_host(), one_rgb) and np.array_equal(fb2.yuv_host(), one_yuv), "SAH frames differ between launches"
            assert one_cnt is None or cnt == one_cnt, "SAH counters differ between launches"
            one_cnt = cnt
        diff = int((fb2.rgb_host() != ref[0]).sum())
        SAH_STATS["cases"] += 1; SAH_STATS["bytes"] += ref[0].size
        SAH_STATS["cases_differ"] += diff > 0; SAH_STATS["bytes_differ"] += diff
        tr2.close()
    return W, H, spp, depth, mode, cover, frames, len(tris)


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    t0 = time.time()
    for seed in range(first, first + count):
        try:
            desc = one(seed)
        except Exception as e:          # noqa: BLE001
            print("FAIL seed", seed, repr(e)[:400], flush=True)
            sys.exit(1)
        if seed % 10 == 0:
            print("seed %d ok %s  (%.0f s)" % (seed, desc, time.time() - t0), flush=True)
    print("all %d cases bit-exact" % count)
    if SAH_STATS is not None:
        print("SAH tree vs oracle images:", SAH_STATS)


if __name__ == "__main__":
    main()
