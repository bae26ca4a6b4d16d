#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json): ray-segments/s and
ms/frame on Cornell box 1920x1080, 256 spp, 8 bounces, on 1..N MI355X.

A "step" is one frame = RenderManager.renderFrame(): every rank renders its row bands of the
frame (scene replicated per GPU) and the bands are gathered to rank 0 (the path's one
exchange step).  Inputs (scene, BVH) are resident in HBM before the timed region.  The
image is deterministic (per-pixel seeds), so every frame does identical work.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|c3|c4]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see README / DESIGN.md "Measurement").
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)

CONFIGS = {
    # name: (scene, W, H, spp, depth, description)
    "c1": ("box", 400, 400, 16, 4, "cornell_box.glb 400x400 16spp 4bounce"),
    "c2": ("box", 1920, 1080, 256, 8, "cornell_box.glb 1920x1080 256spp 8bounce"),
    "c3": ("duck", 1920, 1080, 256, 8, "cornell_duck.glb 1920x1080 256spp 8bounce"),
    "c4": ("terrain", 1920, 1080, 64, 8, "8M-triangle synthetic terrain 1920x1080 64spp 8bounce"),
    "c5": ("box", 3840, 2160, 1024, 16, "cornell_box.glb 3840x2160 1024spp 16bounce"),
}
BOX_CAMERA = ((-255.5, 252.5, 700.0), (0.0, 0.0, -1.0))      # SURVEY.md 8d benchmark convention
DUCK_CAMERA = ((0.0, 0.0, 0.5), (0.0, 0.0, -0.5))            # main.cu:40
TERRAIN_CAMERA = ((0.0, 220.0, 620.0), (0.0, -0.35, -1.0))


def algorithmic_bytes(cnt, pixels):
    """B = 32*n_box + 40*n_tri + 40*n_light + 64*segments + 4.5 B/pixel."""
    return (32 * cnt["box_tests"] + 40 * cnt["tri_tests"] + 40 * cnt["light_tests"] + 64 * cnt["segments"]
            + 4.5 * pixels)


def measured_traffic(config, workload=None):
    """HBM bytes per launch of the render kernel from the committed rocprofv3 PMC passes
    (profiles/*_<config>_pmc_summary.json, FETCH_SIZE / WRITE_SIZE collected in separate passes and
    corrected as MI355X_MICROARCH.md prescribes); None when no profile of this config AND the same
    workload string (frame size, spp, depth, tree) is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_pmc_summary.json" % config)))
    for path in reversed(files):
        try:
            with open(path) as f:
                d = json.load(f)
            if workload is not None and d["bench_line_under_kernel_trace"]["config"]["workload"] != workload:
                continue
            return float(d["hbm_bytes_per_launch"]["total"]), os.path.relpath(path, ROOT), d.get("valu_roofline")
        except Exception:
            continue
    return None, None, None


def load_scene(pt, kind):
    if kind == "box":
        sc = pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_box.glb")).applyCornellConvention()
        return sc, BOX_CAMERA
    if kind == "duck":
        return pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_duck.glb")), DUCK_CAMERA
    return pt.make_terrain_scene(2048), TERRAIN_CAMERA


def cpu_baseline(scene, camera, W, H, spp, depth, budget_rows):
    """The oracle (CPU restatement of the reference integrator) timed on this host's cores on
    a bounded sample of the SAME workload: a centred strip (or, for very dear scenes, patch) of the frame."""
    from oracle import binding as ob
    cores = min(os.cpu_count() or 1, 16)
    osc = ob.OracleScene(scene.triangles, scene.materials, scene.textures)
    cam = (camera[0], camera[1], 45.0, 45.0)
    # The sample grows until it is long enough to be a fair timing and then stops, whatever the scene costs per
    # pixel (the 8 M-triangle terrain with the reference's tree is ~1e6 times dearer per pixel than the box): a
    # centred patch one row per thread high, doubled in width up to the full width, then doubled in height up to
    # budget_rows; the first run of >= 5 s (or the biggest allowed) is the one reported.  Short runs are not
    # extrapolated from: the oracle's thread pool takes a few tenths of a second of work to reach its rate.
    sw, rows = 4, cores
    while True:
        ox, oy = (W - sw) // 2, (H - rows) // 2
        t0 = time.perf_counter()
        _, _, cnt = osc.render(cam, W, H, spp, depth, task=(sw, rows, ox, oy), threads=cores)
        dt = time.perf_counter() - t0
        if dt >= 5.0:
            break
        if sw < W:
            sw = min(W, sw * (2 if dt > 1.0 else 4))
        elif rows < min(H, budget_rows):
            rows = min(H, budget_rows, rows * (2 if dt > 1.0 else 4))
        else:
            break
    return {
        "value": cnt["segments"] / dt, "unit": "ray-segments/s", "cores": cores, "kind": "port",
        "sample": "task %dx%d at offset (%d, %d) of the %dx%d frame (%.3g%% of pixels), %d spp, %d bounces, %.1f s"
                  % (sw, rows, ox, oy, W, H, 100.0 * sw * rows / (W * H), spp, depth, dt),
    }


ROOFLINE_NOTES = {
    "c2": "algorithmic bytes (SURVEY 8d) are served by the scalar cache / L1 on this KB-sized scene; "
          "the HBM traffic of a launch is `traffic` bytes, the kernel is VALU-issue bound (DESIGN.md 4)",
    "c3": "algorithmic bytes (SURVEY 8d) are served by L1/L2 (0.7 MB of tree + 3 MB texture); the HBM traffic of "
          "a launch is `traffic` bytes; the SAH kernel is issue/divergence bound (DESIGN.md 4)",
    "c4": "1 GB scene: `traffic` (PMC FETCH_SIZE x2 + WRITE_SIZE) is what leaves the L2s (Infinity-Cache hits are "
          "counted in it, MI355X_MICROARCH.md); the SAH kernel is memory-latency bound here (one dependent node or "
          "triangle fetch per micro-step, DESIGN.md 4)",
}
ROOFLINE_NOTES["c1"] = ROOFLINE_NOTES["c5"] = ROOFLINE_NOTES["c2"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--band-rows", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--tile-schedule", type=int, default=1, choices=[0, 1],
                    help="0: tiles in raster order; 1: ordered by their cost in the previous frame (pt_set_tile_schedule)")
    ap.add_argument("--cpu-rows", type=int, default=360)
    ap.add_argument("--ppm", default=None, help="write the gathered frame of the last step here")
    ap.add_argument("--bvh", default=None, choices=["reference", "sah"],
                    help="acceleration structure: the reference's topology (default; results identical to the "
                         "reference traversal) or the opt-in SAH tree (default for c4, where the reference's tree "
                         "makes every ray test ~730k triangles)")
    ap.add_argument("--width", type=int, default=0, help="diagnostic: override the config's width")
    ap.add_argument("--height", type=int, default=0, help="diagnostic: override the config's height")
    ap.add_argument("--spp", type=int, default=0, help="diagnostic: override the config's samples per pixel")
    ap.add_argument("--walk", type=int, default=0, choices=[0, 1, 2, 3],
                    help="diagnostic: pt_set_traversal_mode (0 auto, 1 per-lane, 2 wave-uniform mask, 3 wave-uniform stack)")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on a box with ONE GPU: every rank uses cuda:0 and the exchange runs "
                         "over gloo (RCCL needs one GPU per rank); checks the N > 1 code path, not its speed")
    ap.add_argument("--share-of", type=int, default=0,
                    help="diagnostic (1 GPU): time only rank 0's row bands of an N-rank split, print a "
                         "diagnostic line instead of the metric line")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    pt = importlib.import_module("multi-gpu-path-tracer_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is mandatory, there is no CPU fallback")
    if args.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node == --gpus"

    kind, W, H, spp, depth, desc = CONFIGS[args.config]
    if args.width or args.height:
        W, H = args.width or W, args.height or H
        desc += " [diagnostic size %dx%d]" % (W, H)
    if args.spp:
        spp = args.spp
        desc += " [diagnostic spp %d]" % spp
    scene, camera = load_scene(pt, kind)
    cfg = pt.RendererConfig(samplesPerPixel=spp, recursionDepth=depth, gpuNumber=world,
                            resolution=pt.Resolution(W, H), showTasks=False)
    cam = pt.CameraConfig(camera[0], camera[1])
    bvh = args.bvh or ("sah" if args.config == "c4" else "reference")
    desc += " [bvh=%s]" % bvh
    t_up = time.perf_counter()
    mgr = pt.RenderManager(cfg, scene, cam, device_idx=local_rank, band_rows=args.band_rows,
                           bvh_mode=1 if bvh == "sah" else 0)
    upload_s = time.perf_counter() - t_up
    mgr.tracer.setTileSchedule(args.tile_schedule)
    if args.walk:
        mgr.tracer.setTraversalMode(args.walk)
        desc += " [diagnostic walk %d]" % args.walk

    if args.share_of > 1 and world == 1:
        # what one rank of an N-GPU run computes, measured on this GPU alone (no gather)
        tr = mgr.tracer
        for _ in range(args.warmup):
            tr.renderBandsAsync(args.band_rows, args.share_of, 0); tr.waitForRenderTask()
        tr.resetCounters(); tr.resetKernelTime()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.renderBandsAsync(args.band_rows, args.share_of, 0); tr.waitForRenderTask()
        dt = (time.perf_counter() - t0) / args.steps
        kms, n = tr.kernelTime()
        print(json.dumps({"diagnostic": "share_of", "n_ranks": args.share_of, "ms_per_share": dt * 1e3,
                          "kernel_ms": kms / n, "segments_per_share": tr.readCounters()["segments"] // args.steps}))
        return

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        mgr.renderFrame()
    sync_all()
    mgr.tracer.resetCounters()
    mgr.tracer.resetKernelTime()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mgr.renderFrame()
    sync_all()
    elapsed = time.perf_counter() - t0

    # max over ranks of the timed region; sum over ranks of the exact work counters
    cnt = mgr.tracer.readCounters()
    kernel_ms, launches = mgr.tracer.kernelTime()
    keys = ["paths", "segments", "box_tests", "tri_tests", "light_tests"]
    if world > 1:
        rdev = "cpu" if args.rehearse else "cuda"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=rdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([float(cnt[k]) for k in keys], dtype=torch.float64, device=rdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total = {k: int(v) for k, v in zip(keys, tot.tolist())}
    else:
        total = cnt

    if rank == 0:
        if args.ppm:
            pt.write_ppm(args.ppm, mgr.getCurrentFrame())
        seg_per_s = total["segments"] / elapsed
        # roofline of the dominant (only) kernel, from rank 0's own launches
        my_pixels = sum(t.width * t.height for t in mgr.renderTasks_) * args.steps
        bytes_rank0 = algorithmic_bytes(cnt, my_pixels)
        achieved = bytes_rank0 / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic, traffic_src, valu = (None, None, None)
        if world == 1:                                   # matched on the workload string (size, spp, depth, tree)
            traffic, traffic_src, valu = measured_traffic(args.config, desc)
        out = {
            "metric": "ray-segments/s", "value": seg_per_s, "unit": "ray-segments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "scene": kind, "width": W, "height": H, "spp": spp, "bounces": depth,
                       "tile_schedule": "cost order from the previous frame" if args.tile_schedule else "raster order",
                       "parallelism": "row-bands x%d (band_rows=%d), scene replicated, gather to rank 0"
                                      % (world, args.band_rows)},
            "segments_per_frame": total["segments"] // args.steps,
            "scene_upload_s": upload_s,
            "segments_per_path": total["segments"] / max(total["paths"], 1),
            "tests_per_segment": {k: total[k] / max(total["segments"], 1) for k in ("box_tests", "tri_tests", "light_tests")},
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bytes_rank0 / max(launches, 1),
                "valu_roofline_from_profile": valu,
                "note": ROOFLINE_NOTES.get(args.config, ROOFLINE_NOTES["c2"]),
                "kernel": "pt::dev::render_kernel", "kernel_ms_avg": kernel_ms / max(launches, 1),
                "launches": launches,
                "algorithmic_bytes_per_segment": bytes_rank0 / max(cnt["segments"], 1),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, camera, W, H, spp, depth, args.cpu_rows)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
