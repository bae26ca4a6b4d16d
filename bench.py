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
# VALU lane-instruction peak: 256 CUs x 4 SIMD-32 x 32 lanes/clk x 2.4 GHz (a wave64 VALU instruction holds its SIMD
# for 2 cycles, MI355X_MICROARCH.md "Wave scheduling").  Every VALU instruction of these kernels is ONE IEEE op
# (no FMA contraction by contract), so this is also the op rate.
VALU_PEAK_LANE_INSTR_S = 256 * 4 * 32 * 2.4e9
PARITY_NOTE = ("bit-identical (bytes + work counters) to this repo's CPU restatement of the reference integrator; "
               "parity with the CUDA renderer itself is unpinned (the reference ships no vectors and cannot be built here)")

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


def kernel_build_id():
    """sha256 (16 hex digits) over the sources every render kernel is compiled from + the compiler flags: a PMC
    profile only describes the kernels of the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "multi-gpu-path-tracer_amd", "csrc")
    for name in ("pt_device.hip", "pt_walk.h", "pt_shade.h", "pt_math.h", "pt_kernels.h", "pt_internal.h", "Makefile"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    variant = os.environ.get("PTAMD_LIBRARY")
    if variant:                 # a diagnostic variant build (tools/*.sh): never the same build as a committed profile's
        with open(variant, "rb") as f:
            h.update(b"variant:" + hashlib.sha256(f.read()).digest())
    return h.hexdigest()[:16]


def matching_profile(config, workload, build_id, frames_in_flight=None):
    """The committed rocprofv3 PMC summary (profiles/*_<config>_pmc_summary.json; FETCH_SIZE / WRITE_SIZE / SQ
    counters collected in separate --pmc passes, corrected as MI355X_MICROARCH.md prescribes) of this exact
    workload string (scene, frame size, spp, depth, tree).  Returns (summary dict, path, same_build) or
    (None, None, False).  The work of a frame is deterministic (per-pixel seeds), so the per-launch counts of a
    profile of the same build hold for every launch of this run; what this run measures itself is the time."""
    import glob
    # by name: the round / step tag leads the file name (r02m_... < r03a_...); mtimes are arbitrary after a checkout
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_%s_pmc_summary.json" % config)))
    best = (None, None, False)
    for path in files:
        try:
            with open(path) as f:
                d = json.load(f)
            line = d["bench_line_under_kernel_trace"]
            if line["config"]["workload"] != workload:
                continue
            # a launch that shares the device with another one is a different launch (issue priority off, another split rule):
            # its per-launch counts only describe runs with the same number of frames in flight
            if frames_in_flight is not None and int(line.get("frames_in_flight", 1)) != int(frames_in_flight):
                continue
            same = line.get("kernel_build_id") == build_id
            if same or best[0] is None or not best[2]:
                best = (d, os.path.relpath(path, ROOT), same)
        except Exception:
            continue
    return best


def op_table():
    """profiles/*_op_table.json (tools/op_table.py): VALU instructions of one slab test, one Moller-Trumbore test, one light-pdf
    test, one scattering bounce and one path start with all lanes live, counted in the ISA of this build's own functions."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_op_table.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def algorithmic_ops(cnt, launches):
    """Per launch: what the reference's source asks for, in instructions of this build with every lane live -- the tests the
    kernel counted times the table's per-test instructions, a scattering bounce for every segment that is not a path's last,
    a path start per path.  Votes, merges, the walks' address arithmetic, the SAH forms' decode / sort / push and idle lanes
    are NOT in it: executed - algorithmic is the overhead."""
    t, path = op_table()
    if t is None:
        return None
    n = max(launches, 1)
    scattering = max(cnt["segments"] - cnt["paths"], 0)
    ops = (cnt["box_tests"] * t["per_box_test"] + cnt["tri_tests"] * t["per_triangle_test"] + cnt["light_tests"] * t["per_light_test"]
           + scattering * t["per_scattering_segment"] + cnt["paths"] * t["per_path_start"]) / n
    return {"ops_per_launch": ops, "ops_per_segment": ops * n / max(cnt["segments"], 1), "table": path,
            "per_test": {k: t[k] for k in ("per_box_test", "per_triangle_test", "per_light_test", "per_scattering_segment", "per_path_start")},
            "table_build_id": t.get("kernel_build_id")}


def roofline_block(config, workload, cnt, pixels, kernel_ms, launches, kernel_name, device_ms_per_launch=None, frames_in_flight=1):
    """The bound that binds, with the time measured in THIS run (HIP events on the launch stream, pt_kernel_time_read).
    device_ms_per_launch: frames in flight -- the launches overlap, so the time that goes with one launch's counts is the
    device time all timed launches took together (first start to last end, HIP events on their streams) / their number."""
    build_id = kernel_build_id()
    prof, path, same_build = matching_profile(config, workload, build_id, frames_in_flight)
    per_launch_s = (device_ms_per_launch if device_ms_per_launch is not None else kernel_ms / max(launches, 1)) * 1e-3
    alg = algorithmic_bytes(cnt, pixels) / max(launches, 1)
    hbm_bound = config == "c4"         # the only scene (1 GB) that does not live in the scalar cache / L1 / L2
    traffic = lane_instr = None
    if prof is not None:
        if "hbm_bytes_per_launch" in prof:
            traffic = float(prof["hbm_bytes_per_launch"]["total"])
        c = prof.get("counters", {})
        if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c and "SQ_INSTS_VALU" in c:
            # executed VALU lane-instructions of one launch = wave-instructions x 64 x lane utilisation, the convention of
            # tools/make_profile_summary.py (so the line and its profile agree to the digit)
            util = float(c["SQ_THREAD_CYCLES_VALU"]) / (float(c["SQ_ACTIVE_INST_VALU"]) * 64.0)
            lane_instr = float(c["SQ_INSTS_VALU"]) * 64.0 * util
    out = {"bound": "hbm" if hbm_bound else "valu-issue"}
    if hbm_bound:
        achieved = traffic / per_launch_s / 1e9 if (traffic and per_launch_s > 0) else None
        out.update({"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS if achieved is not None else None})
    else:
        achieved = lane_instr / per_launch_s if (lane_instr and per_launch_s > 0 and same_build) else None
        out.update({"achieved": achieved, "peak": VALU_PEAK_LANE_INSTR_S, "unit": "lane-instructions/s",
                    "frac": achieved / VALU_PEAK_LANE_INSTR_S if achieved is not None else None})
    out.update({
        "traffic": traffic,
        "source": ("per-launch counts (HBM bytes = FETCH_SIZE x2 + WRITE_SIZE; VALU lane-instructions = SQ_THREAD_CYCLES_VALU) "
                   "from %s, %s; kernel time measured in this run with HIP events on the launch stream"
                   % (path, "taken on this build of the kernels" if same_build else
                      "taken on ANOTHER build of the kernels (VALU figure withheld, HBM bytes indicative)"))
                  if prof is not None else "no committed PMC profile of this workload: only the time is measured",
        "kernel_build_id": build_id,
        "kernel": kernel_name, "kernel_ms_avg": kernel_ms / max(launches, 1), "launches": launches,
        "device_ms_per_launch": device_ms_per_launch if device_ms_per_launch is not None else kernel_ms / max(launches, 1),
        # SURVEY 8(d)'s byte model, kept as a secondary figure: on the KB-sized Cornell scenes these bytes are served by
        # SGPRs / the scalar cache / LDS, never by HBM, so they are NOT divided by the HBM peak
        "algorithmic_hbm_equiv": {
            "bytes_per_launch": alg, "bytes_per_segment": alg * max(launches, 1) / max(cnt["segments"], 1),
            "GB_per_s": alg / per_launch_s / 1e9 if per_launch_s > 0 else None,
            "note": "32*n_box + 40*n_tri + 40*n_light + 64*segments + 4.5 B/pixel from the kernel's own counters; an "
                    "operand-delivery rate (registers/caches), not HBM traffic"},
        "note": ROOFLINE_NOTES.get(config, ROOFLINE_NOTES["c2"]),
    })
    if hbm_bound and lane_instr and same_build and per_launch_s > 0:
        out["valu_issue_frac"] = lane_instr / per_launch_s / VALU_PEAK_LANE_INSTR_S
    alg_ops = algorithmic_ops(cnt, launches)
    if alg_ops is not None and per_launch_s > 0:
        # the ALGORITHMIC numerator (SURVEY 8d asks for algorithmic work per unit): a kernel that executes
        # more instructions for the same segments raises `frac` and lowers nothing here
        out["algorithmic_ops_per_launch"] = alg_ops["ops_per_launch"]
        out["algorithmic_ops_per_segment"] = alg_ops["ops_per_segment"]
        out["algorithmic_frac"] = alg_ops["ops_per_launch"] / per_launch_s / VALU_PEAK_LANE_INSTR_S
        out["algorithmic_source"] = ("%s (VALU instructions per test with all lanes live, from the ISA of this build's own functions: %s) x the "
                                     "kernel's exact counters; over the same time and peak as frac" % (alg_ops["table"], alg_ops["per_test"]))
        if lane_instr and same_build:
            out["executed_over_algorithmic"] = lane_instr / alg_ops["ops_per_launch"]
    return out


def load_scene(pt, kind):
    if kind == "box":
        sc = pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_box.glb")).applyCornellConvention()
        return sc, BOX_CAMERA
    if kind == "duck":
        return pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_duck.glb")), DUCK_CAMERA
    return pt.make_terrain_scene(2048), TERRAIN_CAMERA


def cpu_baseline(scene, camera, W, H, spp, depth, budget_rows, kind="box"):
    """The oracle (CPU restatement of the reference integrator) timed on this host's cores on
    a bounded sample of the SAME workload: a centred strip (or, for very dear scenes, patch) of the frame."""
    from oracle import binding as ob
    host_cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))      # the cores this process may be scheduled on ...
    except Exception:
        cores = host_cores
    quota = None
    try:                                          # ... and the CPU time its cgroup grants it (a GPU box gives a share of the host:
        with open("/sys/fs/cgroup/cpu.max") as f:  # more runnable threads than that only fight over it)
            q, period = f.read().split()[:2]
        if q != "max":
            quota = max(1, int(round(int(q) / int(period))))
    except Exception:
        pass
    if quota is None:
        try:                                      # cgroup v1
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if q > 0:
                quota = max(1, int(round(q / period)))
        except Exception:
            pass
    if quota is not None:
        cores = min(cores, quota)
    osc = ob.OracleScene(scene.triangles, scene.materials, scene.textures)
    cam = (camera[0], camera[1], 45.0, 45.0)
    if quota is None and cores > 16:
        # no quota to read, many visible cores: a box that grants this process fewer CPUs than it shows runs the oracle
        # SLOWER with a thread per visible core.  Probe 16 threads (a one-GPU box's share) against all of them on a
        # small patch and keep the faster; `cores` below is what the reported sample really used.
        def rate(nthreads):
            ph = min(H, max(nthreads, 64))
            t0 = time.perf_counter()
            _, _, c = osc.render(cam, W, H, max(1, min(spp, 8)), depth, task=(min(W, 256), ph, 0, (H - ph) // 2), threads=nthreads)
            return c["segments"] / max(time.perf_counter() - t0, 1e-9)
        if kind != "terrain" and rate(16) > rate(cores):
            cores = 16
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
        "value": cnt["segments"] / dt, "unit": "ray-segments/s", "cores": cores, "host_cpu_count": host_cores, "cpu_quota_cores": quota, "kind": "port",
        "sample": "task %dx%d at offset (%d, %d) of the %dx%d frame (%.3g%% of pixels), %d spp, %d bounces, %.1f s"
                  % (sw, rows, ox, oy, W, H, 100.0 * sw * rows / (W * H), spp, depth, dt),
    }


ROOFLINE_NOTES = {
    "c2": "2.4 KB scene held in SGPRs / the scalar cache: the kernel is bound by VALU issue (every op a single IEEE "
          "instruction); `traffic` = HBM bytes of a launch (framebuffer + spills), a rounding error against 8 TB/s",
    "c3": "0.7 MB tree + 3 MB texture live in L1/L2: bound by VALU issue / divergence; `traffic` = HBM bytes of a launch",
    "c4": "1 GB scene: `traffic` (PMC FETCH_SIZE x2 + WRITE_SIZE, an upper bound) is what leaves the L2s (Infinity-Cache "
          "hits are counted in it, MI355X_MICROARCH.md).  The walk is a per-lane gather of 64-byte nodes and 48-byte triangles: "
          "what it saturates first is the vector-memory REQUEST path (texture-address units ~70 % busy at one cycle per lane "
          "and 16-byte request, profiles/r03_sah_lane_census.json), not HBM bytes, so frac is low by construction",
}
ROOFLINE_NOTES["c1"] = ROOFLINE_NOTES["c5"] = ROOFLINE_NOTES["c2"]


def native_host_leg(args):
    """bench.py --host native: the reference-shaped host (RenderManager.h:76-112: ONE process, a StreamThread per GPU and
    stream; here csrc/host/*.h -> pt_render) as a child process, its own JSON line relayed with the keys of the Python leg.
    The frame = RenderManager::renderFrame incl. pt_gather_frame (RCCL ncclSend / ncclRecv to GPU 0, or peer copies) and the
    copy to the host; counters are the engines' own."""
    import subprocess
    kind, W, H, spp, depth, desc = CONFIGS[args.config]
    if args.width or args.height:
        W, H = args.width or W, args.height or H
    if args.spp:
        spp = args.spp
    exe = os.path.join(ROOT, "multi-gpu-path-tracer_amd", "pt_render")
    if not os.path.exists(exe):
        raise SystemExit("bench.py --host native needs %s (make -C multi-gpu-path-tracer_amd/csrc all)" % exe)
    if kind == "terrain":
        raise SystemExit("bench.py --host native: the terrain is generated in the library, pt_render loads files (c1 / c2 / c3 / c5)")
    model = os.path.join(ROOT, "models", "cornell_box.glb" if kind == "box" else "cornell_duck.glb")
    cam = BOX_CAMERA if kind == "box" else DUCK_CAMERA
    bvh = args.bvh or "reference"
    fif = args.frames_in_flight or (4 if args.gpus >= 8 else 3 if args.gpus > 1 else 2)
    cmd = [exe, "bench", model, "--width", str(W), "--height", str(H), "--spp", str(spp), "--depth", str(depth),
           "--gpus", str(args.gpus), "--frames", str(args.steps), "--warmup", str(args.warmup), "--frames-in-flight", str(fif),
           "--lookfrom", "%g,%g,%g" % cam[0], "--front", "%g,%g,%g" % cam[1], "--json", "--out", ""]
    if kind == "box":
        cmd.append("--cornell")
    if bvh == "sah":
        cmd += ["--bvh", "sah"]
    if args.rehearse:
        cmd.append("--same-device")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    sys.stderr.write(r.stderr)
    line = None
    for l in r.stdout.splitlines():
        if l.startswith("{"):
            line = json.loads(l)
    if r.returncode != 0 or line is None:
        sys.stderr.write(r.stdout)
        return r.returncode or 1
    seg = line["segments"]
    out = {
        "metric": "ray-segments/s", "value": seg / (line["ms_total"] * 1e-3), "unit": "ray-segments/s",
        "n_gpus": args.gpus, "ranks": line["devices"], "backend": line["gather_transport"] or "none",
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": line["ms_total"] / max(args.steps, 1),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "host": "native (pt_render: one process, %d worker thread(s), csrc/host/RenderManager.h)" % line["workers"],
        "config": {"workload": "%s [bvh=%s]" % (desc, bvh), "scene": kind, "width": W, "height": H, "spp": spp, "bounces": depth,
                   "parity": PARITY_NOTE,
                   "parallelism": "one task per (GPU, stream) (%s), scene replicated, pt_gather_frame to GPU 0 + one copy to the host, %d frame%s in flight per GPU"
                                  % (line["algorithm"], line["frames_in_flight"], "" if line["frames_in_flight"] == 1 else "s")},
        "frames_in_flight": line["frames_in_flight"],
        "frame_delivery": "host-visible: the shared host Framebuffer holds every completed frame (RenderManager::renderFrame)",
        "segments_per_frame": seg // max(args.steps, 1),
        "segments_per_path": seg / max(line["paths"], 1),
        "tests_per_segment": {"box_tests": line["box_tests"] / max(seg, 1), "tri_tests": line["tri_tests"] / max(seg, 1), "light_tests": line["light_tests"] / max(seg, 1)},
        "kernel_build_id": kernel_build_id(),
    }
    print(json.dumps(out), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)      # (frames in flight: the pipeline is empty at both ends of the timed
    ap.add_argument("--warmup", type=int, default=3)      #  region, so a handful of frames would mostly measure its filling)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--band-rows", type=int, default=8)
    ap.add_argument("--algo", default=None, choices=["fsfl", "dsfl"],
                    help="N > 1: how row bands are dealt to ranks -- fsfl: band b to rank b %% N (fixed split); dsfl: re-dealt "
                         "every frame from the previous frame's per-band costs, longest first (default for c5, whose "
                         "BASELINE config names a dynamic load-balancer; fsfl otherwise)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-visible", action="store_true", help="skip the extra leg that times one frame at a time (a frame's latency)")
    ap.add_argument("--device-frames", action="store_true",
                    help="diagnostic: leave the frames on the device (no copy to the host inside the timed region): the kernel-only "
                         "figure of rounds 1-3; the default delivers every frame host-visible, as SURVEY 8(d) defines a frame")
    ap.add_argument("--host", default="python", choices=["python", "native"],
                    help="python: one process per GPU over torch.distributed / RCCL (the default, what the driver launches); native: "
                         "the C++ host mirror (pt_render: ONE process, a worker thread per GPU and stream, pt_gather_frame over RCCL "
                         "or peer copies), started as a child process, its line relayed with the same keys")
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
    ap.add_argument("--walk", type=int, default=0, choices=[0, 1, 2, 3, 4, 5],
                    help="diagnostic: pt_set_traversal_mode (0 auto, 1 per-lane, 2 wave-uniform mask, 3 wave-uniform stack, "
                         "4 mask walk with dealt sparse leaves, 5 stack walk with dealt sparse leaves)")
    ap.add_argument("--rehearse", action="store_true",
                    help="multi-rank rehearsal on a box with ONE GPU: every rank uses cuda:0 and the exchange runs "
                         "over gloo (RCCL needs one GPU per rank); checks the N > 1 code path, not its speed")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6],
                    help="frames rendered concurrently, each in its own framebuffer on its own stream: the next frame's waves "
                         "take the wave slots the current frame's last chains leave idle.  0 = the default (2 on one GPU, 3 from two ranks "
                         "on, 4 from eight; 1 with the cost-fed dealers)")
    ap.add_argument("--share-of", type=int, default=0,
                    help="diagnostic (1 GPU): time only rank 0's row bands of an N-rank split, print a "
                         "diagnostic line instead of the metric line")
    args = ap.parse_args()

    # More than three launches only overlap on the device when each has a hardware queue of its own: the HIP runtime maps
    # streams onto 4 queues by default, and a launch behind another stream's persistent kernel on a shared queue waits for
    # it (an eighth of C2 with 4 frames in flight: 9.9 ms per frame on 4 queues, 7.8 on 8).  Read when the runtime
    # initialises: set before torch is imported, and handed on to the ranks this process may start.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    if args.host == "native":
        raise SystemExit(native_host_leg(args))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher (`python bench.py --gpus N`): this process becomes the launcher.  It has not
        # imported torch nor touched the GPU, and it starts the N ranks as a CHILD process (never exec: see the
        # environment's rule on replacing a process that may have initialised the GPU), relays rank 0's JSON line and
        # leaves with the child's exit code.  The reference is one process driving every GPU (RenderManager.h:76-112);
        # here that process is torch.distributed.run and the per-GPU threads are ranks.
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import torch
    import torch.distributed as dist
    pt = importlib.import_module("multi-gpu-path-tracer_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node == --gpus, "
                         "or without a launcher (bench.py then starts the ranks itself)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is mandatory, there is no CPU fallback")
    if args.rehearse:
        local_rank = 0
    if not args.rehearse and world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks but %d visible GPUs (RCCL needs one GPU per rank; --rehearse runs the "
                         "N-rank flow on one GPU over gloo)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    kind, W, H, spp, depth, desc = CONFIGS[args.config]
    if args.width or args.height:
        W, H = args.width or W, args.height or H
        desc += " [diagnostic size %dx%d]" % (W, H)
    if args.spp:
        spp = args.spp
        desc += " [diagnostic spp %d]" % spp
    scene, camera = load_scene(pt, kind)
    algo = args.algo or ("dsfl" if args.config == "c5" else "fsfl")
    cfg = pt.RendererConfig(samplesPerPixel=spp, recursionDepth=depth, gpuNumber=world,
                            resolution=pt.Resolution(W, H), showTasks=False,
                            algorithmType=pt.DSFL if algo == "dsfl" else pt.FSFL)
    cam = pt.CameraConfig(camera[0], camera[1])
    bvh = args.bvh or ("sah" if args.config == "c4" else "reference")
    desc += " [bvh=%s]" % bvh
    t_up = time.perf_counter()
    # Frames in flight (scheduling.py RenderManager): one GPU keeps two frames queued, the ranks of a multi-GPU run three
    # (four from eight ranks on, where a share is an eighth of a frame and its dearest chains leave the device idle for half
    # of its time; the third changes nothing for the box but much for the latency-bound SAH kernels), each in its own
    # framebuffer on its own stream: the next frame's waves take the wave slots the current frame's last chains leave empty
    # (one GPU, C2: 59.6 -> 55.6 ms per frame; an eighth of C2: 13.7 -> 7.8 ms; profiles/r03_frames_in_flight.json).  The
    # timed region starts and ends with an empty pipeline.  The cost-fed dealers keep one frame in flight.
    # (the cost-fed dealers pipeline too since round 4: every frame in flight carries its own band assignment)
    fif = args.frames_in_flight or (4 if world >= 8 else 3 if world > 1 else 2)
    mgr = pt.RenderManager(cfg, scene, cam, device_idx=local_rank, band_rows=args.band_rows,
                           bvh_mode=1 if bvh == "sah" else 0, frames_in_flight=fif, host_frames=not args.device_frames)
    fif = mgr.frames_in_flight
    upload_s = time.perf_counter() - t_up
    mgr.tracer.setTileSchedule(args.tile_schedule)
    if args.walk:
        mgr.tracer.setTraversalMode(args.walk)
        desc += " [diagnostic walk %d]" % args.walk

    if args.share_of > 1 and world == 1:
        # what one rank of an N-GPU run computes, measured on this GPU alone (no gather)
        tr = mgr.tracer
        F = max(1, args.frames_in_flight)
        fbs = [mgr.getFramebuffer()] + [pt.Framebuffer(pt.Resolution(W, H), device="cuda:%d" % local_rank) for _ in range(F - 1)]
        streams = [torch.cuda.Stream(device=local_rank) for _ in range(F)]

        def frames(n):          # n shares, at most F of them in flight: share i waits for share i - F (its framebuffer, its stream)
            for i in range(n):
                tr.synchronizeStream(streams[i % F])
                if F > 1:
                    tr.setFramebuffer(fbs[i % F])
                tr.renderBandsAsync(args.band_rows, args.share_of, 0, streams[i % F])
            for st in streams:
                tr.synchronizeStream(st)
        frames(max(args.warmup, F))
        tr.resetCounters(); tr.resetKernelTime()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames(args.steps)
        dt = (time.perf_counter() - t0) / args.steps
        kms, n = tr.kernelTime()
        print(json.dumps({"diagnostic": "share_of", "n_ranks": args.share_of, "frames_in_flight": F, "ms_per_share": dt * 1e3,
                          "kernel_ms": kms / n, "segments_per_share": tr.readCounters()["segments"] // args.steps}))
        return

    def sync_all():
        mgr.flush()                    # frames in flight: every launched frame rendered, exchanged and in place on rank 0
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        mgr.renderFrame()
    sync_all()
    mgr.tracer.resetCounters()
    mgr.tracer.resetKernelTime()
    mgr.record_spans = True
    mgr.resetSpans()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mgr.renderFrame()
    sync_all()
    elapsed = time.perf_counter() - t0
    mgr.record_spans = False
    spans = mgr.launchSpans()          # HIP events on the launches' streams: [(start, end)] relative to the first start
    device_ms_per_launch = ((max(b for _, b in spans) - min(a for a, _ in spans)) / len(spans)) if spans else None

    # max over ranks of the timed region; sum over ranks of the exact work counters
    cnt = mgr.tracer.readCounters()
    kernel_ms, launches = mgr.tracer.kernelTime()
    try:
        split_tiles = mgr.tracer.splitTiles()
    except Exception:
        split_tiles = None
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
    if world > 1:
        # the launch that bounds the frame is the slowest rank's; bytes and pixels are the whole frame's
        kmax = torch.tensor([kernel_ms], dtype=torch.float64, device=rdev)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        kernel_ms = float(kmax.item())

    # SURVEY 8(d) defines the frame as "incl. gather to host-visible RGB" (RenderManager.h:410-431 ends on a barrier and the
    # host reads the managed framebuffer, Framebuffer.h:27-35).  The timed region above IS that frame since round 4: every
    # completed frame is copied, RGB + YUV, into pinned host memory on a stream of its own under the rendering of the next
    # (RenderManager(host_frames=True)), and the region ends when the last copy has landed.  What it cannot show is a single
    # frame's LATENCY: this leg renders one frame at a time, each delivered to the host before the next starts.
    latency_ms = None
    if not args.no_host_visible:
        n_hv = max(1, min(args.steps, 5))
        sync_all()
        t0 = time.perf_counter()
        for _ in range(n_hv):
            mgr.renderFrame()
            mgr.flush()                                # one frame at a time: rendered, exchanged, copied (flush waits for the copy)
            if rank == 0 and not args.device_frames:
                assert mgr.hostFrame() is not None
        sync_all()
        latency_ms = (time.perf_counter() - t0) * 1e3 / n_hv

    if rank == 0:
        if args.ppm:
            pt.write_ppm(args.ppm, mgr.getCurrentFrame())
        seg_per_s = total["segments"] / elapsed
        # roofline of the dominant (only) kernel: one launch per rank and frame
        out = {
            "metric": "ray-segments/s", "value": seg_per_s, "unit": "ray-segments/s",
            "n_gpus": world, "ranks": dist.get_world_size() if world > 1 else 1,
            "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else " (rehearsal on one GPU)")) if world > 1 else "none",
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "scene": kind, "width": W, "height": H, "spp": spp, "bounces": depth,
                       "parity": PARITY_NOTE,
                       "tile_schedule": "cost order from the previous frame" if args.tile_schedule else "raster order",
                       "parallelism": "row-bands x%d (band_rows=%d, dealer=%s), scene replicated, true-size sends to rank 0, %d frame%s in flight per GPU"
                                      % (world, args.band_rows, algo if world > 1 else "n/a", fif, "" if fif == 1 else "s")},
            "frames_in_flight": fif,
            "frame_delivery": ("device framebuffer only (diagnostic)" if args.device_frames else
                               "host-visible: RGB + YUV of every frame copied into pinned host memory on rank 0, overlapped with the next frame's rendering; "
                               "the timed region ends when the last copy has landed (SURVEY 8d's frame)"),
            "segments_per_frame": total["segments"] // args.steps,
            "scene_upload_s": upload_s,
            "segments_per_path": total["segments"] / max(total["paths"], 1),
            "tests_per_segment": {k: total[k] / max(total["segments"], 1) for k in ("box_tests", "tri_tests", "light_tests")},
            "kernel_build_id": kernel_build_id(),
            "split_tiles_next_launch": split_tiles,      # tiles the schedule hands to several waves each (0: none is critical)
            # N > 1: whole-frame counters over the slowest rank's kernel time (per-launch profile counts are those of a
            # whole-frame launch and do not apply to a 1/N share, so `achieved` stays null there)
            "roofline": roofline_block(args.config, desc if world == 1 else desc + " [1/%d shares]" % world, total,
                                       W * H * args.steps, kernel_ms, launches, mgr.tracer.kernelName(),
                                       device_ms_per_launch if fif > 1 else None, fif),
        }
        if latency_ms is not None:
            out["ms_per_frame_latency"] = latency_ms               # one frame at a time, each delivered to the host before the next starts
            out["ms_per_step_host_visible"] = latency_ms           # (the key of rounds 2-3 for the same quantity)
        if fif > 1:
            out["frames_in_flight_note"] = ("ms_per_step = the timed region (K frames launched, rendered, exchanged, delivered to the host; pipeline "
                                            "empty at both ends) / K with %d frames queued per GPU: a frame RATE; ms_per_frame_latency renders one "
                                            "frame at a time, each host-visible before the next starts: a frame's LATENCY" % fif)
            out["roofline"]["kernel_ms_note"] = ("launches overlap (%d frames in flight): kernel_ms_avg is a launch's own duration, device_ms_per_launch = "
                                                 "(last end - first start of the timed launches, HIP events on their streams) / launches is what one "
                                                 "launch's counts are divided by" % fif)
            out["launch_spans_ms"] = [[round(a, 3), round(b, 3)] for a, b in spans]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, camera, W, H, spp, depth, args.cpu_rows, kind)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
