"""GPU tests of the callers around the kernel: golden fixtures through the C ABI, the Python
RenderManager (single rank), the C++ RenderManager / StreamThread / Framebuffer mirror via the
`pt_render` .glb -> PPM driver, and size-independent properties at BASELINE.json's full size."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import BOX_CAMERA, DUCK_CAMERA, ROOT

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("name,scene", [("box_64x64_4spp_4b", "box"), ("box_160x90_8spp_8b", "box"),
                                        ("duck_64x36_2spp_4b", "duck"), ("duck_96x54_3spp_8b", "duck"),
                                        ("box_400x400_16spp_4b", "box"),      # BASELINE.json configs[0] (C1)
                                        ("duck_160x90_4spp_8b", "duck")])
def test_hip_path_reproduces_golden_fixtures(pt, box_scene, duck_scene, gpu_device, name, scene):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    W, H, spp, depth = [int(v) for v in g["params"]]
    sc, cam = (box_scene, BOX_CAMERA) if scene == "box" else (duck_scene, DUCK_CAMERA)
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), sc, fb, pt.CameraConfig(cam[0], cam[1]))
    tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0))
    tr.waitForRenderTask()
    assert np.array_equal(fb.rgb_host(), g["rgb"]) and np.array_equal(fb.yuv_host(), g["yuv"])
    c = tr.readCounters()
    assert [c[k] for k in ("paths", "segments", "box_tests", "tri_tests", "light_tests")] == g["counters"][:5].tolist()
    tr.close()


def test_python_render_manager_single_rank(pt, box_scene, box_oracle, gpu_device):
    cfg = pt.RendererConfig(samplesPerPixel=4, recursionDepth=4, resolution=pt.Resolution(64, 64), showTasks=False)
    mgr = pt.RenderManager(cfg, box_scene, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    mgr.renderFrame()
    ref = box_oracle.render(BOX_CAMERA, 64, 64, 4, 4)
    assert np.array_equal(mgr.getCurrentFrame(), ref[0]) and np.array_equal(mgr.getYUVFrame(), ref[1])
    # hot-swap of parameters, as RenderManager.h:214-230
    mgr.setSamplesPerPixel(2); mgr.setRecursionDepth(3); mgr.setResolution(pt.Resolution(48, 32))
    mgr.renderFrame()
    ref = box_oracle.render(BOX_CAMERA, 48, 32, 2, 3)
    assert np.array_equal(mgr.getCurrentFrame(), ref[0]) and np.array_equal(mgr.getYUVFrame(), ref[1])
    mgr.reset()


def test_band_tasks_on_several_streams_equal_whole_frame(pt, duck_scene, duck_oracle, gpu_device):
    """What a rank of an N-GPU run does: only its bands, on 2 streams; the union over ranks
    (here rendered one after another on the one GPU) is the oracle's frame."""
    import torch
    W, H, spp, depth, band, world = 64, 36, 2, 4, 8, 3
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), duck_scene, fb, pt.CameraConfig(DUCK_CAMERA[0], DUCK_CAMERA[1]))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rank in range(world):
        for i, t in enumerate(pt.row_band_tasks(W, H, band, world, rank)):
            tr.renderTaskAsync(t, streams[i % 2])
    for s in streams:
        tr.synchronizeStream(s)
    ref = duck_oracle.render(DUCK_CAMERA, W, H, spp, depth, threads=8)
    assert np.array_equal(fb.rgb_host(), ref[0]) and np.array_equal(fb.yuv_host(), ref[1])
    tr.close()


@pytest.mark.parametrize("name,parts,by_lanes", [("box", 2, 1), ("box", 4, 1), ("box", 8, 1), ("box", 4, 0), ("box", 8, 0), ("duck", 4, 1)])
def test_dealt_kernel_split_tiles_never_change_a_byte(pt, box_scene, duck_scene, box_oracle, duck_oracle, gpu_device, monkeypatch, name, parts, by_lanes):
    """The dealt kernels (mask walk for the box, stack walk for the duck), like the SAH kernel (test_gpu_sah.py), hand the
    dearest tiles of a launch whose longest chain bounds it to several waves of a few pixel rows each.
    PTAMD_SPLIT_PCT = 10 makes that the case on this small frame; every frame -- raster order, cost order, split tiles --
    must equal the oracle's, counters included.  The box's parts of 16 and 8 pixels walk by lanes (part_walk_by_lanes: all
    child boxes of the tree tested at once, 4 or 8 lanes per ray) unless PTAMD_PART_LANES=0 keeps them on the pair-by-pair
    walk; its parts of 32 pixels (more (ray, triangle) pairs than the item list holds) and the duck's always are."""
    monkeypatch.setenv("PTAMD_SPLIT_PCT", "10")
    monkeypatch.setenv("PTAMD_DEAL_SPLIT", str(parts))
    monkeypatch.setenv("PTAMD_PART_LANES", str(by_lanes))
    scene, oracle, cam, kernel = (box_scene, box_oracle, BOX_CAMERA, "dealt<") if name == "box" else (duck_scene, duck_oracle, DUCK_CAMERA, "dealt_stack")
    W, H, spp, depth = (200, 168, 3, 5) if name == "box" else (176, 160, 2, 4)
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), scene, fb, pt.CameraConfig(cam[0], cam[1]))
    ref = oracle.render(cam, W, H, spp, depth, threads=8)
    seen = 0
    for frame in range(5):
        fb.rgb.zero_(); fb.yuv.zero_()
        tr.resetCounters()
        tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
        assert kernel in tr.kernelName()
        cnt = tr.readCounters()
        assert np.array_equal(fb.rgb_host(), ref[0]) and np.array_equal(fb.yuv_host(), ref[1]), "frame %d" % frame
        for k in ("paths", "segments", "box_tests", "tri_tests", "light_tests"):      # the reference tree: the oracle's counts
            assert cnt[k] == ref[2][k], (frame, k, cnt[k], ref[2][k])
        seen = max(seen, tr.splitTiles())
    assert seen > 0
    tr.close()


def test_split_tiles_in_band_launches_keep_frame_and_band_costs(pt, box_scene, box_oracle, gpu_device, monkeypatch):
    """A rank's band launch (what the multi-GPU path renders) with its dearest tiles split: the union over ranks is the
    oracle's frame on every frame, and the per-band costs the dealer is fed do not move when tiles start being split
    (a dealt tile's cost is the trip count of its busiest pixel, whichever part holds that pixel)."""
    monkeypatch.setenv("PTAMD_SPLIT_PCT", "10")
    W, H, spp, depth, band, world = 320, 200, 2, 4, 8, 2
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    costs, splits = [], []
    for frame in range(4):
        fb.rgb.zero_(); fb.yuv.zero_()
        per_rank = []
        for rank in range(world):
            tr.renderBandsAsync(band, world, rank); tr.waitForRenderTask()
            per_rank.append(tr.bandCosts())
            splits.append(tr.splitTiles())
        assert np.array_equal(fb.rgb_host(), ref[0]) and np.array_equal(fb.yuv_host(), ref[1]), "frame %d" % frame
        costs.append(per_rank)
    assert max(splits) > 0
    assert all(c == costs[0] for c in costs[1:])
    tr.close()


@pytest.mark.parametrize("bvh,kernel", [(0, "dealt"), (1, "sah")])
def test_persistent_kernels_with_many_launches_in_flight(pt, box_scene, box_oracle, gpu_device, bvh, kernel):
    """The dealt and SAH kernels are persistent: a launch's waves take their tiles from a counter of that launch.  Many
    small launches queued on three streams without waiting (more than the 64 the event pool folds at, so counters
    are recycled while others are in flight) must still paint exactly the oracle's frame, and a second pass over the
    recycled counters must change nothing."""
    import torch
    W, H, spp, depth = 96, 72, 3, 4
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]), bvhMode=bvh)
    streams = [torch.cuda.Stream() for _ in range(3)]
    tasks = [pt.RenderTask(24, 8, ox, oy) for oy in range(0, H, 8) for ox in range(0, W, 24)] + \
            [pt.RenderTask(W, 8, 0, oy) for oy in range(0, H, 8)] + [pt.RenderTask(W, H, 0, 0)]
    ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    for sweep in range(3):
        for i, t in enumerate(tasks):
            tr.renderTaskAsync(t, streams[i % 3])
        for s in streams:
            tr.synchronizeStream(s)
        assert kernel in tr.kernelName()
        assert np.array_equal(fb.rgb_host(), ref[0]) and np.array_equal(fb.yuv_host(), ref[1]), "sweep %d" % sweep
    ms, n = tr.kernelTime()
    assert n == 3 * len(tasks)
    tr.close()


@pytest.mark.parametrize("W,H,band,world", [(64, 36, 8, 3), (100, 50, 4, 4), (64, 64, 16, 2)])
def test_single_launch_bands_equal_whole_frame(pt, box_scene, box_oracle, gpu_device, W, H, band, world):
    """pt_render_bands_async: one launch per rank covering all its bands (incl. a partial last
    band) -- union over ranks == oracle frame; each rank touches only its own rows."""
    spp, depth = 3, 4
    ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    tr.renderBandsAsync(band, world, 1)
    tr.waitForRenderTask()
    mine = pt.band_row_ranges(H, band, world, 1)
    img = fb.rgb_host()
    for r in range(H):
        owned = any(a <= r < b for a, b in mine)
        assert np.array_equal(img[r], ref[0][r]) if owned else not img[r].any()
    for rank in (0, 2, 3)[:world - 1]:
        tr.renderBandsAsync(band, world, rank)
    tr.waitForRenderTask()
    assert np.array_equal(fb.rgb_host(), ref[0]) and np.array_equal(fb.yuv_host(), ref[1])
    tr.close()


def _rank_worker(rank, world, port, W, H, spp, depth, band, out_path, algo=0, frames=2):
    import importlib
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    pt = importlib.import_module("multi-gpu-path-tracer_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)          # one GPU on this box: RCCL needs one per rank
    scene = pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_box.glb")).applyCornellConvention()
    cfg = pt.RendererConfig(samplesPerPixel=spp, recursionDepth=depth, gpuNumber=world,
                            resolution=pt.Resolution(W, H), showTasks=False, algorithmType=algo)
    mgr = pt.RenderManager(cfg, scene, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]), device_idx=0, band_rows=band)
    rgbs, statics, loads = [], [], []
    for _ in range(frames):                                               # later frames: buffers are reused
        mgr.getFramebuffer().rgb.zero_(); mgr.getFramebuffer().yuv.zero_()
        statics.append(mgr.static_)
        mgr.renderFrame()
        if rank == 0:
            rgbs.append(mgr.getCurrentFrame().copy())
    cnt = mgr.tracer.readCounters()
    tot = torch.tensor([float(cnt["segments"])], dtype=torch.float64)
    dist.all_reduce(tot)
    if rank == 0:
        costs = mgr.band_costs_ or []
        loads = [sum(costs[b] for b in l) for l in mgr.assignment_] if costs else []
        np.savez(out_path, rgb=mgr.getCurrentFrame(), yuv=mgr.getYUVFrame(), segments=np.array([tot.item() / frames]),
                 all_rgb=np.stack(rgbs), statics=np.array(statics), loads=np.array(loads, np.float64),
                 sizes=np.array([len(l) for l in mgr.assignment_]))
    dist.barrier()
    mgr.reset()
    dist.destroy_process_group()


def _rank_worker_in_flight(rank, world, port, W, H, spp, depth, band, out_path, fif, dealer="fsfl", host_frames=False):
    import importlib
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    pt = importlib.import_module("multi-gpu-path-tracer_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pt.SceneLoader().load(os.path.join(ROOT, "models", "cornell_box.glb")).applyCornellConvention()
    cfg = pt.RendererConfig(samplesPerPixel=spp, recursionDepth=depth, gpuNumber=world,
                            resolution=pt.Resolution(W, H), showTasks=False, algorithmType=pt.DSFL if dealer == "dsfl" else pt.FSFL)
    mgr = pt.RenderManager(cfg, scene, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]), device_idx=0, band_rows=band,
                           frames_in_flight=fif, host_frames=host_frames)
    assert mgr.frames_in_flight == fif and len(mgr._ring) == fif + 1      # (the cost-fed dealers keep their frames in flight too)
    frames, complete, host_ok = 2 * fif + 3, [], True
    first_buffer = mgr.getFramebuffer()
    for i in range(frames):
        if i == fif + 2:                       # every buffer of the ring has held a frame by now: wipe them (after a flush, so
            mgr.flush()                        # that nothing in flight is touched) -- what is read from here on was rendered from here on
            for fb in mgr._ring:
                fb.rgb.zero_(); fb.yuv.zero_()
            torch.cuda.synchronize()
        done = mgr.frames_completed
        mgr.renderFrame()
        if mgr.frames_completed == 0:
            # nothing is complete yet -- what getFramebuffer() hands out must not be a buffer a launch is writing to
            # (frame 0 renders into ring slot 1), and frameCompleted() says so
            torch.cuda.synchronize()
            assert not mgr.frameCompleted() and mgr.getFramebuffer() is first_buffer
            assert not bool(first_buffer.rgb.any()) and not bool(first_buffer.yuv.any())
            assert mgr.hostFrame() is None
        if rank == 0 and mgr.frames_completed > done:      # a frame was completed: it stays readable until the next renderFrame()
            complete.append(mgr.getCurrentFrame().copy())
            if host_frames:                                # ... and its copy in pinned host memory is that frame (RGB and YUV)
                hf = mgr.hostFrame()
                host_ok = host_ok and hf is not None and np.array_equal(hf[0], complete[-1]) and np.array_equal(hf[1], mgr.getYUVFrame())
    mgr.flush()
    assert host_ok, "a host-visible frame differed from the device frame"
    if dealer == "dsfl" and world > 1:
        assert not mgr.static_, "the dealer never took over"
    cnt = mgr.tracer.readCounters()
    tot = torch.tensor([float(cnt["segments"])], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot)
    if rank == 0:
        ring = np.stack([fb.rgb_host() for fb in mgr._ring])
        np.savez(out_path, rgb=mgr.getCurrentFrame(), yuv=mgr.getYUVFrame(), segments=np.array([tot.item() / frames]),
                 complete=np.stack(complete), ring=ring)
    if world > 1:
        dist.barrier()
    mgr.reset()
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fif,W,H,band,dealer,host_frames", [(2, 2, 96, 64, 8, "fsfl", False), (3, 3, 64, 40, 8, "fsfl", False), (1, 2, 96, 64, 8, "fsfl", True),
                                                                   (4, 4, 64, 64, 8, "fsfl", False), (3, 3, 384, 168, 8, "dsfl", True),      # (336 tiles per rank: costs are recorded)
                                                                   (1, 3, 64, 64, 8, "fsfl", True)])
def test_frames_in_flight_render_the_same_frames(box_oracle, gpu_device, tmp_path, world, fif, W, H, band, dealer, host_frames):
    """RenderManager(frames_in_flight = F): renderFrame() launches frame i and completes frame i - F + 1, F launches queued
    per GPU, each in its own buffer of a ring of F + 1 on its own stream, the exchange per frame in launch order.  Every
    complete frame handed out is the oracle's, every ring buffer is rendered afresh (they are wiped mid-run), the ranks'
    segments add up to frames x the oracle's.  World 1 too: one process, no exchange.  dealer = dsfl: the cost-fed dealer with
    frames in flight (every frame carries the assignment it was launched with).  host_frames: every completed
    frame is also delivered into pinned host memory (SURVEY 8d's frame) and equals the device frame."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "frame.npz")
    spp, depth = 3, 4
    mp.spawn(_rank_worker_in_flight, args=(world, port, W, H, spp, depth, band, out, fif, dealer, host_frames), nprocs=world, join=True)
    got = np.load(out)
    ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    assert np.array_equal(got["rgb"], ref[0]) and np.array_equal(got["yuv"], ref[1])
    for k, img in enumerate(got["complete"]):
        assert np.array_equal(img, ref[0]), "complete frame %d differs from the oracle" % k
    assert len(got["complete"]) == 5            # 3 before the mid-run flush, 2 after it; the flushes complete the others
    for k, img in enumerate(got["ring"]):
        assert np.array_equal(img, ref[0]), "ring buffer %d was not rendered afresh" % k
    assert int(got["segments"][0]) == ref[2]["segments"]


@pytest.mark.parametrize("world,W,H,band", [(2, 96, 64, 8), (3, 64, 36, 8)])
def test_multi_rank_render_manager_frame_equals_oracle(box_oracle, gpu_device, tmp_path, world, W, H, band):
    """The N > 1 path end to end on real kernels: `world` processes (all on this box's one GPU,
    `gloo` for the exchange) each render only their row bands in ONE launch, rank 0 gathers --
    the gathered frame must be the oracle's, and the ranks' segment counts must add up to it."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "frame.npz")
    spp, depth = 3, 4
    mp.spawn(_rank_worker, args=(world, port, W, H, spp, depth, band, out), nprocs=world, join=True)
    got = np.load(out)
    ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    assert np.array_equal(got["rgb"], ref[0]) and np.array_equal(got["yuv"], ref[1])
    assert int(got["segments"][0]) == ref[2]["segments"]


def test_multi_rank_cost_fed_dealer_on_real_kernels(box_oracle, gpu_device, tmp_path):
    """algorithmType = DSFL on the multi-rank path: frame 1 is dealt b % N, then the bands' recorded costs re-deal them
    longest-processing-time-first (non-strided lists through pt_render_band_list_async, true-size sends).  Every frame
    is the oracle's frame, the ranks' segments add up, the dealt loads are within one band of each other."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "frame.npz")
    world, W, H, band, spp, depth, frames = 3, 384, 168, 8, 2, 4, 4       # 336 tiles per rank: costs are recorded
    mp.spawn(_rank_worker, args=(world, port, W, H, spp, depth, band, out, 1, frames), nprocs=world, join=True)
    got = np.load(out)
    ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    for f in range(frames):
        assert np.array_equal(got["all_rgb"][f], ref[0]), "frame %d differs from the oracle" % f
    assert np.array_equal(got["yuv"], ref[1])
    assert int(got["segments"][0]) == ref[2]["segments"]
    assert got["statics"].tolist()[0] is True and not got["statics"].tolist()[-1]       # the dealer took over
    loads = got["loads"]
    assert len(loads) == world and loads.max() - loads.min() <= loads.sum() / 21 * 2     # 21 bands: within ~2 mean bands
    assert int(got["sizes"].sum()) == 21


def test_band_list_launch_and_band_costs(pt, box_scene, box_oracle, gpu_device):
    """pt_render_band_list_async renders exactly the listed bands (partial last band included) and refuses lists the
    kernel's row map cannot express; pt_band_costs_read gives a band the same cost whichever launch rendered it."""
    W, H, band, spp, depth = 2048, 68, 8, 2, 3                  # 9 bands, the last of 4 rows; 256 tiles per row band
    ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    costs = {}
    for bands in ([1, 4, 6], [0, 2, 3, 5, 7, 8]):
        tr.renderBandListAsync(band, bands); tr.waitForRenderTask()
        c = tr.bandCosts()
        assert len(c) == len(bands) and all(v > 0 for v in c)
        costs.update(dict(zip(bands, c)))
    assert np.array_equal(fb.rgb_host(), ref[0]) and np.array_equal(fb.yuv_host(), ref[1])
    fb.rgb.zero_(); fb.yuv.zero_()
    tr.renderBandListAsync(band, list(range(9))); tr.waitForRenderTask()
    assert tr.bandCosts() == [costs[b] for b in range(9)]                      # a band's cost does not depend on its launch
    assert np.array_equal(fb.rgb_host(), ref[0])
    tr.renderBandsAsync(band, 2, 1); tr.waitForRenderTask()                    # the strided form records costs too
    assert tr.bandCosts() == [costs[b] for b in (1, 3, 5, 7)]
    for bad in ([0, 0], [9], [-1], [8, 2]):                                    # repeated, out of range, partial band not last
        with pytest.raises(pt.capi.PtError) as e:
            tr.renderBandListAsync(band, bad)
        assert e.value.status == 1
    tr.renderBandListAsync(band, [])                                           # nothing to do
    tr.close()


def _read_ppm(path):
    b = open(path, "rb").read()
    magic, dims, mx, data = b.split(b"\n", 3)
    w, h = map(int, dims.split())
    assert magic == b"P6" and mx == b"255"
    return np.frombuffer(data, np.uint8).reshape(h, w, 3)


@pytest.mark.parametrize("extra", [[], ["--streams", "2"], ["--streams", "4", "--algo", "dsfl", "--frames", "3"],
                                   ["--streams", "4", "--algo", "dsdl", "--frames", "3", "--max-tasks-in-row", "4"],
                                   # the intra-frame band queue: workers pull 8-row bands as their streams drain
                                   ["--streams", "4", "--algo", "dqrb", "--frames", "3"], ["--streams", "3", "--algo", "dqrb"]])
def test_cpp_driver_glb_to_ppm_matches_oracle(pt, box_oracle, gpu_device, tmp_path, extra):
    exe = os.path.join(ROOT, "multi-gpu-path-tracer_amd", "pt_render")
    assert os.path.exists(exe), "pt_render missing: __graft_entry__.build() builds it"
    out = tmp_path / "o.ppm"
    yuv = tmp_path / "o.yuv"
    cmd = [exe, "0", os.path.join(ROOT, "models", "cornell_box.glb"), "--cornell", "--width", "96", "--height", "64",
           "--spp", "4", "--depth", "4", "--out", str(out), "--yuv", str(yuv)] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "Path Tracing took:" in res.stdout and "initializing in:" in res.stdout      # main.cu:64,85
    ref = box_oracle.render(BOX_CAMERA, 96, 64, 4, 4, threads=8)
    assert np.array_equal(_read_ppm(str(out)), ref[0])                                   # any scheduler, same image
    assert np.array_equal(np.frombuffer(yuv.read_bytes(), np.uint8), ref[1])
    import json
    frames = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert all(f["segments"] == ref[2]["segments"] for f in frames)                      # every pixel rendered exactly once, every frame
    if "dqrb" in extra:
        assert all(f["tasks"] == 8 for f in frames)                                      # 64 rows in bands of 8


@pytest.mark.parametrize("extra", [["--frames-in-flight", "2"], ["--frames-in-flight", "3", "--streams", "2"],
                                   ["--frames-in-flight", "2", "--streams", "4", "--max-tasks-in-row", "2"]])
def test_cpp_driver_frames_in_flight(pt, box_oracle, gpu_device, tmp_path, extra):
    """The C++ host with RendererConfig::framesInFlight > 1 (`pt_render --frames-in-flight F`): every worker launches frame i
    into slot i % (F + 1) of its engine's ring of device framebuffers on stream i % F and copies frame i - F + 1 out of ITS
    slot; RenderManager::flush() completes the rest.  Seven frames: the last one handed out is the oracle's, and the
    engines' segment counters add up to seven frames' worth -- every frame was rendered whole, none twice."""
    exe = os.path.join(ROOT, "multi-gpu-path-tracer_amd", "pt_render")
    out, yuv = tmp_path / "o.ppm", tmp_path / "o.yuv"
    cmd = [exe, "0", os.path.join(ROOT, "models", "cornell_box.glb"), "--cornell", "--width", "96", "--height", "64",
           "--spp", "4", "--depth", "4", "--frames", "7", "--out", str(out), "--yuv", str(yuv)] + extra
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    ref = box_oracle.render(BOX_CAMERA, 96, 64, 4, 4, threads=8)
    assert np.array_equal(_read_ppm(str(out)), ref[0])
    assert np.array_equal(np.frombuffer(yuv.read_bytes(), np.uint8), ref[1])
    import json
    line = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")][-1]
    assert line["frames"] == 7 and line["frames_in_flight"] == int(extra[1]) and line["segments"] == 7 * ref[2]["segments"]


def test_cpp_driver_dsdl_on_a_frame_too_thin_to_bisect(pt, box_oracle, gpu_device, tmp_path):
    """a 16x8 frame is a 2x1 block grid; eight workers ask for three levels of bisection and two of them
    cannot cut.  The degenerate branch must still yield 2^(depth - level) tasks (most of them empty), so that DSDL's new
    layout is adopted instead of silently dropped -- and the image is the oracle's either way."""
    exe = os.path.join(ROOT, "multi-gpu-path-tracer_amd", "pt_render")
    out = tmp_path / "o.ppm"
    cmd = [exe, "0", os.path.join(ROOT, "models", "cornell_box.glb"), "--cornell", "--width", "16", "--height", "8", "--spp", "4",
           "--depth", "4", "--out", str(out), "--streams", "8", "--algo", "dsdl", "--frames", "3", "--max-tasks-in-row", "4"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    ref = box_oracle.render(BOX_CAMERA, 16, 8, 4, 4, threads=2)
    assert np.array_equal(_read_ppm(str(out)), ref[0])


def test_cpp_driver_error_behaviour(gpu_device, tmp_path):
    exe = os.path.join(ROOT, "multi-gpu-path-tracer_amd", "pt_render")
    res = subprocess.run([exe, "0", str(tmp_path / "nope.glb")], capture_output=True, text=True, timeout=60)
    assert res.returncode == 1 and "cannot open" in res.stderr
    # cornell_box.glb as-is has no emissive material: the reference crashes; here exit(99)
    res = subprocess.run([exe, "0", os.path.join(ROOT, "models", "cornell_box.glb")], capture_output=True, text=True, timeout=60)
    assert res.returncode == 99 and "no emissive face" in res.stderr


def _fullsize(name):
    import json
    with open(os.path.join(GOLD, "fullsize.json")) as f:
        return json.load(f)[name]


def _band_hashes(rgb, yuv, W, H, rows=8):
    """As tests/golden/make_fullsize.py::band_hashes: names the rows when a full-size frame differs."""
    out = []
    rgb = rgb.reshape(H, W * 3)
    yuv = yuv.reshape(-1)
    y = yuv[:W * H].reshape(H, W)
    u = yuv[W * H:W * H + W * H // 4].reshape(H // 2, W // 2)
    v = yuv[W * H + W * H // 4:].reshape(H // 2, W // 2)
    for r0 in range(0, H, rows):
        r1 = min(H, r0 + rows)
        h = hashlib.sha256()
        for part in (rgb[r0:r1], y[r0:r1], u[(r0 + 1) // 2:(r1 + 1) // 2], v[(r0 + 1) // 2:(r1 + 1) // 2]):
            h.update(np.ascontiguousarray(part).tobytes())
        out.append(h.hexdigest()[:16])
    return out


def _assert_equals_fullsize_fixture(name, rgb, yuv, cnt):
    g = _fullsize(name)
    W, H = g["params"][:2]
    bands = _band_hashes(rgb, yuv, W, H)
    bad = [i for i, (a, b) in enumerate(zip(bands, g["band_sha16"])) if a != b]
    assert not bad, "%s: %d of %d 8-row bands differ from the oracle's frame, first band %d" % (name, len(bad), len(bands), bad[0])
    assert hashlib.sha256(rgb.tobytes() + yuv.tobytes()).hexdigest() == g["sha256_rgb_yuv"]
    assert {k: cnt[k] for k in g["counters"]} == g["counters"]


def test_full_size_c2_equals_the_oracle_and_properties(pt, box_scene, gpu_device):
    """BASELINE.json configs[1] (1920x1080, 256 spp, 8 bounces), the configuration bench.py runs, in the default walk:
    the frame's bytes (sha256 over RGB || YUV, and per 8-row band) and all five work counters equal the ORACLE's
    full-size render (tests/golden/fullsize.json, made by make_fullsize.py: 1.33e9 segments on the CPU), plus the
    size-independent properties: determinism across frames, partition invariance (whole frame vs 8-row bands of 3
    interleaved 'ranks'), exact path count, counter ratios."""
    W, H, spp, depth = 1920, 1080, 256, 8
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    tr.resetCounters()
    tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
    h1 = hashlib.sha256(fb.rgb_host().tobytes() + fb.yuv_host().tobytes()).hexdigest()
    c1 = tr.readCounters()
    _assert_equals_fullsize_fixture("box_1920x1080_256spp_8b", fb.rgb_host(), fb.yuv_host(), c1)
    assert c1["paths"] == W * H * spp
    assert abs(c1["segments"] / c1["paths"] - 2.504) < 0.02
    assert abs(c1["box_tests"] / c1["segments"] - 14.3) < 0.15 and abs(c1["tri_tests"] / c1["segments"] - 16.8) < 0.17
    fb.rgb.zero_(); fb.yuv.zero_()
    tr.resetCounters()
    for rank in range(3):
        for t in pt.row_band_tasks(W, H, 8, 3, rank):
            tr.renderTaskAsync(t)
    tr.waitForRenderTask()
    h2 = hashlib.sha256(fb.rgb_host().tobytes() + fb.yuv_host().tobytes()).hexdigest()
    assert h1 == h2 and tr.readCounters() == c1
    img = fb.rgb_host()
    assert img.mean() > 10 and img[:, :, 0].mean() != img[:, :, 1].mean()        # a lit, coloured Cornell box
    tr.close()


@pytest.mark.parametrize("bvh", [0, 1])
@pytest.mark.parametrize("name", ["duck_1920x1080_16spp_8b", "duck_1920x1080_256spp_8b"])
def test_full_size_duck_frame_equals_the_oracle(pt, duck_scene, gpu_device, bvh, name):
    """configs[2]'s scene and frame size (1920x1080, 8 bounces) at 16 spp (the oracle needs minutes) and at the config's own
    256 spp (the oracle needed two hours: 1.33e9 segments x 2 120 triangle tests): bytes and counters of the reference-tree
    walk equal the oracle's full-size frame; the SAH tree gives the same bytes, paths, segments and light tests (its box /
    triangle test counts are its own)."""
    import json
    with open(os.path.join(GOLD, "fullsize.json")) as f:
        if name not in json.load(f):
            pytest.skip("fixture %s not generated (tests/golden/make_fullsize.py)" % name)
    g = _fullsize(name)
    W, H, spp, depth = g["params"]
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), duck_scene, fb, pt.CameraConfig(DUCK_CAMERA[0], DUCK_CAMERA[1]), bvhMode=bvh)
    tr.resetCounters()
    tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
    cnt = tr.readCounters()
    if bvh == 0:
        _assert_equals_fullsize_fixture(name, fb.rgb_host(), fb.yuv_host(), cnt)
    else:
        assert hashlib.sha256(fb.rgb_host().tobytes() + fb.yuv_host().tobytes()).hexdigest() == g["sha256_rgb_yuv"]
        assert all(cnt[k] == g["counters"][k] for k in ("paths", "segments", "light_tests"))
        assert cnt["tri_tests"] * 50 < g["counters"]["tri_tests"]
    tr.close()


def test_reference_80_byte_triangles_render_the_golden_box(pt, box_scene, gpu_device):
    """VERDICT r2 weak 9 / INTEGRATION.md 2: the reference's std::vector<Triangle> is 80-byte records (float2 is 8-byte
    aligned), handed to the C ABI through pt_upload_scene_strided with PT_TRIANGLE_LAYOUT_REFERENCE's offsets; padding
    bytes hold junk.  The frame must be the golden box, byte for byte, counters included."""
    capi = pt.capi
    g = np.load(os.path.join(GOLD, "box_160x90_8spp_8b.npz"))
    W, H, spp, depth = [int(v) for v in g["params"]]
    tris = box_scene.triangles
    raw = np.full(len(tris) * 80, 0xCD, np.uint8)
    rec = raw.view(capi.REFERENCE_TRIANGLE_DTYPE)
    for v in ("v0", "v1", "v2"):
        rec[v]["position"], rec[v]["texCoords"] = tris[v]["position"], tris[v]["texCoords"]
    rec["textureIdx"], rec["materialIdx"] = tris["textureIdx"], tris["materialIdx"]
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    lay = capi.TriangleLayout.of(rec.dtype)
    assert (lay.stride, list(lay.position), list(lay.texCoords), lay.textureIdx, lay.materialIdx) == (80, [0, 24, 48], [16, 40, 64], 72, 76)
    # re-loaded through the documented route for the reference's own records
    tr.uploadSceneStrided(rec, box_scene.materials, box_scene.textures)
    tr.resetCounters()
    tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
    assert np.array_equal(fb.rgb_host(), g["rgb"]) and np.array_equal(fb.yuv_host(), g["yuv"])
    c = tr.readCounters()
    assert [c[k] for k in ("paths", "segments", "box_tests", "tri_tests", "light_tests")] == g["counters"][:5].tolist()
    # the cast INTEGRATION.md used to show (80-byte records read with a 68-byte stride) is a different scene
    wrong = np.frombuffer(raw[:len(tris) * 68].tobytes(), capi.TRIANGLE_DTYPE)
    assert wrong.tobytes() != tris.tobytes()
    tr.close()


def test_bench_without_a_launcher_runs_two_ranks(gpu_device):
    """VERDICT r2 missing 2: `python bench.py --gpus 2` with no torchrun must start two ranks itself (a child
    torch.distributed.run) and print n_gpus = ranks = 2 -- here as a rehearsal (both ranks on the one GPU, gloo)."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--config", "c1",
                          "--steps", "2", "--warmup", "1", "--no-host-visible"], capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"].startswith("gloo")
    assert d["segments_per_frame"] == 5903674                 # C1's exact count (tests/golden/box_400x400_16spp_4b.npz)


@pytest.mark.parametrize("bvh", [0, 1])
def test_cost_ordered_tile_schedule_never_changes_a_byte(pt, box_scene, gpu_device, bvh):
    """pt_set_tile_schedule(1) reorders tiles from the previous launch's per-tile costs.  Every frame
    (first = raster order, later = re-sorted each time) must equal the raster-order frame byte for
    byte with identical work counters -- whole-frame tasks, single-launch row bands, and the same
    task issued back to back on two streams (the schedule is shared state between those launches).
    The framebuffer is cleared before every frame, so a tile dropped or duplicated by a broken
    permutation would show."""
    import torch
    W, H, spp, depth = 1024, 512, 4, 4                       # 8192 tiles: above the 4096-tile threshold
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    if bvh:
        tr.setBvhMode(1)
    whole = pt.RenderTask(W, H, 0, 0)

    def frame(launch):
        fb.rgb.zero_(); fb.yuv.zero_()
        torch.cuda.synchronize()
        tr.resetCounters()
        launch()
        tr.waitForRenderTask()
        return hashlib.sha256(fb.rgb_host().tobytes() + fb.yuv_host().tobytes()).hexdigest(), tr.readCounters()

    tr.setTileSchedule(0)
    want = frame(lambda: tr.renderTaskAsync(whole))
    assert want[1]["paths"] == W * H * spp
    tr.setTileSchedule(1)
    for _ in range(4):
        assert frame(lambda: tr.renderTaskAsync(whole)) == want

    def bands():
        tr.renderBandsAsync(8, 2, 0)
        tr.renderBandsAsync(8, 2, 1)
    for _ in range(3):
        assert frame(bands) == want

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def two_streams():                                        # 2 x the same frame: counters double
        tr.renderTaskAsync(whole, s1)
        tr.renderTaskAsync(whole, s2)
        tr.renderTaskAsync(whole, s1)
        tr.renderTaskAsync(whole, s2)
        tr.synchronizeStream(s1); tr.synchronizeStream(s2)
    h, c = frame(two_streams)
    assert h == want[0] and c["segments"] == 4 * want[1]["segments"] and c["tri_tests"] == 4 * want[1]["tri_tests"]
    # more task geometries than the 16 tracked ones: the cache evicts or falls back, images stay put
    for k in range(20):
        t = pt.RenderTask(W - 8 * k, H, 0, 0)
        fb.rgb.zero_(); fb.yuv.zero_()
        tr.renderTaskAsync(t); tr.renderTaskAsync(t); tr.waitForRenderTask()
    tr.setTileSchedule(0)
    tr.close()


def _tasks_c(pt, tasks):
    arr = (pt.capi.Task * max(len(tasks), 1))()
    for i, t in enumerate(tasks):
        arr[i] = pt.capi.Task(t.width, t.height, t.offset_x, t.offset_y)
    return arr


def test_native_pack_equals_python_pack_bands_and_round_trips(pt, box_scene, gpu_device):
    """The native gather's device-side packing (pt_fb_pack_tasks, what every peer GPU runs before its ncclSend) against
    the Python path's scheduling.pack_bands byte for byte on row bands, then pack -> clear -> unpack on ragged rectangles
    (odd offsets: chroma samples belong to whoever owns the even row / column), and the single-device pt_gather."""
    import ctypes as C
    import torch
    W, H, spp, depth, band = 200, 76, 2, 3, 8
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
    rgb0, yuv0 = fb.rgb.clone(), fb.yuv.clone()
    lib = pt.capi.lib()
    for world, rank in [(3, 1), (2, 0), (4, 3)]:                                  # 76 % 8 = 4: a partial last band
        tasks = pt.row_band_tasks(W, H, band, world, rank)
        ranges = pt.band_row_ranges(H, band, world, rank)
        n = C.c_size_t()
        pt.capi.check(lib.pt_fb_packed_size(W, H, _tasks_c(pt, tasks), len(tasks), C.byref(n)))
        assert n.value == pt.packed_size(W, H, ranges)
        packed = torch.zeros(n.value, dtype=torch.uint8, device="cuda")
        pt.capi.check(lib.pt_fb_pack_tasks(tr._h, _tasks_c(pt, tasks), len(tasks), C.c_void_p(packed.data_ptr()), None))
        torch.cuda.synchronize()
        assert torch.equal(packed, pt.pack_bands(fb.rgb, fb.yuv, W, H, ranges, torch))
    ragged = [pt.RenderTask(37, 30, 0, 0), pt.RenderTask(163, 30, 37, 0), pt.RenderTask(200, 1, 0, 30),
              pt.RenderTask(101, 45, 0, 31), pt.RenderTask(99, 45, 101, 31)]       # a partition of the frame
    sizes = []
    for gi, group in enumerate((ragged[:2], ragged[2:])):
        n = C.c_size_t()
        pt.capi.check(lib.pt_fb_packed_size(W, H, _tasks_c(pt, group), len(group), C.byref(n)))
        sizes.append(n.value)
        packed = torch.zeros(n.value, dtype=torch.uint8, device="cuda")
        pt.capi.check(lib.pt_fb_pack_tasks(tr._h, _tasks_c(pt, group), len(group), C.c_void_p(packed.data_ptr()), None))
        torch.cuda.synchronize()
        if gi == 0:
            keep, keep_group = packed, group
    assert sum(sizes) == rgb0.numel() + yuv0.numel()                                # every byte of the frame owned exactly once
    fb.rgb.zero_(); fb.yuv.zero_()
    pt.capi.check(lib.pt_fb_unpack_tasks(tr._h, _tasks_c(pt, keep_group), len(keep_group), C.c_void_p(keep.data_ptr()), None))
    torch.cuda.synchronize()
    img, ref = fb.rgb_host(), rgb0.cpu().numpy().reshape(H, W, 3)
    assert np.array_equal(img[H - 30:], ref[H - 30:]) and not img[:H - 30].any()     # rows of tasks 0 and 1 only (offset_y 0 = bottom)
    fb.rgb.copy_(rgb0); fb.yuv.copy_(yuv0)
    # single-device gather group: no RCCL involved, the frame reaches the host buffers
    g = C.c_void_p()
    devs = (C.c_void_p * 1)(tr._h)
    pt.capi.check(lib.pt_gather_create(devs, 1, C.byref(g)))
    host_rgb = np.zeros(rgb0.numel(), np.uint8); host_yuv = np.zeros(yuv0.numel(), np.uint8)
    owner = (C.c_int * len(ragged))(*([0] * len(ragged)))
    pt.capi.check(lib.pt_gather_frame(g, _tasks_c(pt, ragged), owner, len(ragged), pt.capi.np_ptr(host_rgb), pt.capi.np_ptr(host_yuv)))
    assert np.array_equal(host_rgb, rgb0.cpu().numpy()) and np.array_equal(host_yuv, yuv0.cpu().numpy())
    name = C.create_string_buffer(256)
    pt.capi.check(lib.pt_gather_transport(g, name, 256))
    assert name.value == b"none"
    pt.capi.check(lib.pt_gather_destroy(g))
    # A group of TWO engines (both on the one GPU of this box, each with a framebuffer of its own): RCCL wants one rank
    # per GPU, so the group takes its peer-copy transport (the fallback of a host without a usable librccl) and the whole
    # multi-device flow of pt_gather_frame runs -- every peer packs its rectangles, the strings travel to engine 0, engine
    # 0 scatters them and hands the frame to the host.  Engine 1 renders the rectangles engine 0 does not.
    fb2 = pt.Framebuffer(pt.Resolution(W, H))
    tr2 = pt.DevicePathTracer(0, spp, depth, (8, 8), box_scene, fb2, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    fb.rgb.zero_(); fb.yuv.zero_()
    own = [0, 1, 1, 0, 1]
    for t, o in zip(ragged, own):
        (tr if o == 0 else tr2).renderTaskAsync(t)
    tr.waitForRenderTask(); tr2.waitForRenderTask()
    assert not torch.equal(fb.rgb, rgb0)                                             # engine 0 holds only its own part
    g2 = C.c_void_p()
    pt.capi.check(lib.pt_gather_create((C.c_void_p * 2)(tr._h, tr2._h), 2, C.byref(g2)))
    pt.capi.check(lib.pt_gather_transport(g2, name, 256))
    assert name.value.startswith(b"peer-copy (one GPU is listed more than once")
    for frame in range(2):                                                           # staging buffers are reused
        host_rgb[:] = 0; host_yuv[:] = 0
        pt.capi.check(lib.pt_gather_frame(g2, _tasks_c(pt, ragged), (C.c_int * len(ragged))(*own), len(ragged),
                                          pt.capi.np_ptr(host_rgb), pt.capi.np_ptr(host_yuv)))
        assert np.array_equal(host_rgb, rgb0.cpu().numpy()) and np.array_equal(host_yuv, yuv0.cpu().numpy())
    bad_owner = (C.c_int * len(ragged))(0, 1, 2, 0, 1)
    with pytest.raises(pt.capi.PtError):
        pt.capi.check(lib.pt_gather_frame(g2, _tasks_c(pt, ragged), bad_owner, len(ragged), pt.capi.np_ptr(host_rgb), pt.capi.np_ptr(host_yuv)))
    pt.capi.check(lib.pt_gather_destroy(g2))
    tr2.close()
    tr.close()


def test_full_size_c5_properties(pt, box_scene, gpu_device):
    """BASELINE.json configs[4] (3840x2160, 1024 spp, 16 bounces; ~0.9 s per frame): the oracle's full-size frame, exact
    path count, determinism across two launches (raster order, then cost order), and the union of the band lists a
    cost-fed dealer hands 8 ranks == the whole frame, bytes and every counter."""
    W, H, spp, depth = 3840, 2160, 1024, 16
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))

    def frame(launch):
        fb.rgb.zero_(); fb.yuv.zero_()
        tr.resetCounters()
        launch()
        tr.waitForRenderTask()
        return hashlib.sha256(fb.rgb_host().tobytes() + fb.yuv_host().tobytes()).hexdigest(), tr.readCounters()

    h1, c1 = frame(lambda: tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)))
    assert c1["paths"] == W * H * spp and 2.0 < c1["segments"] / c1["paths"] < 3.2
    # ... and since round 3 it is no longer beyond the oracle: the CPU restatement rendered this frame once (2.1e10
    # segments; tests/golden/make_fullsize.py): bytes (whole frame and per 8-row band) and all five counters must equal it
    import json
    with open(os.path.join(GOLD, "fullsize.json")) as f:
        have = json.load(f)
    if "box_3840x2160_1024spp_16b" in have:
        _assert_equals_fullsize_fixture("box_3840x2160_1024spp_16b", fb.rgb_host(), fb.yuv_host(), c1)
    assert (h1, c1) == frame(lambda: tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)))
    tr.renderBandListAsync(8, list(range(270))); tr.waitForRenderTask()
    lists = pt.lpt_assignment(tr.bandCosts(), 8)
    assert sorted(b for l in lists for b in l) == list(range(270))

    def dealt():
        for l in lists:
            tr.renderBandListAsync(8, l)
    assert (h1, c1) == frame(dealt)
    tr.close()


def test_bench_line_contract(gpu_device):
    """bench.py prints ONE JSON line the driver parses: the contract's keys, a roofline block whose fraction is a
    fraction (or null when no profile of this build is committed), the CPU baseline with the cores it used."""
    import json
    import sys
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c1", "--steps", "3", "--warmup", "1",
                          "--cpu-rows", "16"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "ray-segments/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["dtype"] == "f32"
    assert d["vs_baseline"] is None and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("valu-issue", "hbm") and r["peak"] > 0 and r["kernel_ms_avg"] > 0 and r["launches"] == 3
    assert r["frac"] is None or 0 < r["frac"] <= 1
    assert (r["achieved"] is None) == (r["frac"] is None)
    assert "render_kernel" in r["kernel"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["host_cpu_count"] >= c["cores"]
    assert d["value"] > 100 * c["value"] / c["cores"]                 # a GPU, not a fallback


def test_bench_native_host_leg(gpu_device):
    """`bench.py --host native` runs the C++ host mirror (pt_render: ONE process, a worker thread per GPU and
    stream, pt_gather_frame) as a child and relays a line with the keys of the Python leg.  Two engines on this box's one GPU
    (--rehearse: the gather then moves its strings with the peer-copy transport), C1's frame: the segment count is C1's exact one."""
    import json
    import sys
    for extra, ranks in (([], 1), (["--gpus", "2", "--rehearse"], 2)):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--host", "native", "--config", "c1", "--steps", "3", "--warmup", "1"] + extra,
                             capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, res.stdout[-2000:]
        d = json.loads(lines[0])
        for k in ("metric", "value", "unit", "n_gpus", "ranks", "backend", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "frames_in_flight", "segments_per_frame", "tests_per_segment", "kernel_build_id", "host"):
            assert k in d, k
        assert d["metric"] == "ray-segments/s" and d["n_gpus"] == ranks and d["ranks"] == ranks and d["host"].startswith("native")
        assert d["segments_per_frame"] == 5903674                 # C1's exact count (tests/golden/box_400x400_16spp_4b.npz)
        assert d["value"] > 0 and d["ms_per_step"] > 0
        if ranks == 2:
            assert d["backend"] and not d["backend"].startswith("none"), d["backend"]      # rccl ..., or the peer-copy transport on one device
