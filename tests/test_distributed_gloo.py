"""N > 1 path on CPU: world_size-2 and -3 `gloo` runs of the frame gather (the path's one
exchange step).  Each rank fills ONLY its own row bands of a local framebuffer with the
bytes a render would produce (taken from a reference frame), then gather_frame() must
reassemble the exact full frame on rank 0 -- RGB and all three YUV420 planes."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _costs(n_bands, seed=5):
    return [int(v) for v in np.random.default_rng(seed).integers(1, 1000, n_bands)]


def _worker(rank, world, port, W, H, band, out_path, dynamic=False):
    sys.path.insert(0, ROOT)
    pt = importlib.import_module(PKG)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = W * H
    rng = np.random.default_rng(42)                      # same reference frame on every rank
    ref_rgb = torch.from_numpy(rng.integers(0, 256, total * 3, dtype=np.uint8))
    ref_yuv = torch.from_numpy(rng.integers(0, 256, total + 2 * (total // 4), dtype=np.uint8))
    rgb = torch.zeros_like(ref_rgb)
    yuv = torch.zeros_like(ref_yuv)
    n_bands = (H + band - 1) // band
    # dynamic: the cost-fed dealer's uneven, non-strided band lists (same on every rank: same costs)
    assignment = pt.lpt_assignment(_costs(n_bands), world) if dynamic else pt.static_assignment(n_bands, world)
    ranges = pt.bands_to_ranges(H, band, assignment[rank])
    assert dynamic or ranges == pt.band_row_ranges(H, band, world, rank)
    # "render": copy only this rank's bands (pack from the reference, unpack locally)
    mine = pt.pack_bands(ref_rgb, ref_yuv, W, H, ranges, torch)
    pt.unpack_bands(mine, rgb, yuv, W, H, ranges)
    packed = pt.pack_rank_bands(rgb, yuv, W, H, band, assignment[rank], torch)
    assert torch.equal(packed, pt.pack_bands(rgb, yuv, W, H, ranges, torch))      # row-selection form == per-band slices
    pt.gather_frame(packed, W, H, band, rgb, yuv, assignment=assignment if dynamic else None)
    if rank == 0:
        ok = bool(torch.equal(rgb, ref_rgb) and torch.equal(yuv, ref_yuv))
        open(out_path, "w").write("ok" if ok else "mismatch")
    else:
        # non-root ranks must still hold only their own bands
        other = [r for r in range(H) if not any(a <= r < b for a, b in ranges)]
        assert all(int(rgb[r * W * 3:(r + 1) * W * 3].sum()) == 0 for r in other[:5])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,H,band", [(2, 64, 36, 8), (3, 48, 50, 4), (2, 32, 6, 8),
                                              (8, 1920, 1080, 8)])     # the scaling run: 135 bands, ranks own 17 or 16
def test_gather_frame_gloo(tmp_path, world, W, H, band):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), W, H, band, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


@pytest.mark.parametrize("world,W,H,band", [(2, 64, 36, 8), (3, 48, 50, 4), (3, 64, 64, 8), (8, 1920, 1080, 8)])
def test_gather_frame_gloo_with_the_cost_fed_dealer(tmp_path, world, W, H, band):
    """The same exchange under the dynamic dealer: ranks own different NUMBERS of bands in no regular pattern, every
    rank sends its true size (no padding), a partial last band of the frame (36 % 8, 50 % 4) is some rank's last band."""
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), W, H, band, str(out), True), nprocs=world, join=True)
    assert out.read_text() == "ok"


def test_lpt_dealer_properties(pt):
    for n_bands, world, seed in [(135, 8, 1), (270, 8, 2), (17, 3, 3), (5, 8, 4), (1, 2, 5)]:
        costs = _costs(n_bands, seed)
        a = pt.lpt_assignment(costs, world)
        assert a == pt.lpt_assignment(list(costs), world)                               # deterministic
        assert sorted(b for l in a for b in l) == list(range(n_bands))                  # every band exactly once
        assert all(l == sorted(l) for l in a)                                           # ascending: a partial last band is last
        loads = [sum(costs[b] for b in l) for l in a]
        assert max(loads) <= sum(costs) / world + max(costs)                            # the LPT bound
        static = [sum(costs[b] for b in l) for l in pt.static_assignment(n_bands, world)]
        assert max(loads) <= max(static)
    # equal costs: the dealer reproduces a round-robin-sized split (no rank more than one band ahead)
    a = pt.lpt_assignment([7] * 135, 8)
    assert sorted(len(l) for l in a) == [16] + [17] * 7
    assert pt.uniform_bands(1920, 1080, 8) and not pt.uniform_bands(64, 36, 8) and not pt.uniform_bands(48, 50, 4)


def test_single_rank_gather_is_identity(pt):
    W, H, band = 16, 8, 2
    total = W * H
    rgb = torch.arange(total * 3, dtype=torch.int64).to(torch.uint8)
    yuv = torch.arange(total + 2 * (total // 4), dtype=torch.int64).to(torch.uint8)
    r2, y2 = rgb.clone(), yuv.clone()
    packed = pt.pack_bands(rgb, yuv, W, H, pt.band_row_ranges(H, band, 1, 0), torch)
    pt.gather_frame(packed, W, H, band, r2, y2)
    assert torch.equal(r2, rgb) and torch.equal(y2, yuv)


class _FakeTracer:
    """Stands in for DevicePathTracer in the orchestration test below: "rendering" a band copies its bytes out of a
    reference frame, a band's cost is a fixed function of its index (as the real per-tile trip counts are)."""

    def __init__(self, pt, ref_rgb, ref_yuv, fb, W, H, band, costs):
        self.pt, self.ref_rgb, self.ref_yuv, self.fb, self.W, self.H, self.band, self.costs = pt, ref_rgb, ref_yuv, fb, W, H, band, costs
        self.last = []
        self.launches = []               # every launch's band list, in launch order
        self.unread = []                 # launches whose costs have not been read yet (frames complete in launch order)

    def _render(self, bands):
        ranges = self.pt.bands_to_ranges(self.H, self.band, bands)
        self.pt.unpack_bands(self.pt.pack_bands(self.ref_rgb, self.ref_yuv, self.W, self.H, ranges, torch),
                             self.fb.rgb, self.fb.yuv, self.W, self.H, ranges)
        self.last = list(bands)
        self.launches.append(list(bands))
        self.unread.append(list(bands))

    def renderBandsAsync(self, band_rows, world, rank, stream=None):
        self._render(list(range(rank, (self.H + band_rows - 1) // band_rows, world)))

    def renderBandListAsync(self, band_rows, bands, stream=None):
        self._render(bands)

    def synchronizeStream(self, s):
        pass

    def setFramebuffer(self, fb):
        self.fb = fb

    def bandCosts(self, stream=None):
        # frames in flight: the costs asked for are those of the frame being COMPLETED, the oldest launch not read yet
        bands = self.unread.pop(0) if self.unread else self.last
        return [self.costs[b] for b in bands]

    def close(self):
        pass


class _FakeFb:
    pass


def _order_worker(rank, world, port, W, H, band, out_dir):
    """RenderManager.renderFrame under DSFL with every torch.distributed call of the frame logged: RCCL runs the
    point-to-point batches and the collectives of one group on one communicator / one stream, so the SEQUENCE of calls
    must be the same on every rank (rank 0 posted recv, all-reduce while the peers posted all-reduce, send
    -> deadlock on the nccl backend, invisible on gloo)."""
    sys.path.insert(0, ROOT)
    pt = importlib.import_module(PKG)
    sched = importlib.import_module(PKG + ".scheduling")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    real_batch, real_allreduce = dist.batch_isend_irecv, dist.all_reduce

    def batch(ops):
        log.append("p2p")
        return real_batch(ops)

    def allreduce(*a, **k):
        log.append("all_reduce")
        return real_allreduce(*a, **k)

    dist.batch_isend_irecv, dist.all_reduce = batch, allreduce
    total = W * H
    rng = np.random.default_rng(7)
    ref_rgb = torch.from_numpy(rng.integers(0, 256, total * 3, dtype=np.uint8))
    ref_yuv = torch.from_numpy(rng.integers(0, 256, total + 2 * (total // 4), dtype=np.uint8))
    n_bands = (H + band - 1) // band
    mgr = object.__new__(sched.RenderManager)              # the orchestration only: no GPU, no DevicePathTracer
    mgr._torch = torch
    mgr.config_ = pt.RendererConfig(samplesPerPixel=1, recursionDepth=1, gpuNumber=world, resolution=pt.Resolution(W, H),
                                    showTasks=False, algorithmType=pt.DSFL)
    mgr.rank, mgr.world, mgr.band_rows, mgr.streams = rank, world, band, [None]
    fb = _FakeFb()
    fb.rgb, fb.yuv = torch.zeros_like(ref_rgb), torch.zeros_like(ref_yuv)
    mgr.framebuffer_ = fb
    mgr.tracer = _FakeTracer(pt, ref_rgb, ref_yuv, fb, W, H, band, _costs(n_bands, 11))
    mgr._make_tasks()
    frames = []
    for frame in range(3):
        log.clear()
        mgr.renderFrame()
        frames.append(list(log))
        if rank == 0:
            assert torch.equal(fb.rgb, ref_rgb) and torch.equal(fb.yuv, ref_yuv), "frame %d" % frame
            fb.rgb.zero_(); fb.yuv.zero_()
    assert not mgr.static_ and mgr.assignment_ == pt.lpt_assignment(_costs(n_bands, 11), world)   # the dealer took over
    mgr.reset()
    open(os.path.join(out_dir, "order_%d.txt" % rank), "w").write(repr(frames))
    dist.barrier()
    dist.destroy_process_group()


def _in_flight_worker(rank, world, port, W, H, band, fif, out_dir):
    """RenderManager with frames in flight (fixed split), orchestration only: every rank issues one point-to-point batch per
    COMPLETED frame, in launch order, and nothing else; every complete frame rank 0 hands out is whole; a ring buffer is
    not written again before the frame after the one it was handed out in."""
    import collections
    sys.path.insert(0, ROOT)
    pt = importlib.import_module(PKG)
    sched = importlib.import_module(PKG + ".scheduling")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    real_batch = dist.batch_isend_irecv

    def batch(ops):
        log.append("p2p")
        return real_batch(ops)

    dist.batch_isend_irecv = batch
    total = W * H
    rng = np.random.default_rng(7)
    ref_rgb = torch.from_numpy(rng.integers(0, 256, total * 3, dtype=np.uint8))
    ref_yuv = torch.from_numpy(rng.integers(0, 256, total + 2 * (total // 4), dtype=np.uint8))
    n_bands = (H + band - 1) // band
    mgr = object.__new__(sched.RenderManager)              # the orchestration only: no GPU, no DevicePathTracer
    mgr._torch = torch
    mgr.config_ = pt.RendererConfig(samplesPerPixel=1, recursionDepth=1, gpuNumber=world, resolution=pt.Resolution(W, H),
                                    showTasks=False, algorithmType=pt.FSFL)
    mgr.rank, mgr.world, mgr.band_rows = rank, world, band
    mgr.frames_in_flight, mgr.streams, mgr._inflight, mgr._launched = fif, [None] * fif, collections.deque(), 1
    mgr._ring = []
    for _ in range(fif + 1):
        fb = _FakeFb()
        fb.rgb, fb.yuv = torch.zeros_like(ref_rgb), torch.zeros_like(ref_yuv)
        mgr._ring.append(fb)
    mgr.framebuffer_ = mgr._ring[0]
    mgr.tracer = _FakeTracer(pt, ref_rgb, ref_yuv, mgr._ring[0], W, H, band, _costs(n_bands, 11))
    mgr._make_tasks()
    calls, handed = [], 0
    first_buffer = mgr.getFramebuffer()
    for frame in range(2 * fif + 3):
        log.clear()
        done = mgr.frames_completed
        mgr.renderFrame()
        if mgr.frames_completed == 0:                      # what a caller can read before the first frame is complete
            assert not mgr.frameCompleted() and mgr.getFramebuffer() is first_buffer
            assert not first_buffer.rgb.any() and not first_buffer.yuv.any(), "a frame is being rendered into the buffer handed out"
        calls.append(list(log))
        assert mgr.frames_completed - done == (1 if frame >= fif - 1 else 0)
        if rank == 0 and mgr.frames_completed > done:
            cur = mgr.getFramebuffer()
            assert cur is mgr._ring[(frame - fif + 2) % (fif + 1)]      # frame 0 went into ring slot 1: slot 0 (what getFramebuffer() hands
                                                                         # out before any frame is complete) is never being rendered into
            assert torch.equal(cur.rgb, ref_rgb) and torch.equal(cur.yuv, ref_yuv), "frame %d" % frame
            cur.rgb.zero_(); cur.yuv.zero_()               # the next frame in this buffer must be rendered and exchanged afresh
            handed += 1
    log.clear()
    mgr.flush()
    calls.append(list(log))
    assert mgr.frames_completed == 2 * fif + 3 and not mgr._inflight
    if rank == 0:
        cur = mgr.getFramebuffer()
        assert torch.equal(cur.rgb, ref_rgb) and torch.equal(cur.yuv, ref_yuv)
    open(os.path.join(out_dir, "calls_%d.txt" % rank), "w").write(repr(calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fif", [(2, 2), (3, 3), (8, 4)])      # (8, 4): what bench.py runs on a whole node
def test_frames_in_flight_exchange_one_batch_per_completed_frame_in_order(tmp_path, world, fif):
    mp.spawn(_in_flight_worker, args=(world, _free_port(), 64, 72, 8, fif, str(tmp_path)), nprocs=world, join=True)
    calls = [eval((tmp_path / ("calls_%d.txt" % r)).read_text()) for r in range(world)]
    want = [[] if f < fif - 1 else ["p2p"] for f in range(2 * fif + 3)] + [["p2p"] * (fif - 1)]
    for r in range(world):
        assert calls[r] == want, (r, calls[r], want)


@pytest.mark.parametrize("world", [2, 3])
def test_every_rank_enqueues_the_frames_collectives_in_the_same_order(tmp_path, world):
    mp.spawn(_order_worker, args=(world, _free_port(), 64, 72, 8, str(tmp_path)), nprocs=world, join=True)
    orders = [eval((tmp_path / ("order_%d.txt" % r)).read_text()) for r in range(world)]
    for r in range(world):
        assert orders[r] == orders[0], (r, orders[r], orders[0])
    assert orders[0][0] == ["p2p", "all_reduce"], orders[0]      # the exchange first, the cost all-reduce after it


def _dealer_in_flight_worker(rank, world, port, W, H, band, fif, out_dir):
    """RenderManager under DSFL WITH frames in flight, orchestration only: every frame in flight carries the
    band assignment it was launched with; per completed frame every rank issues its point-to-point batch and then the cost
    all-reduce, in launch order; the re-deal fed by frame i's costs applies from frame i + F on, on every rank alike; every
    frame rank 0 hands out is whole."""
    import collections
    sys.path.insert(0, ROOT)
    pt = importlib.import_module(PKG)
    sched = importlib.import_module(PKG + ".scheduling")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    real_batch, real_allreduce = dist.batch_isend_irecv, dist.all_reduce

    def batch(ops):
        log.append("p2p")
        return real_batch(ops)

    def allreduce(*a, **k):
        log.append("all_reduce")
        return real_allreduce(*a, **k)

    dist.batch_isend_irecv, dist.all_reduce = batch, allreduce
    total = W * H
    rng = np.random.default_rng(7)
    ref_rgb = torch.from_numpy(rng.integers(0, 256, total * 3, dtype=np.uint8))
    ref_yuv = torch.from_numpy(rng.integers(0, 256, total + 2 * (total // 4), dtype=np.uint8))
    n_bands = (H + band - 1) // band
    costs = _costs(n_bands, 11)
    mgr = object.__new__(sched.RenderManager)
    mgr._torch = torch
    mgr.config_ = pt.RendererConfig(samplesPerPixel=1, recursionDepth=1, gpuNumber=world, resolution=pt.Resolution(W, H),
                                    showTasks=False, algorithmType=pt.DSFL)
    mgr.rank, mgr.world, mgr.band_rows = rank, world, band
    mgr.frames_in_flight, mgr.streams, mgr._inflight, mgr._launched = fif, [None] * fif, collections.deque(), 1
    mgr._ring = []
    for _ in range(fif + 1):
        fb = _FakeFb()
        fb.rgb, fb.yuv = torch.zeros_like(ref_rgb), torch.zeros_like(ref_yuv)
        mgr._ring.append(fb)
    mgr.framebuffer_ = mgr._ring[0]
    mgr.tracer = _FakeTracer(pt, ref_rgb, ref_yuv, mgr._ring[0], W, H, band, costs)
    mgr._make_tasks()
    n_frames = 2 * fif + 3
    calls = []
    for frame in range(n_frames):
        log.clear()
        done = mgr.frames_completed
        mgr.renderFrame()
        calls.append(list(log))
        assert mgr.frames_completed - done == (1 if frame >= fif - 1 else 0)
        if rank == 0 and mgr.frames_completed > done:
            cur = mgr.getFramebuffer()
            assert torch.equal(cur.rgb, ref_rgb) and torch.equal(cur.yuv, ref_yuv), "frame %d" % frame
            cur.rgb.zero_(); cur.yuv.zero_()
    log.clear()
    mgr.flush()
    calls.append(list(log))
    if rank == 0:
        cur = mgr.getFramebuffer()
        assert torch.equal(cur.rgb, ref_rgb) and torch.equal(cur.yuv, ref_yuv)
    # which assignment every frame was LAUNCHED with: the fixed split until the costs of frame 0 have gone round
    # (completed in call F - 1, all-reduced, applied at the next launch: frame F), the dealer's from then on
    static, dealt = pt.static_assignment(n_bands, world)[rank], pt.lpt_assignment(costs, world)[rank]
    assert static != dealt
    for frame, bands in enumerate(mgr.tracer.launches):
        assert bands == (static if frame < fif else dealt), (frame, bands)
    mgr.reset()
    open(os.path.join(out_dir, "dealer_calls_%d.txt" % rank), "w").write(repr(calls))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,fif", [(2, 2), (8, 3)])      # (8, 3): the shape VERDICT r3 item 5 names
def test_cost_fed_dealer_keeps_frames_in_flight(tmp_path, world, fif):
    mp.spawn(_dealer_in_flight_worker, args=(world, _free_port(), 64, 136, 8, fif, str(tmp_path)), nprocs=world, join=True)
    calls = [eval((tmp_path / ("dealer_calls_%d.txt" % r)).read_text()) for r in range(world)]
    per_frame = ["p2p", "all_reduce"]                        # the exchange first, the frame's cost all-reduce after it (RCCL: one order everywhere)
    want = [[] if f < fif - 1 else per_frame for f in range(2 * fif + 3)] + [per_frame * (fif - 1)]
    for r in range(world):
        assert calls[r] == want, (r, calls[r], want)
