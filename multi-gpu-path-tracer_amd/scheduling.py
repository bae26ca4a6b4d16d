"""Task generation and the frame orchestrator (RenderManager) for 1..N GPUs.

Reference: Scheduling/TaskGenerator.h:46-80, RenderManager.h:42-59 (task layout), :76-112
(setup), :410-431 (renderFrame), StreamThread.h:64-104 (worker).  The reference runs one
process with a thread per (GPU, stream) and lets CUDA managed memory "gather" the frame.
Here: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI, or "gloo" in
CPU tests), the scene is replicated per rank, the frame is cut into full-width row bands
dealt round-robin to ranks (results are partition-invariant: per-pixel seeds, no cross-pixel
state), and the only exchange step is ONE gather of each rank's band bytes to rank 0.
"""
import ctypes as C
import time
from typing import List, Optional

import numpy as np

from . import _capi as capi
from .host import (CameraConfig, DevicePathTracer, Framebuffer, HostScene, RendererConfig, RenderTask,
                   Resolution)


class TaskGenerator:
    """TaskGenerator (Scheduling/TaskGenerator.h)."""

    def generateEqualTasks(self, task_count: int, *args) -> List[RenderTask]:
        out = (capi.Task * task_count)()
        if len(args) == 2:                      # (task_count, width, height), :46-55
            width, height = args
            capi.check(capi.lib().pt_tasks_equal_row(task_count, width, height, out))
        else:                                   # (taskCount, taskLayout, width, height), :58-80
            layout, width, height = args
            max_in_row = max(len(r) for r in layout)
            capi.check(capi.lib().pt_tasks_equal_layout(task_count, max_in_row, width, height, out))
        return [RenderTask(t.width, t.height, t.offset_x, t.offset_y) for t in out]


def get_task_layout(thread_count: int, max_tasks_in_row: int) -> List[List[int]]:
    """RenderManager::getTaskLayout, RenderManager.h:42-59."""
    layout, task = [], 0
    while task < thread_count:
        layout.append([])
        in_row = 0
        while in_row < max_tasks_in_row and task < thread_count:
            layout[-1].append(task)
            task += 1
            in_row += 1
    return layout


def row_band_tasks(width: int, height: int, band_rows: int, n_ranks: int, rank: int) -> List[RenderTask]:
    """Bands of `band_rows` framebuffer rows; band b belongs to rank b % n_ranks."""
    n_bands = (height + band_rows - 1) // band_rows
    out = (capi.Task * max(n_bands, 1))()
    n = C.c_int()
    capi.check(capi.lib().pt_tasks_row_bands(width, height, band_rows, n_ranks, rank, out, n_bands, C.byref(n)))
    return [RenderTask(out[i].width, out[i].height, out[i].offset_x, out[i].offset_y) for i in range(n.value)]


def band_row_ranges(height: int, band_rows: int, n_ranks: int, rank: int):
    """Framebuffer row ranges [start, end) (row 0 = top) owned by `rank`, in band order."""
    n_bands = (height + band_rows - 1) // band_rows
    return [(b * band_rows, min(height, (b + 1) * band_rows)) for b in range(rank, n_bands, n_ranks)]


def _chroma_rows(r0: int, r1: int, height: int, half: int):
    """Chroma rows owned by framebuffer rows [r0, r1): those of its even rows."""
    c1 = min((r1 + 1) // 2, (height // 2) if half else 0)
    c0 = min((r0 + 1) // 2, c1)
    return c0, c1


def pack_bands(rgb_flat, yuv_flat, width: int, height: int, ranges, xp):
    """This rank's band bytes, plane-major: RGB rows of all its bands, then Y rows, then U rows,
    then V rows (the chroma rows of a band are those of its even framebuffer rows).
    xp = torch or numpy."""
    total = width * height
    uv = total // 4
    half = width // 2
    rgb_p, y_p, u_p, v_p = [], [], [], []
    for (r0, r1) in ranges:
        rgb_p.append(rgb_flat[r0 * width * 3:r1 * width * 3])
        y_p.append(yuv_flat[r0 * width:r1 * width])
        c0, c1 = _chroma_rows(r0, r1, height, half)
        u_p.append(yuv_flat[total + c0 * half: total + c1 * half])
        v_p.append(yuv_flat[total + uv + c0 * half: total + uv + c1 * half])
    return xp_cat(rgb_p + y_p + u_p + v_p, xp)


def unpack_bands(packed, rgb_flat, yuv_flat, width: int, height: int, ranges):
    """Inverse of pack_bands: scatter a rank's packed bands into full-frame buffers."""
    total = width * height
    uv = total // 4
    half = width // 2
    pos = 0
    for (r0, r1) in ranges:
        n = (r1 - r0) * width * 3
        rgb_flat[r0 * width * 3:r1 * width * 3] = packed[pos:pos + n]; pos += n
    for (r0, r1) in ranges:
        n = (r1 - r0) * width
        yuv_flat[r0 * width:r1 * width] = packed[pos:pos + n]; pos += n
    for plane in (0, 1):
        for (r0, r1) in ranges:
            c0, c1 = _chroma_rows(r0, r1, height, half)
            n = (c1 - c0) * half
            base = total + plane * uv
            yuv_flat[base + c0 * half: base + c1 * half] = packed[pos:pos + n]; pos += n
    return pos


def packed_size(width: int, height: int, ranges) -> int:
    half = width // 2
    n = 0
    for (r0, r1) in ranges:
        c0, c1 = _chroma_rows(r0, r1, height, half)
        n += (r1 - r0) * width * 4 + 2 * (c1 - c0) * half
    return n


def xp_cat(parts, xp):
    if xp is np:
        return np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return xp.cat(parts) if parts else xp.zeros(0, dtype=xp.uint8)


def static_assignment(n_bands: int, world: int) -> List[List[int]]:
    """Band b belongs to rank b % world (FSFL: fixed split)."""
    return [list(range(r, n_bands, world)) for r in range(world)]


def lpt_assignment(costs, world: int) -> List[List[int]]:
    """Cost-fed dealing for the multi-rank path (BASELINE configs[4]'s "dynamic tile load-balancer"; the analogue of
    the reference's frame-to-frame DSFL feedback, RenderManager.h:334-408, which moves task borders toward the
    equal-time point): bands in order of decreasing cost of the PREVIOUS frame, each to the rank with the least load so
    far (longest processing time first).  Deterministic -- ties go to the lower band / lower rank -- so every rank
    computes the same assignment from the same all-reduced costs, and image-neutral (per-pixel seeds).  Each rank's
    list comes back ascending, which also puts a partial last band of the frame last, as the kernel requires."""
    order = sorted(range(len(costs)), key=lambda b: (-int(costs[b]), b))
    load = [0] * world
    out = [[] for _ in range(world)]
    for b in order:
        r = min(range(world), key=lambda k: (load[k], k))
        out[r].append(b)
        load[r] += int(costs[b])
    return [sorted(l) for l in out]


def bands_to_ranges(height: int, band_rows: int, bands):
    """Framebuffer row ranges [start, end) (row 0 = top) of the listed bands, in list order."""
    return [(b * band_rows, min(height, (b + 1) * band_rows)) for b in bands]


def uniform_bands(width: int, height: int, band_rows: int) -> bool:
    """ONE predicate for both ends of the exchange (pack on every rank, unpack on rank 0): every band of the frame has
    the same number of bytes in each of the four planes, so a plane is a [n_bands, bytes] matrix and a rank's bands are
    a row selection of it.  Otherwise (a partial last band; odd sizes never get here, pt_bind_framebuffer refuses them)
    both ends take the generic per-band slices."""
    return height % band_rows == 0 and band_rows % 2 == 0 and width % 2 == 0 and height % 2 == 0


def _plane_views(rgb_flat, yuv_flat, width: int, height: int, band_rows: int):
    """The four planes as [n_bands, bytes-per-band] views (needs height % band_rows == 0)."""
    nb = height // band_rows
    total = width * height
    uv = total // 4
    return (rgb_flat.view(nb, -1), yuv_flat[:total].view(nb, -1),
            yuv_flat[total:total + uv].view(nb, -1), yuv_flat[total + uv:total + 2 * uv].view(nb, -1))


def pack_bands_strided(rgb_flat, yuv_flat, width: int, height: int, band_rows: int, world: int, rank: int):
    """pack_bands for torch tensors when band_rows divides the height: four strided gathers
    (same byte layout as pack_bands) instead of one slice per band and plane."""
    import torch
    return torch.cat([p[rank::world].reshape(-1) for p in _plane_views(rgb_flat, yuv_flat, width, height, band_rows)])


def unpack_bands_strided(packed, rgb_flat, yuv_flat, width: int, height: int, band_rows: int, world: int, rank: int):
    pos = 0
    for p in _plane_views(rgb_flat, yuv_flat, width, height, band_rows):
        dst = p[rank::world]
        n = dst.shape[0] * dst.shape[1]
        dst.copy_(packed[pos:pos + n].view(dst.shape[0], dst.shape[1]))
        pos += n
    return pos


_BAND_INDEX_CACHE = {}


def _band_index(bands, device):
    """The band list as an index tensor on `device`, made once per (list, device): building it is a blocking host-to-device
    copy, and with the fixed split the same eight lists come back every frame (rank 0 unpacks N - 1 of them per frame)."""
    import torch
    key = (tuple(int(b) for b in bands), str(device))
    idx = _BAND_INDEX_CACHE.get(key)
    if idx is None:
        if len(_BAND_INDEX_CACHE) >= 256:              # cost-fed dealers make new lists every frame: keep the table small
            _BAND_INDEX_CACHE.clear()
        idx = _BAND_INDEX_CACHE[key] = torch.as_tensor(list(key[0]), dtype=torch.long, device=device)
    return idx


def pack_bands_indexed(rgb_flat, yuv_flat, width: int, height: int, band_rows: int, bands):
    """pack_bands for torch tensors and an arbitrary band list when uniform_bands(): four row selections."""
    import torch
    idx = _band_index(bands, rgb_flat.device)
    return torch.cat([p.index_select(0, idx).reshape(-1) for p in _plane_views(rgb_flat, yuv_flat, width, height, band_rows)])


def unpack_bands_indexed(packed, rgb_flat, yuv_flat, width: int, height: int, band_rows: int, bands):
    idx = _band_index(bands, rgb_flat.device)
    pos = 0
    for p in _plane_views(rgb_flat, yuv_flat, width, height, band_rows):
        n = len(bands) * p.shape[1]
        p.index_copy_(0, idx, packed[pos:pos + n].view(len(bands), p.shape[1]))
        pos += n
    return pos


def pack_rank_bands(rgb_flat, yuv_flat, width: int, height: int, band_rows: int, bands, xp):
    """A rank's bands in exchange layout (plane-major: RGB rows of all its bands, then Y, U, V rows)."""
    if xp is not np and uniform_bands(width, height, band_rows):
        return pack_bands_indexed(rgb_flat, yuv_flat, width, height, band_rows, bands)
    return pack_bands(rgb_flat, yuv_flat, width, height, bands_to_ranges(height, band_rows, bands), xp)


def unpack_rank_bands(packed, rgb_flat, yuv_flat, width: int, height: int, band_rows: int, bands):
    if not isinstance(rgb_flat, np.ndarray) and uniform_bands(width, height, band_rows):
        return unpack_bands_indexed(packed, rgb_flat, yuv_flat, width, height, band_rows, bands)
    return unpack_bands(packed, rgb_flat, yuv_flat, width, height, bands_to_ranges(height, band_rows, bands))


class BandGather:
    """The path's one exchange step for an arbitrary band assignment: every rank's bands travel to rank 0 at their
    TRUE size (point-to-point, one grouped batch of receives on rank 0 -- ncclGroupStart / ncclRecv x (N-1) /
    ncclGroupEnd on RCCL; no padding to the largest share, nothing reduced, each peer on its own xGMI link to the
    root).  Rank 0 posts its receives BEFORE it renders its own bands (post()), so the peers' bytes land while it is
    still rendering, and scatters them into the frame afterwards (finish()).  Other ranks just send (finish())."""

    def __init__(self, width: int, height: int, band_rows: int, assignment, like, group=None):
        import torch
        import torch.distributed as dist
        self.width, self.height, self.band_rows, self.assignment, self.group = width, height, band_rows, assignment, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.sizes = [packed_size(width, height, bands_to_ranges(height, band_rows, bands)) for bands in assignment]
        # RCCL ("nccl") moves device buffers directly over xGMI; with the CPU-only "gloo" backend (tests, rehearsals
        # on one GPU) the same bytes are staged through host memory
        self.xdev = like.device if (self.world > 1 and dist.get_backend(group) == "nccl") else torch.device("cpu")
        self.recv = None
        self.reqs = []

    def post(self):
        import torch
        import torch.distributed as dist
        if self.world == 1 or self.rank != 0:
            return
        self.recv = {r: torch.empty(self.sizes[r], dtype=torch.uint8, device=self.xdev)
                     for r in range(1, self.world) if self.sizes[r] > 0}
        ops = [dist.P2POp(dist.irecv, buf, r, group=self.group) for r, buf in self.recv.items()]
        self.reqs = dist.batch_isend_irecv(ops) if ops else []

    def finish(self, packed, rgb_flat, yuv_flat):
        import torch.distributed as dist
        if self.world == 1:
            unpack_rank_bands(packed, rgb_flat, yuv_flat, self.width, self.height, self.band_rows, self.assignment[0])
            return
        if self.rank != 0:
            if self.sizes[self.rank] > 0:
                assert packed.numel() == self.sizes[self.rank]
                for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, packed.to(self.xdev), 0, group=self.group)]):
                    req.wait()
            return
        if self.recv is None:
            self.post()
        for req in self.reqs:
            req.wait()
        for r, buf in self.recv.items():                 # rank 0's own bands are already in place
            unpack_rank_bands(buf.to(rgb_flat.device), rgb_flat, yuv_flat, self.width, self.height, self.band_rows,
                              self.assignment[r])
        self.recv, self.reqs = None, []


def gather_frame(packed, width: int, height: int, band_rows: int, rgb_flat, yuv_flat, group=None, assignment=None):
    """The path's one exchange step: every rank contributes its packed bands (static b % world split unless an
    `assignment` -- one band list per rank -- is given), rank 0 scatters them into the full frame (BandGather)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_bands = (height + band_rows - 1) // band_rows
    g = BandGather(width, height, band_rows, assignment or static_assignment(n_bands, world), rgb_flat, group)
    g.post()
    g.finish(packed, rgb_flat, yuv_flat)


class RenderManager:
    """RenderManager (RenderManager.h:27-659) for the render path: owns the Framebuffer and
    this rank's DevicePathTracer, renders this rank's bands, gathers to rank 0.

    `renderFrame()` = RenderManager::renderFrame (:410-431) + the gather that replaces UVM.
    config.algorithmType picks the dealer of the multi-rank path: FSFL = fixed split (band b to rank b % N, the
    reference's fixed-size tasks, TaskGenerator.h:58-80); DSFL / DSDL = the reference's frame-to-frame feedback
    (:264-408 adapt task rectangles from last frame's task times) in this path's form: the previous frame's per-band
    costs, summed over ranks with one small all-reduce, re-deal the bands longest-processing-time-first
    (lpt_assignment).  Images are identical under every dealer.
    """

    frames_in_flight = 1          # (class defaults: one frame at a time, nothing in flight, no host copies)
    frames_completed = 0
    _inflight = ()
    host_frames = False
    _pending_costs = None
    record_spans = False          # bench.py: a pair of timing events around every frame's launch, on the launch's stream

    def __init__(self, config: RendererConfig, hScene: HostScene, cameraConfig: CameraConfig,
                 device_idx: int = 0, band_rows: int = 8, streams_per_gpu: Optional[int] = None,
                 bvh_mode: int = 0, frames_in_flight: int = 1, host_frames: bool = False):
        import collections
        import torch
        import torch.distributed as dist
        self._torch = torch
        self.config_ = config
        self.hScene_ = hScene
        self.cameraConfig_ = cameraConfig
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.band_rows = band_rows
        self.device_idx = device_idx
        self.framebuffer_ = Framebuffer(config.resolution, device="cuda:%d" % device_idx)
        self.tracer = DevicePathTracer(device_idx, config.samplesPerPixel, config.recursionDepth,
                                       config.threadBlockSize, hScene, self.framebuffer_, cameraConfig,
                                       bvhMode=bvh_mode)        # 1 = opt-in SAH tree (not the reference's traversal)
        n_streams = streams_per_gpu or config.streamsPerGpu
        self.streams = [torch.cuda.Stream(device=device_idx) for _ in range(max(1, n_streams))]
        # FRAMES IN FLIGHT (no counterpart in the reference, whose renderFrame ends on a barrier, RenderManager.h:410-431).
        # A rank's share of a frame has fewer tiles than its GPU has wave slots and lasts as long as its dearest pixels'
        # chains of dependent trips (a pixel's samples are one RNG stream): for most of that time most of the device idles.
        # With F > 1 renderFrame() launches frame i and then completes frame i - F + 1: up to F launches are queued on the
        # device, each in its own framebuffer on its own stream (the library keeps a tile schedule per framebuffer), and the
        # waves of the younger ones take the slots the oldest one's last chains leave empty.  A frame's latency grows, the
        # frame rate approaches what the work allows (an eighth of C2: 13.7 ms per frame alone, 10.3 / 8.8 ms with 2 / 3 in
        # flight; profiles/r03_frames_in_flight.json).  F + 1 framebuffers form a ring, so the last COMPLETE frame
        # (getFramebuffer / getCurrentFrame) stays untouched until the next renderFrame(); frame 0 goes into ring slot 1, so
        # that what getFramebuffer() hands out before any frame is complete (slot 0) is never a buffer being rendered into
        # (frameCompleted() says whether a frame is there yet; getCurrentFrame() completes one first if not).  Images are the
        # same frames in the same order.  The cost-fed dealers (DSFL / DSDL) pipeline too: every frame in flight carries the
        # band assignment it was launched with, the costs of a completed frame go into their all-reduce right after its
        # exchange, and the re-deal they feed applies from the next launch on -- F - 1 frames after the frame it was
        # measured on, on every rank alike (the wait for the all-reduce is at a fixed place in the call sequence).
        # More than three launches only overlap when each has a hardware queue of its own: the HIP runtime maps the streams
        # of a process onto 4 queues unless GPU_MAX_HW_QUEUES says otherwise BEFORE the runtime initialises (bench.py sets 8).
        self.frames_in_flight = max(1, int(frames_in_flight))
        hwq = capi.lib().pt_hw_queue_count()      # launches only overlap while each has a hardware queue of its own
        if self.frames_in_flight > max(1, hwq):
            self.frames_in_flight = max(1, hwq)
        self._inflight = collections.deque()      # (framebuffer, stream, assignment, static) of launched, not yet completed frames, oldest first
        self._launched = 1 if self.frames_in_flight > 1 else 0      # frame 0 renders into ring slot 1 (see above)
        self.frames_completed = 0                 # how many frames have been completed so far
        # HOST-VISIBLE FRAMES (a frame includes "gather to host-visible RGB"; the reference's host reads the managed
        # framebuffer after renderFrame, Framebuffer.h:27-35).  With host_frames every completed frame is copied, RGB + YUV,
        # into pinned host memory on a stream of its own, under the rendering of the next frame(s); hostFrame() waits for the
        # latest copy.  Rank 0 only (it owns the gathered frame).
        self.host_frames = bool(host_frames) and self.rank == 0
        self._pins, self._delivered, self._host_latest = [], 0, -1
        if self.host_frames:
            self._copy_stream = torch.cuda.Stream(device=device_idx)
        if self.frames_in_flight > 1:
            self._ring = [self.framebuffer_] + [Framebuffer(config.resolution, device="cuda:%d" % device_idx)
                                                for _ in range(self.frames_in_flight)]
            self.streams = [torch.cuda.Stream(device=device_idx) for _ in range(self.frames_in_flight)]
        self._make_tasks()
        self.last_frame_ms = 0.0

    def _make_tasks(self):
        res = self.config_.resolution
        self.n_bands_ = (res.height + self.band_rows - 1) // self.band_rows
        self.assignment_ = static_assignment(self.n_bands_, self.world)
        self.static_ = True                       # assignment_ is the b % N split (single strided launch)
        self.band_costs_ = None                   # previous frame's cost per band of the frame, all ranks
        if getattr(self, "_pending_costs", None) is not None:
            self._pending_costs[1].wait()         # a cost exchange of the old geometry: complete it, drop it
        self._pending_costs = None
        self._refresh_tasks()

    def _refresh_tasks(self):
        res = self.config_.resolution
        mine = self.assignment_[self.rank]
        self.ranges_ = bands_to_ranges(res.height, self.band_rows, mine)
        self.renderTasks_ = [RenderTask(res.width, r1 - r0, 0, res.height - r1) for (r0, r1) in self.ranges_]
        # adjacent bands of one rank (world == 1) are merged into one launch
        if self.world == 1:
            self.renderTasks_ = [RenderTask(res.width, res.height, 0, 0)]

    def dynamic(self) -> bool:
        from .host import FSFL
        return self.world > 1 and self.config_.algorithmType != FSFL

    # -- setters mirror RenderManager.h:214-236
    def setSamplesPerPixel(self, samples: int):
        self.config_.samplesPerPixel = samples
        self.tracer.setSamplesPerPixel(samples)

    def setRecursionDepth(self, depth: int):
        self.config_.recursionDepth = depth
        self.tracer.setRecursionDepth(depth)

    def setResolution(self, res: Resolution):
        self.flush()
        self.config_.resolution = res
        if self.frames_in_flight > 1:
            for fb in self._ring:
                fb.setResolution(res)
        self.framebuffer_.setResolution(res)
        self.tracer.setFramebuffer(self.framebuffer_)
        self._make_tasks()

    def getFramebuffer(self) -> Framebuffer:
        return self.framebuffer_

    def frameCompleted(self) -> bool:
        """Whether getFramebuffer() holds a rendered frame yet (frames in flight: the first F - 1 renderFrame() calls only
        launch).  The C++ host's RenderManager::frameCompleted()."""
        return self.frames_completed > 0

    # -- host-visible frames
    def _deliver(self, fb):
        """Rank 0, host_frames: enqueue the copy of the completed frame `fb` into the next pinned pair."""
        if not self.host_frames:
            return
        torch = self._torch
        res = self.config_.resolution
        n_rgb, n_yuv = res.width * res.height * 3, res.width * res.height * 3 // 2
        want = max(2, self.frames_in_flight + 1)
        if len(self._pins) != want or self._pins[0][0].numel() != n_rgb:
            self._copy_stream.synchronize()
            self._pins = [(torch.empty(n_rgb, dtype=torch.uint8).pin_memory(), torch.empty(n_yuv, dtype=torch.uint8).pin_memory(),
                           torch.cuda.Event()) for _ in range(want)]
        k = self._delivered % len(self._pins)
        pin_rgb, pin_yuv, ev = self._pins[k]
        cs = self._copy_stream
        cs.wait_stream(torch.cuda.current_stream(fb.rgb.device))      # the exchange's unpacking ran on the current stream
        with torch.cuda.stream(cs):
            pin_rgb.copy_(fb.rgb.reshape(-1), non_blocking=True)
            pin_yuv.copy_(fb.yuv.reshape(-1), non_blocking=True)
            ev.record(cs)
        fb._copy_event = ev                       # the next launch into this framebuffer waits for it
        self._delivered += 1
        self._host_latest = k

    def hostFrame(self):
        """(rgb, yuv) of the latest completed frame in pinned host memory, as numpy views (valid until F more frames are
        delivered); None before the first one.  Waits for that frame's copy only."""
        if not self.host_frames or self._host_latest < 0:
            return None
        pin_rgb, pin_yuv, ev = self._pins[self._host_latest]
        ev.synchronize()
        res = self.config_.resolution
        return pin_rgb.numpy().reshape(res.height, res.width, 3), pin_yuv.numpy()

    def resetSpans(self):
        self._spans = []

    def launchSpans(self):
        """[(start_ms, end_ms)] of every frame launched since resetSpans() with record_spans on, relative to the first one's
        start, from HIP events recorded on the launches' own streams (call after flush() / a device synchronisation).  With
        frames in flight the launches overlap: the device time they took TOGETHER is max(end) - min(start)."""
        spans = getattr(self, "_spans", [])
        if not spans:
            return []
        base = spans[0][0]
        return [(base.elapsed_time(a), base.elapsed_time(b)) for a, b in spans]

    def renderTasks(self, stream=None):
        """Launch this rank's share of the frame (StreamThread::threadMain, StreamThread.h:83-84):
        ONE kernel launch covering all of its row bands."""
        stream = stream if stream is not None else self.streams[0]
        if self.record_spans and stream is not None:
            a, b = self._torch.cuda.Event(enable_timing=True), self._torch.cuda.Event(enable_timing=True)
            a.record(stream)
            self._launch(stream)
            b.record(stream)
            if not hasattr(self, "_spans"):
                self._spans = []
            self._spans.append((a, b))
        else:
            self._launch(stream)

    def _launch(self, stream):
        ev = getattr(getattr(self.tracer, "framebuffer_", None), "_copy_event", None)      # its previous frame is still on its way to the host
        if ev is not None and stream is not None:
            stream.wait_event(ev)
        if self.world == 1:
            self.tracer.renderTaskAsync(self.renderTasks_[0], stream)
        elif self.static_:
            self.tracer.renderBandsAsync(self.band_rows, self.world, self.rank, stream)
        else:
            self.tracer.renderBandListAsync(self.band_rows, self.assignment_[self.rank], stream)

    def synchronize(self):
        for s in self.streams:
            self.tracer.synchronizeStream(s)

    def _begin_redeal(self, assignment=None, stream=None):
        """DSFL / DSDL in this path's form, first half (right after this rank's launch has finished): the costs of the
        bands it just rendered go into one ASYNCHRONOUS all-reduce (n_bands + 1 64-bit words) that gives every rank the
        whole vector.  Nobody waits here -- the peers are not held up by rank 0's unpacking, and rank 0 not by them.
        Frames in flight: `assignment` / `stream` are those of the COMPLETED frame (several launches are queued; the costs
        are read from the band launch that ran on that stream)."""
        import torch.distributed as dist
        torch = self._torch
        mine = (assignment if assignment is not None else self.assignment_)[self.rank]
        costs = self.tracer.bandCosts(stream=stream)
        have = torch.zeros(self.n_bands_ + 1, dtype=torch.int64)
        if len(costs) == len(mine):
            have[torch.as_tensor(mine, dtype=torch.long)] = torch.as_tensor(costs, dtype=torch.int64)
            have[self.n_bands_] = 1               # this rank had costs to report
        xdev = self.framebuffer_.rgb.device if dist.get_backend() == "nccl" else torch.device("cpu")
        have = have.to(xdev)
        self._pending_costs = (have, dist.all_reduce(have, async_op=True))

    def _finish_redeal(self):
        """Second half, at the start of the next frame: the same deterministic dealer runs on every rank over the summed
        costs (per-tile trip counts of the render kernel: the same whichever GPU rendered the band)."""
        have, work = self._pending_costs
        self._pending_costs = None
        work.wait()
        have = have.cpu()
        if int(have[self.n_bands_]) != self.world:
            return                                # some rank recorded nothing (tile schedule off / tiny launch): keep the split
        self.band_costs_ = have[:self.n_bands_].tolist()
        new = lpt_assignment(self.band_costs_, self.world)
        if new != self.assignment_:
            self.assignment_ = new
            self.static_ = False
            self._refresh_tasks()

    def _complete_oldest(self, gather: bool):
        """Frames in flight: wait for the oldest launched frame, run its exchange (with the band assignment IT was launched
        with), hand its costs to the dealer, make it the current frame."""
        fb, stream, assignment, _static = self._inflight.popleft()
        self.tracer.synchronizeStream(stream)
        if gather and self.world > 1:
            res = self.config_.resolution
            exchange = BandGather(res.width, res.height, self.band_rows, assignment, fb.rgb)
            # (rank 0's own bands are in place already: only the peers pack)
            packed = None if self.rank == 0 else pack_rank_bands(fb.rgb, fb.yuv, res.width, res.height, self.band_rows,
                                                                 assignment[self.rank], self._torch)
            exchange.finish(packed, fb.rgb, fb.yuv)       # every rank: frames in launch order, one batch per frame
            # ORDER on RCCL (see renderFrame): the frame's exchange first, its cost all-reduce after it, on every rank
            if self.dynamic():
                if self._pending_costs is not None:       # (flush(): several completions without a launch in between)
                    self._finish_redeal()
                self._begin_redeal(assignment, stream)
        self.framebuffer_ = fb
        self.frames_completed += 1
        self._deliver(fb)

    def flush(self, gather: bool = True):
        """Frames in flight: complete every launched frame, oldest first (no-op otherwise).  Afterwards
        getFramebuffer() / getCurrentFrame() is the last frame launched."""
        while self._inflight:
            self._complete_oldest(gather)
        if self.host_frames:
            self._copy_stream.synchronize()

    def _render_frame_in_flight(self, gather: bool):
        torch = self._torch
        F = self.frames_in_flight
        if self._pending_costs is not None:       # the re-deal fed by the frame completed in the previous call: from this launch on
            self._finish_redeal()
        fb = self._ring[self._launched % (F + 1)]
        stream = self.streams[self._launched % F]
        # whatever read or unpacked this buffer last (the current stream's work) is done before the kernel writes to it
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream(fb.rgb.device))
        self.tracer.setFramebuffer(fb)
        self.renderTasks(stream)
        self._inflight.append((fb, stream, self.assignment_, self.static_))
        self._launched += 1
        if len(self._inflight) >= F:
            self._complete_oldest(gather)

    def renderFrame(self, gather: bool = True):
        t0 = time.perf_counter()
        if self.frames_in_flight > 1:
            self._render_frame_in_flight(gather)
            self.last_frame_ms = (time.perf_counter() - t0) * 1e3
            return
        res = self.config_.resolution
        if self._pending_costs is not None:
            self._finish_redeal()                 # this frame's split, from the previous frame's costs
        exchange = None
        if gather and self.world > 1:
            exchange = BandGather(res.width, res.height, self.band_rows, self.assignment_, self.framebuffer_.rgb)
            exchange.post()                       # rank 0: receives are in flight while it renders its own bands
        self.renderTasks()
        self.synchronize()
        if exchange is not None:
            packed = None if self.rank == 0 else pack_rank_bands(self.framebuffer_.rgb, self.framebuffer_.yuv, res.width, res.height,
                                                                 self.band_rows, self.assignment_[self.rank], self._torch)
            exchange.finish(packed, self.framebuffer_.rgb, self.framebuffer_.yuv)
        # ORDER MATTERS on RCCL: point-to-point batches and collectives of one process group share one communicator and
        # one stream, so every rank must enqueue them in the same order -- the frame's exchange first (rank 0's grouped
        # receives were posted before rendering, the peers' sends are in finish()), the cost all-reduce after it.  The
        # other order on the peers (all-reduce, then send) would leave rank 0's receive waiting for a send queued behind
        # an all-reduce that waits for rank 0's, queued behind that receive.  tests/test_distributed_gloo.py pins it.
        if self.dynamic():
            self._begin_redeal()
        self.frames_completed += 1
        self._deliver(self.framebuffer_)
        self.last_frame_ms = (time.perf_counter() - t0) * 1e3

    def getCurrentFrame(self) -> np.ndarray:
        """The last complete frame.  Frames in flight: while none is complete yet (the first F - 1 calls of renderFrame()
        only launch) the frames launched so far are completed first -- the caller never sees a buffer that is being
        rendered into, nor a blank one."""
        if self.frames_completed == 0 and self._inflight:
            self.flush()
        return self.framebuffer_.rgb_host()

    def getYUVFrame(self) -> np.ndarray:
        if self.frames_completed == 0 and self._inflight:
            self.flush()
        return self.framebuffer_.yuv_host()

    def reset(self):
        self.flush(gather=False)
        if self._pending_costs is not None:
            self._pending_costs[1].wait()
            self._pending_costs = None
        self.tracer.close()
