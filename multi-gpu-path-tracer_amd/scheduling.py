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


def gather_frame(packed, width: int, height: int, band_rows: int, rgb_flat, yuv_flat, group=None):
    """The path's one exchange step: every rank contributes its packed bands, rank 0
    scatters them into the full frame.  Uses torch.distributed.gather (RCCL on GPUs)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    strided = height % band_rows == 0 and width % 2 == 0 and height % 2 == 0
    if world == 1:
        unpack_bands(packed, rgb_flat, yuv_flat, width, height, band_row_ranges(height, band_rows, 1, 0))
        return
    sizes = [packed_size(width, height, band_row_ranges(height, band_rows, world, r)) for r in range(world)]
    cap = max(sizes)
    # RCCL ("nccl") moves device buffers directly over xGMI; with the CPU-only "gloo" backend
    # (tests, rehearsals on one GPU) the same bytes are staged through host memory
    xdev = packed.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    if packed.numel() == cap and packed.device == xdev:
        send = packed
    else:
        send = torch.zeros(cap, dtype=torch.uint8, device=xdev)
        send[:packed.numel()] = packed
    recv = [torch.empty(cap, dtype=torch.uint8, device=xdev) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0, group=group)
    if rank == 0:
        for r in range(1, world):          # rank 0's own bands are already in place
            part = recv[r][:sizes[r]].to(rgb_flat.device)
            if strided:
                unpack_bands_strided(part, rgb_flat, yuv_flat, width, height, band_rows, world, r)
            else:
                unpack_bands(part, rgb_flat, yuv_flat, width, height, band_row_ranges(height, band_rows, world, r))


class RenderManager:
    """RenderManager (RenderManager.h:27-659) for the render path: owns the Framebuffer and
    this rank's DevicePathTracer, renders this rank's tasks, gathers to rank 0.

    `renderFrame()` = RenderManager::renderFrame (:410-431) + the gather that replaces UVM.
    The frame-to-frame adaptive policies DSFL / DSDL (:264-408) are not part of this path's
    single-frame metric; FSFL row bands (static, partition-invariant) are what runs.
    """

    def __init__(self, config: RendererConfig, hScene: HostScene, cameraConfig: CameraConfig,
                 device_idx: int = 0, band_rows: int = 8, streams_per_gpu: Optional[int] = None,
                 bvh_mode: int = 0):
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
        self._make_tasks()
        self.last_frame_ms = 0.0

    def _make_tasks(self):
        res = self.config_.resolution
        self.renderTasks_ = row_band_tasks(res.width, res.height, self.band_rows, self.world, self.rank)
        self.ranges_ = band_row_ranges(res.height, self.band_rows, self.world, self.rank)
        # adjacent bands of one rank (world == 1) are merged into one launch
        if self.world == 1:
            self.renderTasks_ = [RenderTask(res.width, res.height, 0, 0)]

    # -- setters mirror RenderManager.h:214-236
    def setSamplesPerPixel(self, samples: int):
        self.config_.samplesPerPixel = samples
        self.tracer.setSamplesPerPixel(samples)

    def setRecursionDepth(self, depth: int):
        self.config_.recursionDepth = depth
        self.tracer.setRecursionDepth(depth)

    def setResolution(self, res: Resolution):
        self.config_.resolution = res
        self.framebuffer_.setResolution(res)
        self.tracer.setFramebuffer(self.framebuffer_)
        self._make_tasks()

    def getFramebuffer(self) -> Framebuffer:
        return self.framebuffer_

    def renderTasks(self):
        """Launch this rank's share of the frame (StreamThread::threadMain, StreamThread.h:83-84):
        ONE kernel launch covering all of its row bands."""
        if self.world == 1:
            self.tracer.renderTaskAsync(self.renderTasks_[0], self.streams[0])
        else:
            self.tracer.renderBandsAsync(self.band_rows, self.world, self.rank, self.streams[0])

    def synchronize(self):
        for s in self.streams:
            self.tracer.synchronizeStream(s)

    def renderFrame(self, gather: bool = True):
        t0 = time.perf_counter()
        self.renderTasks()
        self.synchronize()
        if gather and self.world > 1:
            res = self.config_.resolution
            if res.height % self.band_rows == 0:
                packed = pack_bands_strided(self.framebuffer_.rgb, self.framebuffer_.yuv, res.width, res.height,
                                            self.band_rows, self.world, self.rank)
            else:
                packed = pack_bands(self.framebuffer_.rgb, self.framebuffer_.yuv, res.width, res.height,
                                    self.ranges_, self._torch)
            gather_frame(packed, res.width, res.height, self.band_rows, self.framebuffer_.rgb,
                         self.framebuffer_.yuv)
        self.last_frame_ms = (time.perf_counter() - t0) * 1e3

    def getCurrentFrame(self) -> np.ndarray:
        return self.framebuffer_.rgb_host()

    def getYUVFrame(self) -> np.ndarray:
        return self.framebuffer_.yuv_host()

    def reset(self):
        self.tracer.close()
