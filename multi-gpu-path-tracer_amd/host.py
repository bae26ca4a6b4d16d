"""Python face of the reference's plug-in surface for the hot path, over the C ABI.

Names, argument meaning and error behaviour follow the reference's C++ classes so that the
parity tests read like tests of the reference:

  SceneLoader.load / HostScene      HostScene.h:55-82, HostScene.cpp:98-139
  CameraConfig                      CameraConfig.h:5-17
  Resolution / RendererConfig       RendererConfig.h:6-37
  RenderTask                        DevicePathTracer.h:19-25
  Framebuffer                       Framebuffer.h:6-89
  DevicePathTracer                  DevicePathTracer.h:167-392

torch supplies device memory and streams only; all compute is in libptamd.so.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _capi as capi


# ----------------------------------------------------------------------------- scene --
class HostScene:
    """HostScene (HostScene.h:55-59): triangles, textures, materials, owned by the library."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    def __del__(self):
        try:
            if self._h:
                capi.lib().pt_scene_free(self._h)
                self._h = None
        except Exception:
            pass

    def legacyMaterials(self) -> np.ndarray:
        """EXTENSION: LEGACY_MATERIAL_DTYPE records that obj_loader.h:65-96's name-prefix rule gives this scene's
        materials (OBJ + MTL input; all kind 0 otherwise), ready for DevicePathTracer.setLegacyScene."""
        n = capi.lib().pt_scene_legacy_materials(self._h, None, 0)
        out = np.zeros(n, capi.LEGACY_MATERIAL_DTYPE)
        capi.lib().pt_scene_legacy_materials(self._h, capi.np_ptr(out), n)
        return out

    @staticmethod
    def from_arrays(triangles: np.ndarray, materials: np.ndarray, textures: Optional[List[np.ndarray]] = None):
        # any record layout with the fields of HostScene.h:41-47 (e.g. capi.REFERENCE_TRIANGLE_DTYPE, the reference's true
        # 80-byte Triangle) goes through the strided entry point, field by field; the packed form is taken as it is
        strided = getattr(triangles, "dtype", None) is not None and triangles.dtype.names and triangles.dtype != capi.TRIANGLE_DTYPE
        tris = np.ascontiguousarray(triangles) if strided else np.ascontiguousarray(triangles, dtype=capi.TRIANGLE_DTYPE)
        mats = np.ascontiguousarray(materials, dtype=capi.MATERIAL_DTYPE)
        textures = textures or []
        keep = [np.ascontiguousarray(t, dtype=np.float32) for t in textures]
        tex = (capi.Texture * max(len(keep), 1))()
        for i, t in enumerate(keep):
            assert t.ndim == 3 and t.shape[2] == 3, "texture must be [height, width, 3] float texels (0..255)"
            tex[i].height, tex[i].width = t.shape[0], t.shape[1]
            tex[i].data = t.ctypes.data_as(C.POINTER(C.c_float))
        out = C.c_void_p()
        if strided:
            layout = capi.TriangleLayout.of(tris.dtype)
            capi.check(capi.lib().pt_scene_create_strided(capi.np_ptr(tris), len(tris), C.byref(layout), capi.np_ptr(mats),
                                                          len(mats), tex, len(keep), C.byref(out)))
        else:
            capi.check(capi.lib().pt_scene_create(capi.np_ptr(tris), len(tris), capi.np_ptr(mats), len(mats),
                                                  tex, len(keep), C.byref(out)))
        return HostScene(out.value)

    @property
    def triangles(self) -> np.ndarray:
        n = capi.lib().pt_scene_triangle_count(self._h)
        ptr = capi.lib().pt_scene_triangles(self._h)
        buf = (C.c_char * (n * capi.TRIANGLE_DTYPE.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=capi.TRIANGLE_DTYPE).copy()

    @property
    def materials(self) -> np.ndarray:
        n = capi.lib().pt_scene_material_count(self._h)
        ptr = capi.lib().pt_scene_materials(self._h)
        buf = (C.c_char * (n * capi.MATERIAL_DTYPE.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=capi.MATERIAL_DTYPE).copy()

    @property
    def material_names(self) -> List[str]:
        n = capi.lib().pt_scene_material_count(self._h)
        return [capi.lib().pt_scene_material_name(self._h, i).decode() for i in range(n)]

    @property
    def textures(self) -> List[np.ndarray]:
        n = capi.lib().pt_scene_texture_count(self._h)
        arr = capi.lib().pt_scene_textures(self._h)
        out = []
        for i in range(n):
            w, h = arr[i].width, arr[i].height
            out.append(np.ctypeslib.as_array(arr[i].data, shape=(h, w, 3)).copy())
        return out

    def applyCornellConvention(self):
        """Benchmark convention for models/cornell_box.glb."""
        capi.check(capi.lib().pt_scene_apply_cornell_convention(self._h))
        return self


class SceneLoader:
    """SceneLoader::load(std::string&), HostScene.cpp:98-139 (.glb); .obj via the C ABI too."""

    def load(self, path: str) -> HostScene:
        out = C.c_void_p()
        fn = capi.lib().pt_scene_load_obj if path.lower().endswith(".obj") else capi.lib().pt_scene_load_glb
        if not (path.lower().endswith(".obj") or path.lower().endswith(".glb") or path.lower().endswith(".gltf")):
            raise RuntimeError("Unsupported file type" + path)          # HostScene.cpp:134
        try:
            capi.check(fn(path.encode(), C.byref(out)))
        except capi.PtError as e:
            raise RuntimeError(str(e))                                   # loader throws std::runtime_error
        return HostScene(out.value)


def make_terrain_scene(grid: int) -> HostScene:
    out = C.c_void_p()
    capi.check(capi.lib().pt_scene_make_terrain(grid, C.byref(out)))
    return HostScene(out.value)


def write_ppm(path: str, rgb: np.ndarray):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[0], rgb.shape[1]
    capi.check(capi.lib().pt_write_ppm(path.encode(), w, h, capi.np_ptr(rgb)))


# ---------------------------------------------------------------------------- configs --
@dataclass
class CameraConfig:
    """CameraConfig.h:5-17."""
    lookFrom: Tuple[float, float, float]
    front: Tuple[float, float, float]
    vfov: float = 45.0
    hfov: float = 45.0

    def as_c(self) -> capi.Camera:
        c = capi.Camera()
        c.lookFrom[:] = [float(v) for v in self.lookFrom]
        c.front[:] = [float(v) for v in self.front]
        c.vfov, c.hfov = float(self.vfov), float(self.hfov)
        return c


@dataclass
class Resolution:
    width: int = 400
    height: int = 400


FSFL, DSFL, DSDL = 0, 1, 2   # SchedulingAlgorithmType, RendererConfig.h:12-17


@dataclass
class RendererConfig:
    """RendererConfig.h:19-37 defaults."""
    jobId: str = "0"
    samplesPerPixel: int = 10
    recursionDepth: int = 3
    modelPath: str = ""
    gpuNumber: int = 1
    streamsPerGpu: int = 1
    resolution: Resolution = field(default_factory=Resolution)
    algorithmType: int = FSFL
    threadBlockSize: Tuple[int, int] = (8, 8)
    vfov: float = 45.0
    hfov: float = 45.0
    maxTasksInRow: int = 2
    showTasks: bool = True
    kParam: int = 1


@dataclass
class RenderTask:
    """DevicePathTracer.h:19-25."""
    width: int
    height: int
    offset_x: int
    offset_y: int
    time: int = 0

    def as_c(self) -> capi.Task:
        return capi.Task(self.width, self.height, self.offset_x, self.offset_y)


# ------------------------------------------------------------------------ framebuffer --
class Framebuffer:
    """Framebuffer.h:6-89 without managed memory: RGB (3 B/px) and planar YUV420 buffers
    live in HBM as torch uint8 tensors; hosts read them after an explicit copy."""

    def __init__(self, res: Resolution, device="cuda:0"):
        import torch
        self._torch = torch
        self.device = torch.device(device)
        self.resolution_ = Resolution(0, 0)
        self.rgb = None
        self.yuv = None
        self.setResolution(res)

    def setResolution(self, res: Resolution):
        if self.rgb is not None and res.width == self.resolution_.width and res.height == self.resolution_.height:
            return
        total = res.width * res.height
        self.rgb = self._torch.zeros(total * 3, dtype=self._torch.uint8, device=self.device)
        self.yuv = self._torch.zeros(total + 2 * (total // 4), dtype=self._torch.uint8, device=self.device)
        self.resolution_ = Resolution(res.width, res.height)

    def getResolution(self) -> Resolution:
        return self.resolution_

    def getPixelCount(self) -> int:
        return self.resolution_.width * self.resolution_.height

    def getRGBPtr(self) -> int:
        return self.rgb.data_ptr()

    def getYUVPtr(self) -> int:
        return self.yuv.data_ptr()

    def rgb_host(self) -> np.ndarray:
        return self.rgb.cpu().numpy().reshape(self.resolution_.height, self.resolution_.width, 3)

    def yuv_host(self) -> np.ndarray:
        return self.yuv.cpu().numpy()


# ------------------------------------------------------------------ per-GPU engine ----
class DevicePathTracer:
    """DevicePathTracer (DevicePathTracer.h:167-392) on one MI355X through the C ABI."""

    def __init__(self, device_idx: int, samplesPerPixel: int, recursionDepth: int, threadBlockSize,
                 hostScene: HostScene, framebuffer: Framebuffer, cameraConfig: CameraConfig,
                 legacyMaterials=None, spheres=None, bvhMode: int = 0):
        self.device_idx_ = device_idx
        self.samplesPerPixel_ = samplesPerPixel
        self.recursionDepth_ = recursionDepth
        self.threadBlockSize_ = tuple(threadBlockSize)
        self.hostScene_ = hostScene
        self.cameraConfig_ = cameraConfig
        self.framebuffer_ = None
        h = C.c_void_p()
        capi.check(capi.lib().pt_device_create(device_idx, C.byref(h)))
        self._h = h
        capi.check(capi.lib().pt_set_thread_block_size(self._h, *self.threadBlockSize_))
        if legacyMaterials is not None or spheres is not None:      # extension, before the first upload
            self.setLegacyScene(legacyMaterials, spheres, reload=False)
        if bvhMode:                                                  # opt-in SAH tree, chosen before the one upload
            self.setBvhMode(bvhMode, reload=False)
        self.reloadWorld()
        self.reloadCamera()
        self.setFramebuffer(framebuffer)

    def close(self):
        if getattr(self, "_h", None):
            capi.lib().pt_device_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- DevicePathTracer.h:194-226
    def renderTaskAsync(self, task: RenderTask, stream=None):
        cam = self.cameraConfig_.as_c()
        t = task.as_c()
        capi.check(capi.lib().pt_render_task_async(self._h, C.byref(t), C.byref(cam), self.samplesPerPixel_,
                                                   self.recursionDepth_, _stream_ptr(stream)))

    def renderBandsAsync(self, band_rows: int, n_ranks: int, rank: int, stream=None):
        """All row bands of `rank` in one launch (the multi-GPU share of a frame)."""
        cam = self.cameraConfig_.as_c()
        capi.check(capi.lib().pt_render_bands_async(self._h, band_rows, n_ranks, rank, C.byref(cam),
                                                    self.samplesPerPixel_, self.recursionDepth_, _stream_ptr(stream)))

    def renderBandListAsync(self, band_rows: int, bands, stream=None):
        """The listed row bands (indices into the frame's bands, 0 = top) in one launch: what a dealer other than
        b % n_ranks hands this rank."""
        cam = self.cameraConfig_.as_c()
        arr = (C.c_int * max(len(bands), 1))(*[int(b) for b in bands])
        capi.check(capi.lib().pt_render_band_list_async(self._h, band_rows, arr, len(bands), C.byref(cam),
                                                        self.samplesPerPixel_, self.recursionDepth_, _stream_ptr(stream)))

    def bandCosts(self, max_bands: int = 8192, stream=None):
        """Costs of the bands of the latest row-band launch -- with `stream`: of the latest one enqueued on that stream
        (frames in flight) -- in launch order ([] when none were recorded)."""
        out = (C.c_uint64 * max_bands)()
        n = C.c_int()
        if stream is None:
            capi.check(capi.lib().pt_band_costs_read(self._h, out, max_bands, C.byref(n)))
        else:
            capi.check(capi.lib().pt_band_costs_read_on(self._h, _stream_ptr(stream), out, max_bands, C.byref(n)))
        return [int(out[i]) for i in range(n.value)]

    def waitForRenderTask(self):
        capi.check(capi.lib().pt_device_sync(self._h))

    def synchronizeStream(self, stream=None):
        capi.check(capi.lib().pt_stream_sync(self._h, _stream_ptr(stream)))

    # -- mutators, DevicePathTracer.h:230-370
    def reloadCamera(self):
        pass   # the camera basis is derived from cameraConfig_ at every launch (by value, :210)

    def reloadWorld(self):
        capi.check(capi.lib().pt_upload_host_scene(self._h, self.hostScene_._h))

    def uploadSceneStrided(self, triangles: np.ndarray, materials: np.ndarray, textures=None):
        """reloadWorld() for a caller that keeps its own triangle records (the reference's
        std::vector<Triangle> is 80-byte records): pt_upload_scene_strided with the array's own field offsets."""
        tris = np.ascontiguousarray(triangles)
        mats = np.ascontiguousarray(materials, dtype=capi.MATERIAL_DTYPE)
        keep = [np.ascontiguousarray(t, dtype=np.float32) for t in (textures or [])]
        tex = (capi.Texture * max(len(keep), 1))()
        for i, t in enumerate(keep):
            tex[i].height, tex[i].width = t.shape[0], t.shape[1]
            tex[i].data = t.ctypes.data_as(C.POINTER(C.c_float))
        layout = capi.TriangleLayout.of(tris.dtype)
        capi.check(capi.lib().pt_upload_scene_strided(self._h, capi.np_ptr(tris), len(tris), C.byref(layout),
                                                      capi.np_ptr(mats), len(mats), tex, len(keep)))

    def setFramebuffer(self, framebuffer: Framebuffer):
        self.framebuffer_ = framebuffer
        res = framebuffer.getResolution()
        capi.check(capi.lib().pt_bind_framebuffer(self._h, res.width, res.height,
                                                  C.c_void_p(framebuffer.getRGBPtr()),
                                                  C.c_void_p(framebuffer.getYUVPtr())))

    def setSamplesPerPixel(self, samplesPerPixel: int):
        self.samplesPerPixel_ = samplesPerPixel

    def setRecursionDepth(self, recursionDepth: int):
        self.recursionDepth_ = recursionDepth

    def setThreadBlockSize(self, threadBlockSize):
        self.threadBlockSize_ = tuple(threadBlockSize)
        capi.check(capi.lib().pt_set_thread_block_size(self._h, *self.threadBlockSize_))

    def setLegacyScene(self, materials=None, spheres=None, reload: bool = True):
        """EXTENSION (include/pt_amd.h pt_set_legacy_scene): one LEGACY_MATERIAL_DTYPE record per material
        (kind 0 keeps the UniversalMaterial) and SPHERE_DTYPE records; None / empty switches it off."""
        m = np.ascontiguousarray(materials if materials is not None else [], dtype=capi.LEGACY_MATERIAL_DTYPE)
        sp = np.ascontiguousarray(spheres if spheres is not None else [], dtype=capi.SPHERE_DTYPE)
        capi.check(capi.lib().pt_set_legacy_scene(self._h, capi.np_ptr(m) if len(m) else None, len(m),
                                                  capi.np_ptr(sp) if len(sp) else None, len(sp)))
        if reload:
            self.reloadWorld()

    def setBvhMode(self, mode: int, reload: bool = True):
        """0 = reference-topology BVH (default), 1 = opt-in SAH tree; takes effect at reloadWorld()."""
        capi.check(capi.lib().pt_set_bvh_mode(self._h, mode))
        if reload:
            self.reloadWorld()

    def setCooperativeMode(self, mode: int):
        """-1 automatic (default), 0 off, 1 on: four waves per tile sharing each leaf's triangles (big-leaf trees)."""
        capi.check(capi.lib().pt_set_cooperative_mode(self._h, mode))

    def setTileSchedule(self, mode: int):
        """0 raster order, 1 (default) tiles ordered by their cost in the previous launch of the same task."""
        capi.check(capi.lib().pt_set_tile_schedule(self._h, mode))

    def setTraversalMode(self, mode: int):
        """0 auto, 1 per-lane walk, 2 wave-uniform walk / path mask (<= 64 inner nodes),
        3 wave-uniform walk / LDS stack (<= 4096 inner nodes), 4 mask walk with dealt sparse leaves
        (<= 64 inner nodes, <= 1024 triangles), 5 stack walk with dealt sparse leaves (<= 4096 inner nodes,
        <= 65536 triangles)."""
        capi.check(capi.lib().pt_set_traversal_mode(self._h, mode))

    # -- additions: exact work counters and kernel time (the metric's inputs)
    def readCounters(self) -> dict:
        c = capi.Counters()
        capi.check(capi.lib().pt_counters_read(self._h, C.byref(c)))
        return c.as_dict()

    def readStamps(self):
        """Raw counter words 5..7 (diagnostic PT_STAMPS build only)."""
        raw = (C.c_uint64 * 8)()
        capi.check(capi.lib().pt_debug_counters_read(self._h, raw))
        return {"longest_wave_cycles": int(raw[5]), "sum_traversal_cycles": int(raw[6]), "sum_wave_cycles": int(raw[7]),
                "raw": [int(v) for v in raw]}

    def resetCounters(self):
        capi.check(capi.lib().pt_counters_reset(self._h))

    def kernelTime(self) -> Tuple[float, int]:
        ms = C.c_double()
        n = C.c_uint64()
        capi.check(capi.lib().pt_kernel_time_read(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def splitTiles(self) -> int:
        """Tiles the next launch of the latest rectangle splits over several waves (SAH kernel; waits for that launch)."""
        n = C.c_uint32()
        capi.check(capi.lib().pt_split_tiles_read(self._h, C.byref(n)))
        return n.value

    def kernelName(self) -> str:
        """Symbol of the kernel the latest render launch ran (what a rocprofv3 kernel trace shows)."""
        buf = C.create_string_buffer(256)
        capi.check(capi.lib().pt_kernel_name_read(self._h, buf, len(buf)))
        return buf.value.decode()

    def resetKernelTime(self):
        capi.check(capi.lib().pt_kernel_time_reset(self._h))

    def bvhInfo(self) -> dict:
        info = capi.BvhInfo()
        capi.check(capi.lib().pt_bvh_info_read(self._h, C.byref(info)))
        return info.as_dict()

    def bvhCopy(self):
        info = self.bvhInfo()
        bounds = np.zeros((info["nodes"], 6), np.float32)
        links = np.zeros((info["nodes"], 4), np.uint32)
        idx = np.zeros(info["triangles"], np.int32)
        lights = np.zeros(info["lights"], np.int32)
        capi.check(capi.lib().pt_bvh_copy(self._h, capi.np_ptr(bounds), capi.np_ptr(links), capi.np_ptr(idx),
                                          capi.np_ptr(lights)))
        return bounds, links, idx, lights

    def katArith(self, op: int, a: np.ndarray, b: Optional[np.ndarray] = None):
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32) if b is not None else None
        o0 = np.zeros_like(a)
        o1 = np.zeros_like(a)
        capi.check(capi.lib().pt_kat_arith(self._h, op, capi.np_ptr(a), capi.np_ptr(b), capi.np_ptr(o0),
                                           capi.np_ptr(o1), a.size))
        return o0, o1

    def katExhaustive(self, op: int, first_bits: int = 0, count: int = 1 << 32, max_examples: int = 16):
        """(number of differing inputs, example bit patterns) of pt_kat_exhaustive."""
        bad = C.c_uint64(0)
        ex = np.zeros(max(max_examples, 1), np.uint32)
        capi.check(capi.lib().pt_kat_exhaustive(self._h, op, first_bits, count, C.byref(bad), capi.np_ptr(ex),
                                                max_examples))
        return int(bad.value), ex[:min(int(bad.value), max_examples)]

    def katClosestHit(self, origins: np.ndarray, dirs: np.ndarray):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        t = np.zeros(len(o), np.float32)
        f = np.zeros(len(o), np.int32)
        capi.check(capi.lib().pt_kat_closest_hit(self._h, capi.np_ptr(o), capi.np_ptr(d), capi.np_ptr(t),
                                                 capi.np_ptr(f), len(o)))
        return t, f

    def katWalkCost(self, origins: np.ndarray, dirs: np.ndarray):
        """Box and triangle tests per ray of the walk katClosestHit runs (diagnostic)."""
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        nb = np.zeros(len(o), np.uint32)
        nt = np.zeros(len(o), np.uint32)
        capi.check(capi.lib().pt_kat_walk_cost(self._h, capi.np_ptr(o), capi.np_ptr(d), capi.np_ptr(nb), capi.np_ptr(nt), len(o)))
        return nb, nt


def _stream_ptr(stream):
    if stream is None:
        return None
    if isinstance(stream, int):
        return C.c_void_p(stream)
    return C.c_void_p(stream.cuda_stream)     # torch.cuda.Stream


def host_bvh_build(triangles: np.ndarray):
    """Host-only build of the reference-topology BVH (no GPU needed)."""
    tris = np.ascontiguousarray(triangles, dtype=capi.TRIANGLE_DTYPE)
    n = len(tris)
    info = capi.BvhInfo()
    bounds = np.zeros((2 * n - 1, 6), np.float32)
    links = np.zeros((2 * n - 1, 4), np.uint32)
    idx = np.zeros(n, np.int32)
    capi.check(capi.lib().pt_host_bvh_build(capi.np_ptr(tris), n, C.byref(info), capi.np_ptr(bounds),
                                            capi.np_ptr(links), capi.np_ptr(idx)))
    k = info.nodes
    return info.as_dict(), bounds[:k], links[:k], idx


def host_sah_build(triangles: np.ndarray):
    """Host-only build of the opt-in SAH tree: (info, pairs [n_pairs, 16] float32 view, order)."""
    tris = np.ascontiguousarray(triangles, dtype=capi.TRIANGLE_DTYPE)
    n = len(tris)
    info = capi.BvhInfo()
    pairs = np.zeros((max(n, 1), 16), np.float32)
    order = np.zeros(n, np.int32)
    capi.check(capi.lib().pt_host_sah_build(capi.np_ptr(tris), n, C.byref(info), capi.np_ptr(pairs), capi.np_ptr(order)))
    return info.as_dict(), pairs[:info.nodes], order
