"""MI355X-native hot path of 3DevApps/multi-gpu-path-tracer: the per-pixel path-tracing
megakernel behind the reference's DevicePathTracer / RenderManager / HostScene /
Framebuffer surface.  All compute lives in libptamd.so (hand-written HIP for gfx950,
C ABI in include/pt_amd.h); this package is the Python host mirror used by bench.py and
the tests.  Import it with importlib.import_module("multi-gpu-path-tracer_amd").
"""
import os as _os

# Frames in flight want one hardware queue per frame (include/pt_amd.h pt_hw_queue_count): the HIP runtime reads this once,
# at its first call -- importing this package before anything touches the GPU is early enough (libptamd.so sets the same
# default when it is loaded, which for this package is at the first C-ABI call).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _capi as capi
from .host import (CameraConfig, DevicePathTracer, Framebuffer, HostScene, RendererConfig, RenderTask,
                   Resolution, SceneLoader, FSFL, DSFL, DSDL, host_bvh_build, host_sah_build, make_terrain_scene,
                   write_ppm)
from .scheduling import (BandGather, RenderManager, TaskGenerator, band_row_ranges, bands_to_ranges, gather_frame,
                         get_task_layout, lpt_assignment, pack_bands, pack_bands_strided, pack_rank_bands, packed_size,
                         row_band_tasks, static_assignment, uniform_bands, unpack_bands, unpack_bands_strided,
                         unpack_rank_bands)

__all__ = [
    "capi", "CameraConfig", "DevicePathTracer", "Framebuffer", "HostScene", "RendererConfig", "RenderTask",
    "Resolution", "SceneLoader", "FSFL", "DSFL", "DSDL", "host_bvh_build", "host_sah_build", "make_terrain_scene", "write_ppm",
    "RenderManager", "TaskGenerator", "band_row_ranges", "gather_frame", "get_task_layout", "pack_bands",
    "packed_size", "row_band_tasks", "unpack_bands", "pack_bands_strided", "unpack_bands_strided",
    "BandGather", "bands_to_ranges", "lpt_assignment", "pack_rank_bands", "static_assignment", "uniform_bands",
    "unpack_rank_bands",
]
