"""ctypes binding of the C ABI declared in include/pt_amd.h (libptamd.so).

The library is the product: gfx950 kernels + host code.  If it is missing this module raises
immediately -- there is no Python or CPU fallback for the render path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PTAMD_LIBRARY: a diagnostic VARIANT build (tools/timeline.sh, tools/stamps.sh, tools/exp_lib.sh build theirs beside
# the product library and point here), so that no tool ever has to overwrite the product library in place.
LIB_PATH = os.environ.get("PTAMD_LIBRARY") or os.path.join(_HERE, "libptamd.so")
IS_VARIANT_BUILD = bool(os.environ.get("PTAMD_LIBRARY"))

PT_OK = 0
STATUS_NAMES = {
    0: "PT_OK", 1: "PT_ERR_INVALID", 2: "PT_ERR_HIP", 3: "PT_ERR_NO_LIGHTS", 4: "PT_ERR_STACK",
    5: "PT_ERR_IO", 6: "PT_ERR_UNSUPPORTED", 7: "PT_ERR_NO_SCENE", 8: "PT_ERR_NO_DEVICE",
    9: "PT_ERR_NOMEM", 10: "PT_ERR_INTERNAL",
}

# numpy mirrors of the PODs (HostScene.h:28-59, CameraConfig.h:5-17, DevicePathTracer.h:19-25)
VERTEX_DTYPE = np.dtype([("position", np.float32, 3), ("texCoords", np.float32, 2)])
TRIANGLE_DTYPE = np.dtype([("v0", VERTEX_DTYPE), ("v1", VERTEX_DTYPE), ("v2", VERTEX_DTYPE),
                           ("textureIdx", np.int32), ("materialIdx", np.int32)])
MATERIAL_DTYPE = np.dtype([("baseColor", np.float32, 3), ("baseColorTextureIdx", np.int32),
                           ("emissiveFactor", np.float32, 3), ("emissiveTextureIdx", np.int32)])
assert TRIANGLE_DTYPE.itemsize == 68 and MATERIAL_DTYPE.itemsize == 32
# The reference's TRUE Triangle (HostScene.h:28-47 as nvcc / hipcc lay it out): float2 is 8-byte aligned, so a Vertex is
# 24 bytes with texCoords at 16 and a Triangle 80 bytes.  TRIANGLE_DTYPE above is the C ABI's packed form.
REFERENCE_VERTEX_DTYPE = np.dtype({"names": ["position", "texCoords"], "formats": [(np.float32, 3), (np.float32, 2)],
                                   "offsets": [0, 16], "itemsize": 24})
REFERENCE_TRIANGLE_DTYPE = np.dtype({"names": ["v0", "v1", "v2", "textureIdx", "materialIdx"],
                                     "formats": [REFERENCE_VERTEX_DTYPE] * 3 + [np.int32, np.int32],
                                     "offsets": [0, 24, 48, 72, 76], "itemsize": 80})
# extension records (pt_set_legacy_scene)
LEGACY_MATERIAL_DTYPE = np.dtype([("kind", np.int32), ("albedo", np.float32, 3), ("param", np.float32)])
SPHERE_DTYPE = np.dtype([("center", np.float32, 3), ("radius", np.float32), ("materialIdx", np.int32)])
assert LEGACY_MATERIAL_DTYPE.itemsize == 20 and SPHERE_DTYPE.itemsize == 20


class Texture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("data", C.POINTER(C.c_float))]


class TriangleLayout(C.Structure):
    """pt_triangle_layout: stride of the caller's triangle record and the byte offset of each field in it."""
    _fields_ = [("stride", C.c_size_t), ("position", C.c_size_t * 3), ("texCoords", C.c_size_t * 3),
                ("textureIdx", C.c_size_t), ("materialIdx", C.c_size_t)]

    @staticmethod
    def of(dtype):
        """The layout of a numpy structured dtype with the fields of HostScene.h:41-47 Triangle."""
        l = TriangleLayout()
        l.stride = dtype.itemsize
        for i, v in enumerate(("v0", "v1", "v2")):
            vd, voff = dtype.fields[v][0], dtype.fields[v][1]
            l.position[i] = voff + vd.fields["position"][1]
            l.texCoords[i] = voff + vd.fields["texCoords"][1]
        l.textureIdx, l.materialIdx = dtype.fields["textureIdx"][1], dtype.fields["materialIdx"][1]
        return l


class Camera(C.Structure):
    _fields_ = [("lookFrom", C.c_float * 3), ("front", C.c_float * 3), ("vfov", C.c_float), ("hfov", C.c_float)]


class Task(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("offset_x", C.c_int), ("offset_y", C.c_int)]


class Counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("box_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("light_tests", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class BvhInfo(C.Structure):
    _fields_ = [("nodes", C.c_uint32), ("leaves", C.c_uint32), ("max_leaf", C.c_uint32), ("depth", C.c_uint32),
                ("max_stack", C.c_uint32), ("lights", C.c_uint32), ("triangles", C.c_uint32)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class PtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


_vp = C.c_void_p
_PROTOS = {
    # name: (restype, argtypes)
    "pt_last_error_string": (C.c_char_p, []),
    "pt_version": (C.c_char_p, []),
    "pt_device_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pt_device_destroy": (C.c_int, [_vp]),
    "pt_upload_scene": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(Texture), C.c_size_t]),
    "pt_upload_host_scene": (C.c_int, [_vp, _vp]),
    "pt_upload_scene_strided": (C.c_int, [_vp, _vp, C.c_size_t, C.POINTER(TriangleLayout), _vp, C.c_size_t, C.POINTER(Texture), C.c_size_t]),
    "pt_scene_create_strided": (C.c_int, [_vp, C.c_size_t, C.POINTER(TriangleLayout), _vp, C.c_size_t, C.POINTER(Texture), C.c_size_t, C.POINTER(_vp)]),
    "pt_triangles_repack": (C.c_int, [_vp, C.c_size_t, C.POINTER(TriangleLayout), _vp]),
    "pt_bind_framebuffer": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp]),
    "pt_set_thread_block_size": (C.c_int, [_vp, C.c_uint, C.c_uint]),
    "pt_set_traversal_mode": (C.c_int, [_vp, C.c_int]),
    "pt_set_bvh_mode": (C.c_int, [_vp, C.c_int]),
    "pt_render_task_async": (C.c_int, [_vp, C.POINTER(Task), C.POINTER(Camera), C.c_uint, C.c_uint, _vp]),
    "pt_render_bands_async": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(Camera), C.c_uint, C.c_uint, _vp]),
    "pt_render_band_list_async": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(Camera), C.c_uint, C.c_uint, _vp]),
    "pt_band_costs_read": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_int)]),
    "pt_stream_sync": (C.c_int, [_vp, _vp]),
    "pt_device_sync": (C.c_int, [_vp]),
    "pt_stream_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "pt_stream_destroy": (C.c_int, [_vp, _vp]),
    "pt_counters_read": (C.c_int, [_vp, C.POINTER(Counters)]),
    "pt_counters_reset": (C.c_int, [_vp]),
    "pt_debug_counters_read": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "pt_split_tiles_read": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "pt_kernel_time_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "pt_kernel_time_reset": (C.c_int, [_vp]),
    "pt_kernel_name_read": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "pt_bvh_info_read": (C.c_int, [_vp, C.POINTER(BvhInfo)]),
    "pt_bvh_copy": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "pt_host_bvh_build": (C.c_int, [_vp, C.c_size_t, C.POINTER(BvhInfo), _vp, _vp, _vp]),
    "pt_host_sah_build": (C.c_int, [_vp, C.c_size_t, C.POINTER(BvhInfo), _vp, _vp]),
    "pt_fb_alloc": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]),
    "pt_fb_free": (C.c_int, [_vp, _vp, _vp]),
    "pt_fb_download": (C.c_int, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "pt_fb_download_task": (C.c_int, [_vp, C.POINTER(Task), _vp, _vp, _vp]),
    "pt_fb_packed_size": (C.c_int, [C.c_int, C.c_int, C.POINTER(Task), C.c_int, C.POINTER(C.c_size_t)]),
    "pt_fb_pack_tasks": (C.c_int, [_vp, C.POINTER(Task), C.c_int, _vp, _vp]),
    "pt_fb_unpack_tasks": (C.c_int, [_vp, C.POINTER(Task), C.c_int, _vp, _vp]),
    "pt_gather_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(_vp)]),
    "pt_gather_destroy": (C.c_int, [_vp]),
    "pt_gather_transport": (C.c_int, [_vp, C.c_char_p, C.c_size_t]),
    "pt_gather_frame": (C.c_int, [_vp, C.POINTER(Task), C.POINTER(C.c_int), C.c_int, _vp, _vp]),
    "pt_set_tile_schedule": (C.c_int, [_vp, C.c_int]),
    "pt_set_cooperative_mode": (C.c_int, [_vp, C.c_int]),
    "pt_set_legacy_scene": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t]),
    "pt_scene_legacy_materials": (C.c_size_t, [_vp, _vp, C.c_size_t]),
    "pt_kat_arith": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t]),
    "pt_kat_exhaustive": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint64, _vp, _vp, C.c_int]),
    "pt_kat_closest_hit": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_size_t]),
    "pt_kat_walk_cost": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_size_t]),
    "pt_band_costs_read_on": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "pt_hw_queue_count": (C.c_int, []),
    "pt_scene_load_glb": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "pt_scene_load_obj": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "pt_scene_create": (C.c_int, [_vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(Texture), C.c_size_t, C.POINTER(_vp)]),
    "pt_scene_free": (None, [_vp]),
    "pt_scene_triangle_count": (C.c_size_t, [_vp]),
    "pt_scene_material_count": (C.c_size_t, [_vp]),
    "pt_scene_texture_count": (C.c_size_t, [_vp]),
    "pt_scene_triangles": (_vp, [_vp]),
    "pt_scene_materials": (_vp, [_vp]),
    "pt_scene_textures": (C.POINTER(Texture), [_vp]),
    "pt_scene_material_name": (C.c_char_p, [_vp, C.c_size_t]),
    "pt_scene_apply_cornell_convention": (C.c_int, [_vp]),
    "pt_scene_make_terrain": (C.c_int, [C.c_uint, C.POINTER(_vp)]),
    "pt_write_ppm": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _vp]),
    "pt_tasks_equal_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Task)]),
    "pt_tasks_equal_row": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(Task)]),
    "pt_tasks_row_bands": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Task), C.c_int,
                                      C.POINTER(C.c_int)]),
}

_lib = None


def lib():
    """The loaded C-ABI library; raises if the in-tree build is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). The HIP path is mandatory; there is no fallback." % LIB_PATH)
        # torch bundles its own HIP/HSA runtime (same soname, libamdhip64.so.7).  Two HIP
        # runtimes in one process cannot both own the GPU, so make sure torch's copy is the
        # one already loaded when libptamd.so's DT_NEEDED entry is resolved.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _PROTOS.items():
            fn = getattr(handle, name)      # AttributeError = header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(status):
    if status != PT_OK:
        msg = lib().pt_last_error_string()
        raise PtError(status, msg.decode("utf-8", "replace") if msg else "")


def np_ptr(arr):
    return C.c_void_p(arr.ctypes.data) if arr is not None else None
