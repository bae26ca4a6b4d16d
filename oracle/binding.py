"""ctypes binding of oracle/libpt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
package never imports this module.  See pt_oracle.c's header for the parity status
("parity unpinned" at the image level; structural pins from the survey probe).
"""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpt_oracle.so")

VERTEX_DTYPE = np.dtype([("position", np.float32, 3), ("texCoords", np.float32, 2)])
TRIANGLE_DTYPE = np.dtype([("v0", VERTEX_DTYPE), ("v1", VERTEX_DTYPE), ("v2", VERTEX_DTYPE),
                           ("textureIdx", np.int32), ("materialIdx", np.int32)])
MATERIAL_DTYPE = np.dtype([("baseColor", np.float32, 3), ("baseColorTextureIdx", np.int32),
                           ("emissiveFactor", np.float32, 3), ("emissiveTextureIdx", np.int32)])


class _Texture(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("data", C.POINTER(C.c_float))]


class _Camera(C.Structure):
    _fields_ = [("lookFrom", C.c_float * 3), ("front", C.c_float * 3), ("vfov", C.c_float), ("hfov", C.c_float)]


class _Task(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("offset_x", C.c_int), ("offset_y", C.c_int)]


class _Counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("box_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("light_tests", C.c_uint64), ("rng_draws", C.c_uint64),
                ("max_stack", C.c_uint64), ("nan_pixels", C.c_uint64)]


class _Rng(C.Structure):
    _fields_ = [("d", C.c_uint32), ("v", C.c_uint32 * 5)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s missing: run `make -C oracle`" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.ptor_scene_create.restype = C.c_int
        L.ptor_scene_create.argtypes = [vp, C.c_int, vp, C.c_int, C.POINTER(_Texture), C.c_int, C.POINTER(vp)]
        L.ptor_scene_destroy.argtypes = [vp]
        L.ptor_scene_set_legacy.restype = C.c_int
        L.ptor_scene_set_legacy.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.ptor_scene_node_count.restype = C.c_uint32
        L.ptor_scene_node_count.argtypes = [vp]
        L.ptor_scene_light_count.restype = C.c_int
        L.ptor_scene_light_count.argtypes = [vp]
        L.ptor_scene_copy_nodes.argtypes = [vp, vp, vp]
        L.ptor_scene_copy_indices.argtypes = [vp, vp]
        L.ptor_scene_copy_lights.argtypes = [vp, vp]
        L.ptor_render.restype = C.c_int
        L.ptor_render.argtypes = [vp, C.POINTER(_Camera), C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(_Task),
                                  vp, vp, C.POINTER(_Counters)]
        L.ptor_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ptor_xorwow_init.argtypes = [C.POINTER(_Rng), C.c_uint64]
        L.ptor_xorwow_next.restype = C.c_uint32
        L.ptor_xorwow_next.argtypes = [C.POINTER(_Rng)]
        L.ptor_kat_box_hit.restype = C.c_int
        L.ptor_kat_box_hit.argtypes = [vp, vp, vp]
        L.ptor_kat_tri_hit.restype = C.c_int
        L.ptor_kat_tri_hit.argtypes = [vp, vp, vp, C.c_float, C.c_float, vp]
        L.ptor_kat_closest_hit.restype = C.c_int
        L.ptor_kat_closest_hit.argtypes = [vp, vp, vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.ptor_kat_brute_closest.restype = C.c_int
        L.ptor_kat_brute_closest.argtypes = [vp, C.c_int, vp, vp, C.c_int, vp, vp, vp]
        _lib = L
    return _lib


class OracleScene:
    def __init__(self, triangles, materials, textures=None):
        self.tris = np.ascontiguousarray(triangles, dtype=TRIANGLE_DTYPE)
        self.mats = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self.texs = [np.ascontiguousarray(t, dtype=np.float32) for t in (textures or [])]
        tex = (_Texture * max(len(self.texs), 1))()
        for i, t in enumerate(self.texs):
            tex[i].height, tex[i].width = t.shape[0], t.shape[1]
            tex[i].data = t.ctypes.data_as(C.POINTER(C.c_float))
        h = C.c_void_p()
        rc = lib().ptor_scene_create(self.tris.ctypes.data, len(self.tris), self.mats.ctypes.data, len(self.mats),
                                     tex, len(self.texs), C.byref(h))
        if rc != 0:
            raise RuntimeError("ptor_scene_create failed: %d" % rc)
        self._h = h

    def __del__(self):
        try:
            if self._h:
                lib().ptor_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def set_legacy(self, materials=None, spheres=None):
        """Extension: legacy material kinds {kind, albedo[3], param} per material and spheres
        {center[3], radius, materialIdx}; None switches it off."""
        lm = np.dtype([("kind", np.int32), ("albedo", np.float32, 3), ("param", np.float32)])
        sd = np.dtype([("center", np.float32, 3), ("radius", np.float32), ("materialIdx", np.int32)])
        m = np.ascontiguousarray(materials if materials is not None else [], dtype=lm)
        sp = np.ascontiguousarray(spheres if spheres is not None else [], dtype=sd)
        rc = lib().ptor_scene_set_legacy(self._h, m.ctypes.data if len(m) else None, len(m),
                                         sp.ctypes.data if len(sp) else None, len(sp))
        if rc != 0:
            raise RuntimeError("ptor_scene_set_legacy failed: %d" % rc)

    def bvh(self):
        n = lib().ptor_scene_node_count(self._h)
        bounds = np.zeros((n, 6), np.float32)
        links = np.zeros((n, 4), np.uint32)
        idx = np.zeros(len(self.tris), np.int32)
        nl = lib().ptor_scene_light_count(self._h)
        lights = np.zeros(nl, np.int32)
        lib().ptor_scene_copy_nodes(self._h, bounds.ctypes.data, links.ctypes.data)
        lib().ptor_scene_copy_indices(self._h, idx.ctypes.data)
        if nl:
            lib().ptor_scene_copy_lights(self._h, lights.ctypes.data)
        return bounds, links, idx, lights

    def render(self, camera, width, height, spp, depth, task=None, threads=1, rgb=None, yuv=None):
        """camera = (lookFrom, front, vfov, hfov); task = (w, h, ox, oy) or None for the frame.
        Returns rgb [H,W,3], yuv (flat), counters dict.  threads > 1 splits the task into row
        strips (the image is partition-invariant)."""
        cam = _Camera()
        cam.lookFrom[:] = [float(v) for v in camera[0]]
        cam.front[:] = [float(v) for v in camera[1]]
        cam.vfov, cam.hfov = float(camera[2]), float(camera[3])
        total = width * height
        if rgb is None:
            rgb = np.zeros(total * 3, np.uint8)
        if yuv is None:
            yuv = np.zeros(total + 2 * (total // 4), np.uint8)
        tw, th, ox, oy = task if task is not None else (width, height, 0, 0)
        threads = max(1, min(threads, th))
        # eight strips per thread, handed out as the threads come free: rows differ many-fold in cost (the duck's rows
        # against the background's), and one strip per thread left most threads idle for most of a full-size frame
        n_strips = threads if threads == 1 else min(th, threads * 8)
        strips = []
        for i in range(n_strips):
            y0 = oy + (th * i) // n_strips
            y1 = oy + (th * (i + 1)) // n_strips
            if y1 > y0:
                strips.append(_Task(tw, y1 - y0, ox, y0))
        counters = [_Counters() for _ in strips]

        def run(k):
            return lib().ptor_render(self._h, C.byref(cam), width, height, spp, depth, C.byref(strips[k]),
                                     rgb.ctypes.data, yuv.ctypes.data, C.byref(counters[k]))

        if len(strips) == 1:
            rcs = [run(0)]
        else:
            with ThreadPoolExecutor(max_workers=threads) as ex:
                rcs = list(ex.map(run, range(len(strips))))
        if any(rcs):
            raise RuntimeError("ptor_render failed: %s" % rcs)
        tot = {}
        for name, _ in _Counters._fields_:
            vals = [int(getattr(c, name)) for c in counters]
            tot[name] = max(vals) if name == "max_stack" else sum(vals)
        return rgb.reshape(height, width, 3), yuv, tot

    def closest_hit(self, origin, direction):
        o = np.asarray(origin, np.float32)
        d = np.asarray(direction, np.float32)
        t = C.c_float()
        m = C.c_int()
        h = lib().ptor_kat_closest_hit(self._h, o.ctypes.data, d.ctypes.data, C.byref(t), C.byref(m))
        return h, t.value, m.value


def sincos(x):
    s = C.c_float()
    c = C.c_float()
    lib().ptor_sincosf(C.c_float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def xorwow_uniforms(seed, n):
    st = _Rng()
    lib().ptor_xorwow_init(C.byref(st), seed)
    out = np.zeros(n, np.float32)
    raw = np.zeros(n, np.uint32)
    for i in range(n):
        x = lib().ptor_xorwow_next(C.byref(st))
        raw[i] = x
        out[i] = np.float32(x) * np.float32(2.3283064e-10) + np.float32(2.3283064e-10) / np.float32(2.0)
    return raw, out


def box_hit(b6, o, d):
    b = np.asarray(b6, np.float32); o = np.asarray(o, np.float32); d = np.asarray(d, np.float32)
    return lib().ptor_kat_box_hit(b.ctypes.data, o.ctypes.data, d.ctypes.data)


def tri_hit(v9, o, d, tmin=0.001, tmax=3.402823466e+38):
    v = np.asarray(v9, np.float32).reshape(9); o = np.asarray(o, np.float32); d = np.asarray(d, np.float32)
    out = np.zeros(3, np.float32)
    h = lib().ptor_kat_tri_hit(v.ctypes.data, o.ctypes.data, d.ctypes.data, tmin, tmax, out.ctypes.data)
    return h, out


def brute_closest(triangles, origins, dirs, threads=8):
    """Closest accepted hit of every ray by brute force over ALL triangles in face order (no tree): (t, face, ties) per
    ray; t = -1 / face = -1 for a miss, ties = faces sharing the smallest t."""
    from concurrent.futures import ThreadPoolExecutor
    tris = np.ascontiguousarray(triangles)
    o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    n = len(o)
    t = np.zeros(n, np.float32); f = np.zeros(n, np.int32); ties = np.zeros(n, np.int32)
    L = lib()

    def run(k):
        a, b = n * k // threads, n * (k + 1) // threads
        if b > a:
            L.ptor_kat_brute_closest(tris.ctypes.data, len(tris), o[a:b].ctypes.data, d[a:b].ctypes.data, b - a,
                                     t[a:b].ctypes.data, f[a:b].ctypes.data, ties[a:b].ctypes.data)
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(run, range(threads)))
    return t, f, ties
