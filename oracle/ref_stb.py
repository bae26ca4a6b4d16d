"""ctypes view of oracle/_ref/libstb_ref.so = the reference's own vendored third-party/stb_image.h,
compiled where it lies (oracle/Makefile target `ref`).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libstb_ref.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.ptref_stbi_load_from_memory.restype = C.c_long
        L.ptref_stbi_load_from_memory.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                                  C.POINTER(C.c_int), C.c_void_p, C.c_long]
        L.ptref_stbi_failure_reason.restype = C.c_char_p
        _lib = L
    return _lib


def load_from_memory(data: bytes, req_comp: int = 0):
    """stbi_load_from_memory as HostScene.cpp:19-26 calls it -> (w, h, comp, uint8 array) or None."""
    w, h, comp = C.c_int(0), C.c_int(0), C.c_int(0)
    n = lib().ptref_stbi_load_from_memory(data, len(data), req_comp, C.byref(w), C.byref(h), C.byref(comp), None, 1 << 40)
    if n < 0:
        return None
    out = np.zeros(n, np.uint8)
    n2 = lib().ptref_stbi_load_from_memory(data, len(data), req_comp, C.byref(w), C.byref(h), C.byref(comp),
                                           out.ctypes.data_as(C.c_void_p), n)
    assert n2 == n
    return w.value, h.value, comp.value, out
