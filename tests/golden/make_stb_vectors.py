#!/usr/bin/env python3
"""Regenerates tests/golden/stb_png_vectors.npz and stb_jpeg_vectors.npz with the REFERENCE's own texture decoder:
third-party/stb_image.h of /root/reference, compiled where it lies into oracle/_ref/libstb_ref.so
(`make -C oracle ref`).  Inputs are synthetic PNGs / JPEGs from tests/pngkit.py and tests/jpegkit.py plus the texture embedded in
the reference's models/cornell_duck.glb; outputs are what stbi_load_from_memory(req_comp = 0) returns
for them, i.e. exactly what SceneLoader::loadTextureFromFile (HostScene.cpp:19-26) starts from.
Data only -- no reference source text is stored.  Run from the repo root (needs /root/reference):
    make -C oracle ref && python tests/golden/make_stb_vectors.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegkit  # noqa: E402
import pngkit  # noqa: E402
from oracle import ref_stb  # noqa: E402

UNDEFINED_IN_STB = {"ct3_short_palette"}      # palette index past PLTE: stb reads an uninitialised array


def main():
    assert ref_stb.available(), "build oracle/_ref first: make -C oracle ref"
    out = {}
    names = []
    variants = dict(pngkit.png_variants())
    variants.update({"bad_" + k: v for k, v in pngkit.png_malformed().items()})
    for name, png in variants.items():
        if name in UNDEFINED_IN_STB:
            continue
        names.append(name)
        out["png_" + name] = np.frombuffer(png, np.uint8)
        r = ref_stb.load_from_memory(png)
        if r is None:
            out["meta_" + name] = np.array([-1, -1, -1], np.int32)
        else:
            w, h, comp, px = r
            out["meta_" + name] = np.array([w, h, comp], np.int32)
            out["px_" + name] = px
    duck = pngkit.glb_images(os.path.join(ROOT, "models", "cornell_duck.glb"))[0]
    w, h, comp, px = ref_stb.load_from_memory(duck)
    out["duck_meta"] = np.array([w, h, comp], np.int32)
    out["duck_sha256"] = np.frombuffer(hashlib.sha256(px.tobytes()).digest(), np.uint8)
    out["names"] = np.array(names)
    path = os.path.join(ROOT, "tests", "golden", "stb_png_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(names), "vectors; duck texture", (w, h, comp), os.path.getsize(path), "bytes")
    # JPEG
    out, names = {}, []
    variants = dict(jpegkit.jpeg_variants())
    variants.update({"bad_" + k: v for k, v in jpegkit.jpeg_malformed().items()})
    for name, jpg in variants.items():
        names.append(name)
        out["jpg_" + name] = np.frombuffer(jpg, np.uint8)
        r = ref_stb.load_from_memory(jpg)
        if r is None:
            out["meta_" + name] = np.array([-1, -1, -1], np.int32)
        else:
            out["meta_" + name] = np.array(r[:3], np.int32)
            out["px_" + name] = r[3]
    out["names"] = np.array(names)
    path = os.path.join(ROOT, "tests", "golden", "stb_jpeg_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(names), "vectors", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
