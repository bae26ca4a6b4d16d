"""Tiny PNG / GLB writers for the loader tests (test data generators, no product code)."""
import json
import struct
import zlib

import numpy as np

SPP = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


def _filter_rows(rows, bpp, filters):
    raw = b""
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = filters[y % len(filters)] if filters else 0
        out = bytearray()
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0: p = 0
            elif ft == 1: p = a
            elif ft == 2: p = b
            elif ft == 3: p = (a + b) // 2
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            out.append((v - p) & 0xFF)
        raw += bytes([ft]) + bytes(out)
        prev = row
    return raw


def pack_samples(samples, bit_depth):
    """samples: (H, W*spp) integer array of sample values -> list of packed row bytes."""
    rows = []
    for r in np.asarray(samples):
        if bit_depth == 8:
            rows.append(bytes(int(v) & 0xFF for v in r))
        elif bit_depth == 16:
            rows.append(b"".join(struct.pack(">H", int(v)) for v in r))
        else:
            per = 8 // bit_depth
            out = bytearray()
            for k in range(0, len(r), per):
                byte = 0
                for j in range(per):
                    v = int(r[k + j]) if k + j < len(r) else 0
                    byte |= (v & ((1 << bit_depth) - 1)) << (8 - bit_depth * (j + 1))
                out.append(byte)
            rows.append(bytes(out))
    return rows


def png_from_samples(width, height, color_type, samples, bit_depth=8, palette=None, trns=None, filters=None,
                     interlace=False, split_idat=True):
    """samples: (H, W*spp) array of sample values (palette indices for colour type 3)."""
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    spp = SPP[color_type]
    bpp = max(1, spp * bit_depth // 8)
    samples = np.asarray(samples).reshape(height, width, spp)
    if not interlace:
        raw = _filter_rows(pack_samples(samples.reshape(height, width * spp), bit_depth), bpp, filters)
    else:
        raw = b""
        for (x0, y0, dx, dy) in [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]:
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] == 0 or sub.shape[1] == 0:
                continue
            raw += _filter_rows(pack_samples(sub.reshape(sub.shape[0], -1), bit_depth), bpp, filters)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, bit_depth, color_type, 0, 0,
                                                               1 if interlace else 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(int(v) for v in np.asarray(palette).reshape(-1)))
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    comp = zlib.compress(raw)
    if split_idat:
        data += chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:])
    else:
        data += chunk(b"IDAT", comp)
    return data + chunk(b"IEND", b"")


def minimal_glb(json_obj, bin_bytes=b""):
    js = json.dumps(json_obj).encode()
    js += b" " * (-len(js) % 4)
    bin_bytes = bin_bytes + b"\0" * (-len(bin_bytes) % 4)
    total = 12 + 8 + len(js) + (8 + len(bin_bytes) if bin_bytes else 0)
    out = struct.pack("<III", 0x46546C67, 2, total) + struct.pack("<II", len(js), 0x4E4F534A) + js
    if bin_bytes:
        out += struct.pack("<II", len(bin_bytes), 0x004E4942) + bin_bytes
    return out


def glb_with_texture(image_bytes, mime="image/png"):
    """One emissive triangle whose base-colour and emissive textures are the embedded image."""
    pos = struct.pack("<9f", 0, 0, 0, 1, 0, 0, 0, 1, 0)
    uv = struct.pack("<6f", 0, 0, 1, 0, 0, 1)
    blob = pos + uv + image_bytes
    j = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "buffers": [{"byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 24},
                        {"buffer": 0, "byteOffset": 60, "byteLength": len(image_bytes)}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC2"}],
        "images": [{"bufferView": 2, "mimeType": mime}], "textures": [{"source": 0}],
        "materials": [{"name": "m", "emissiveFactor": [1, 1, 1], "emissiveTexture": {"index": 0},
                       "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "baseColorFactor": [0.5, 0.25, 1, 1]}}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "material": 0}]}],
    }
    return minimal_glb(j, blob)


def glb_images(path):
    """Embedded image byte strings of a .glb file."""
    blob = open(path, "rb").read()
    jlen = struct.unpack_from("<I", blob, 12)[0]
    j = json.loads(blob[20:20 + jlen])
    bin0 = blob[20 + jlen + 8:]
    out = []
    for im in j.get("images", []):
        bv = j["bufferViews"][im["bufferView"]]
        off = bv.get("byteOffset", 0)
        out.append(bin0[off:off + bv["byteLength"]])
    return out


def png_variants(seed=11):
    """name -> PNG bytes: every colour type / bit depth / filter / interlace combination the tests pin."""
    rng = np.random.default_rng(seed)
    out = {}
    W, H = 7, 5
    pal = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    for interlace in (False, True):
        tag = "_adam7" if interlace else ""
        for depth in (8, 16):
            hi = (1 << depth) - 1
            for ct in (0, 2, 4, 6):
                smp = rng.integers(0, hi + 1, (H, W * SPP[ct]))
                out["ct%d_d%d%s" % (ct, depth, tag)] = png_from_samples(W, H, ct, smp, depth, filters=[0, 1, 2, 3, 4], interlace=interlace)
        for depth in (1, 2, 4):
            smp = rng.integers(0, 1 << depth, (H, W))
            out["ct0_d%d%s" % (depth, tag)] = png_from_samples(W, H, 0, smp, depth, filters=[0, 2, 4], interlace=interlace)
        for depth in (1, 2, 4, 8):
            smp = rng.integers(0, 1 << depth, (H, W))
            out["ct3_d%d%s" % (depth, tag)] = png_from_samples(W, H, 3, smp, depth, palette=pal[:1 << depth], filters=[0, 1, 4],
                                                                interlace=interlace)
            out["ct3_d%d_trns%s" % (depth, tag)] = png_from_samples(W, H, 3, smp, depth, palette=pal[:1 << depth], trns=[0, 128],
                                                                     filters=[3], interlace=interlace)
    # colour-key transparency (tRNS on grey / RGB), short palette, one-pixel and one-row images
    smp = rng.integers(0, 4, (H, W * 3)) * 85
    out["ct2_d8_key"] = png_from_samples(W, H, 2, smp, 8, trns=struct.pack(">HHH", 85, 170, 0))
    smp = rng.integers(0, 4, (H, W))
    out["ct0_d2_key"] = png_from_samples(W, H, 0, smp, 2, trns=struct.pack(">H", 2))
    smp16 = rng.integers(0, 3, (H, W * 3)) * 30000
    out["ct2_d16_key"] = png_from_samples(W, H, 2, smp16, 16, trns=struct.pack(">HHH", 30000, 0, 60000))
    out["ct3_short_palette"] = png_from_samples(W, H, 3, rng.integers(0, 8, (H, W)), 4, palette=pal[:5])
    out["ct2_1x1"] = png_from_samples(1, 1, 2, [[9, 8, 7]], 8)
    out["ct6_1x9_adam7"] = png_from_samples(1, 9, 6, rng.integers(0, 256, (9, 4)), 8, interlace=True, filters=[4])
    out["ct2_33x3"] = png_from_samples(33, 3, 2, rng.integers(0, 256, (3, 99)), 8, filters=[1, 3, 4], split_idat=False)
    return out


def png_malformed(seed=12):
    """name -> bytes that stb_image refuses (the reference then throws, HostScene.cpp:33-35)."""
    rng = np.random.default_rng(seed)
    W, H = 4, 3
    rgb = rng.integers(0, 256, (H, W * 3))
    rgba = rng.integers(0, 256, (H, W * 4))
    good = png_from_samples(W, H, 2, rgb, 8, split_idat=False)
    out = {}
    out["trns_on_rgba"] = png_from_samples(W, H, 6, rgba, 8, trns=struct.pack(">HHH", 1, 2, 3))
    out["trns_bad_length"] = png_from_samples(W, H, 2, rgb, 8, trns=b"\0\1\0\2")
    out["palette_without_plte"] = png_from_samples(W, H, 3, rng.integers(0, 4, (H, W)), 2)
    out["bad_filter_type"] = png_from_samples(W, H, 2, rgb, 8, filters=[7])
    out["bad_colour_type"] = good[:25] + b"\x05" + good[26:]
    out["truncated_idat"] = good[:len(good) - 30]
    out["not_a_png"] = b"\xff\xd8\xff\xe0" + bytes(64)
    out["bit_depth_3"] = good[:24] + b"\x03" + good[25:]
    return out


def glb_to_gltf(glb_path, out_dir, images="file", buffer="file"):
    """Rewrites a .glb as .gltf JSON text: the BIN chunk becomes scene.bin (or a base64 data: URI), every image
    a file next to it (images="file"), a data: URI ("data"), or stays a bufferView ("view")."""
    import base64
    import os
    blob = open(glb_path, "rb").read()
    jlen = struct.unpack_from("<I", blob, 12)[0]
    j = json.loads(blob[20:20 + jlen])
    blen = struct.unpack_from("<I", blob, 20 + jlen)[0]
    bin0 = blob[20 + jlen + 8:20 + jlen + 8 + blen]
    if buffer == "file":
        open(os.path.join(out_dir, "scene data.bin"), "wb").write(bin0)
        j["buffers"][0]["uri"] = "scene%20data.bin"                      # percent-encoded space
    else:
        j["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bin0).decode()
    for k, im in enumerate(j.get("images", [])):
        if images == "view":
            continue
        bv = j["bufferViews"][im.pop("bufferView")]
        data = bin0[bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]]
        ext = "png" if data[:4] == b"\x89PNG" else "jpg"
        if images == "file":
            open(os.path.join(out_dir, "img%d.%s" % (k, ext)), "wb").write(data)
            im["uri"] = "img%d.%s" % (k, ext)
        else:
            im["uri"] = "data:image/%s;base64,%s" % ("png" if ext == "png" else "jpeg", base64.b64encode(data).decode())
    path = os.path.join(out_dir, "scene.gltf")
    open(path, "w").write("﻿" + json.dumps(j, indent=1))          # with a BOM, as some exporters write
    return path
