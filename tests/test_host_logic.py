"""CPU tests of the product's host side (no GPU compute): scene I/O, BVH build, task
generators, band packing, error behaviour, and that the C-ABI library exports every symbol
include/pt_amd.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden")


def test_capi_exports_every_declared_symbol(pt):
    hdr = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    names = re.findall(r"PT_API\s+[\w\s\*]+?\b(pt_\w+)\s*\(", hdr)
    assert len(names) >= 40
    handle = C.CDLL(pt.capi.LIB_PATH)
    missing = [n for n in names if not hasattr(handle, n)]
    assert not missing, missing
    # and the Python binding covers the same set
    assert sorted(set(names)) == sorted(pt.capi._PROTOS.keys())
    assert pt.capi.lib().pt_version().decode().startswith("ptamd")


def test_capi_signatures_carry_no_torch_or_hip_types():
    hdr = open(os.path.join(ROOT, "include", "pt_amd.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for banned in ("hipStream_t", "torch", "at::", "std::", "float3", "dim3", "#include <hip"):
        assert banned not in code, banned


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "multi-gpu-path-tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".c", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                for line in text.splitlines():
                    if re.search(r'(#include|import|from|dlopen|CDLL).*\boracle\b', line):
                        pytest.fail("product file %s references the oracle: %s" % (f, line))
                    assert "libpt_oracle" not in line and "ptor_" not in line, (f, line)


@pytest.mark.parametrize("name", ["box", "duck"])
def test_glb_loader_reproduces_hostscene_dump(pt, box_scene, duck_scene, name):
    sc = box_scene if name == "box" else duck_scene
    g = np.load(os.path.join(GOLD, "hostscene_%s.npz" % name))
    assert sc.triangles.tobytes() == g["triangles"].tobytes()
    assert sc.materials.tobytes() == g["materials"].tobytes()


def test_glb_loader_semantics(pt, box_scene, duck_scene):
    # triangles regrouped by material index ascending
    for sc in (box_scene, duck_scene):
        m = sc.triangles["materialIdx"]
        assert np.all(np.diff(m) >= 0)
    # box: mat0 6, mat1 2, mat2 2, mat3 24, mat4 2 triangles
    assert np.bincount(box_scene.triangles["materialIdx"]).tolist() == [6, 2, 2, 24, 2]
    assert len(duck_scene.triangles) == 4224 and np.bincount(duck_scene.triangles["materialIdx"])[4] == 4212
    # benchmark convention: the diffuse_light material became emissive (1,1,1); up = +Y
    assert box_scene.material_names[4] == "diffuse_light"
    assert box_scene.materials["emissiveFactor"][4].tolist() == [1, 1, 1]
    pos = np.concatenate([box_scene.triangles[v]["position"] for v in ("v0", "v1", "v2")])
    assert pos[:, 1].min() > -1 and pos[:, 1].max() > 500            # floor at y ~ 0, ceiling ~ 505
    # duck: emissive (1,1,1) from the file, strength extension ignored; 512x512 palette PNG texture
    assert duck_scene.materials["emissiveFactor"][2].tolist() == [1, 1, 1]
    (tex,) = duck_scene.textures
    assert tex.shape == (512, 512, 3) and tex.min() >= 0 and tex.max() <= 255 and len(np.unique(tex.reshape(-1, 3), axis=0)) <= 54
    uv = duck_scene.triangles["v0"]["texCoords"][duck_scene.triangles["materialIdx"] == 4]
    assert uv.min() >= 0 and uv.max() <= 1 and uv.std() > 0.05
    assert np.all(duck_scene.triangles["v0"]["texCoords"][:12] == 0)  # meshes without TEXCOORD_0 (HostScene.cpp:214)


def test_loader_error_behaviour(pt, tmp_path):
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load(str(tmp_path / "missing.glb"))
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load("model.fbx")                            # HostScene.cpp:134 unsupported type
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"glTF" + b"\x02\x00\x00\x00" + b"\x20\x00\x00\x00" + b"\x00" * 20)
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load(str(bad))


@pytest.mark.parametrize("name", ["box", "duck"])
def test_host_bvh_equals_oracle_and_golden(pt, box_scene, duck_scene, box_oracle, duck_oracle, name):
    sc, osc = (box_scene, box_oracle) if name == "box" else (duck_scene, duck_oracle)
    info, bounds, links, idx = pt.host_bvh_build(sc.triangles)
    ob_b, ob_l, ob_i, _ = osc.bvh()
    g = np.load(os.path.join(GOLD, "bvh_%s.npz" % name))
    for b, l, i in ((ob_b, ob_l, ob_i), (g["bounds"], g["links"], g["indices"])):
        assert np.array_equal(bounds.view(np.uint32), b.view(np.uint32))
        assert np.array_equal(links, l) and np.array_equal(idx, i)
    expect = {"box": (21, 11, 5, 6, 3), "duck": (665, 333, 1603, 18, 8)}[name]
    assert (info["nodes"], info["leaves"], info["max_leaf"], info["depth"], info["max_stack"]) == expect


def test_bvh_root_leaf_and_tiny_scenes(pt, ob):
    """<= 5 triangles: the root is never split and is its own two children (bvh.h:26,:189-190)."""
    tris = np.zeros(3, pt.capi.TRIANGLE_DTYPE)
    for k in range(3):
        tris[k]["v0"]["position"] = (k, 0, -5); tris[k]["v1"]["position"] = (k + 1, 0, -5); tris[k]["v2"]["position"] = (k, 1, -5)
    info, bounds, links, idx = pt.host_bvh_build(tris)
    assert info["nodes"] == 1 and links[0].tolist() == [0, 0, 0, 3] and info["max_stack"] == 1
    mats = np.zeros(1, pt.capi.MATERIAL_DTYPE); mats["emissiveFactor"] = 1
    mats["baseColorTextureIdx"] = -1; mats["emissiveTextureIdx"] = -1
    osc = ob.OracleScene(tris, mats)
    assert np.array_equal(osc.bvh()[1], links)
    _, _, c = osc.render(((0.5, 0.3, 0), (0, 0, -1), 45, 45), 8, 8, 1, 2)
    assert c["box_tests"] == 2 * c["segments"]                       # root tested as both of its children


def test_threaded_reference_build_equals_the_single_threaded_oracle_build(pt, ob):
    """The host build forks subtrees of >= 2^16 triangles and costs the split candidates of nodes of >= 2^19
    triangles concurrently; topology, node numbers and the triangle permutation must be those of the serial
    build (bvh.h:35-94), which the oracle restates with one thread.  524 290 triangles reach both thresholds."""
    sc = pt.make_terrain_scene(512)
    assert len(sc.triangles) >= 1 << 19
    info, bounds, links, idx = pt.host_bvh_build(sc.triangles)
    osc = ob.OracleScene(sc.triangles, sc.materials)
    o_bounds, o_links, o_idx, _ = osc.bvh()
    assert np.array_equal(links, o_links) and np.array_equal(idx, o_idx)
    assert np.array_equal(bounds.view(np.uint32), o_bounds.view(np.uint32))
    info2, bounds2, links2, idx2 = pt.host_bvh_build(sc.triangles)              # and it is deterministic
    assert info2 == info and np.array_equal(links2, links) and np.array_equal(idx2, idx)


@pytest.mark.parametrize("specials", [(np.nan, 0.0, -0.0, 1e-40, -1e-40), (np.nan, np.inf, -np.inf, 3e38, -3e38, 0.0, -0.0)])
def test_reference_build_handles_nan_zero_and_infinite_coordinates_like_the_oracle(pt, ob, specials):
    """The host build replaces aabb::extend's fminf / fmaxf by single min / max instructions and evaluates the split
    candidates from pattern boxes; both rest on 'NaN coordinates are ignored' and 'no bound is a zero'.  Check them
    on a soup with NaN, signed zeros, denormals (deep tree) and with infinities / huge values (degenerate tree)."""
    rng = np.random.default_rng(5)
    n = 3000
    tris = np.zeros(n, pt.capi.TRIANGLE_DTYPE)
    for k in ("v0", "v1", "v2"):
        tris[k]["position"] = rng.normal(0, 50, (n, 3)).astype(np.float32)
    for _ in range(600):
        k = ("v0", "v1", "v2")[int(rng.integers(3))]
        tris[k]["position"][int(rng.integers(n)), int(rng.integers(3))] = specials[int(rng.integers(len(specials)))]
    tris["v0"]["position"][:200, 0] = 0.0; tris["v1"]["position"][:200, 0] = -0.0
    mats = np.zeros(1, pt.capi.MATERIAL_DTYPE); mats["emissiveFactor"] = 1
    mats["baseColorTextureIdx"] = -1; mats["emissiveTextureIdx"] = -1
    info, bounds, links, idx = pt.host_bvh_build(tris)
    o_bounds, o_links, o_idx, _ = ob.OracleScene(tris, mats).bvh()
    assert np.array_equal(links, o_links) and np.array_equal(idx, o_idx)
    assert np.array_equal(bounds.view(np.uint32), o_bounds.view(np.uint32))
    assert info["nodes"] > (100 if not np.isinf(specials).any() else 1)


def test_fsfl_task_layout_matches_reference_rule(pt):
    """TaskGenerator::generateEqualTasks (TaskGenerator.h:58-80) over getTaskLayout
    (RenderManager.h:42-59): rows of <= maxTasksInRow, last column/row absorb remainders."""
    tg = pt.TaskGenerator()
    for count, maxrow, W, H in [(1, 2, 400, 400), (2, 2, 1920, 1080), (3, 2, 401, 399), (8, 2, 1920, 1080), (5, 3, 1001, 777)]:
        layout = pt.get_task_layout(count, maxrow)
        tasks = tg.generateEqualTasks(count, layout, W, H)
        cover = np.zeros((H, W), np.int32)
        th = H // len(layout)
        for i, row in enumerate(layout):
            tw = W // len(row)
            for j, t in enumerate(row):
                assert tasks[t].offset_x == tw * j and tasks[t].offset_y == th * i
                assert tasks[t].width == (tw if j < len(row) - 1 else W - tw * j)
                assert tasks[t].height == (th if i < len(layout) - 1 else H - th * i)
                cover[tasks[t].offset_y:tasks[t].offset_y + tasks[t].height, tasks[t].offset_x:tasks[t].offset_x + tasks[t].width] += 1
        assert np.all(cover == 1)
    row = tg.generateEqualTasks(3, 1000, 50)                         # TaskGenerator.h:46-55
    assert [(t.width, t.height, t.offset_x, t.offset_y) for t in row] == [(333, 50, 0, 0), (333, 50, 333, 0), (334, 50, 666, 0)]


def test_row_bands_cover_the_frame_exactly_once(pt):
    for W, H, band, n in [(1920, 1080, 8, 8), (64, 36, 8, 3), (400, 400, 16, 5), (32, 6, 8, 4), (1920, 1080, 8, 1)]:
        cover = np.zeros(H, np.int32)
        for r in range(n):
            tasks = pt.row_band_tasks(W, H, band, n, r)
            ranges = pt.band_row_ranges(H, band, n, r)
            assert len(tasks) == len(ranges)
            for t, (r0, r1) in zip(tasks, ranges):
                assert t.width == W and t.offset_x == 0 and t.height == r1 - r0 and t.offset_y == H - r1
                assert r0 % 2 == 0                                    # chroma rows stay whole inside a band
                cover[r0:r1] += 1
        assert np.all(cover == 1)
    with pytest.raises(pt.capi.PtError):
        pt.row_band_tasks(64, 64, 7, 2, 0)                           # odd band height splits chroma rows


def test_pack_unpack_bands_roundtrip(pt):
    rng = np.random.default_rng(0)
    W, H, band, n = 64, 36, 8, 3
    total = W * H
    rgb = rng.integers(0, 256, total * 3, dtype=np.uint8)
    yuv = rng.integers(0, 256, total + 2 * (total // 4), dtype=np.uint8)
    out_rgb = np.zeros_like(rgb); out_yuv = np.zeros_like(yuv)
    for r in range(n):
        ranges = pt.band_row_ranges(H, band, n, r)
        packed = pt.pack_bands(rgb, yuv, W, H, ranges, np)
        assert packed.size == pt.packed_size(W, H, ranges)
        assert pt.unpack_bands(packed, out_rgb, out_yuv, W, H, ranges) == packed.size
    assert np.array_equal(out_rgb, rgb) and np.array_equal(out_yuv, yuv)


def test_strided_pack_equals_generic_pack(pt):
    import torch
    rng = np.random.default_rng(1)
    for W, H, band, n in [(64, 40, 8, 3), (32, 16, 2, 4), (1920, 1080, 8, 8)]:
        total = W * H
        rgb = torch.from_numpy(rng.integers(0, 256, total * 3, dtype=np.uint8))
        yuv = torch.from_numpy(rng.integers(0, 256, total + 2 * (total // 4), dtype=np.uint8))
        out_rgb = torch.zeros_like(rgb); out_yuv = torch.zeros_like(yuv)
        for r in range(n):
            ranges = pt.band_row_ranges(H, band, n, r)
            a = pt.pack_bands(rgb, yuv, W, H, ranges, torch)
            b = pt.pack_bands_strided(rgb, yuv, W, H, band, n, r)
            assert torch.equal(a, b)
            assert pt.unpack_bands_strided(b, out_rgb, out_yuv, W, H, band, n, r) == b.numel()
        assert torch.equal(out_rgb, rgb) and torch.equal(out_yuv, yuv)


def test_ppm_obj_and_terrain(pt, tmp_path):
    img = np.arange(4 * 3 * 3, dtype=np.uint8).reshape(3, 4, 3)
    p = tmp_path / "o.ppm"
    pt.write_ppm(str(p), img)
    assert p.read_bytes() == b"P6\n4 3\n255\n" + img.tobytes()
    (tmp_path / "m.mtl").write_text("newmtl lambertian_grey\nKa 0.3 0.4 0.5\nKd 1 1 1\nnewmtl diffuse_light_top\nKd 2 2 2\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\n"
                                    "usemtl lambertian_grey\nf 1/1 2/2 3/3 4/4\nusemtl diffuse_light_top\nf -4 -3 -2\n")
    sc = pt.SceneLoader().load(str(tmp_path / "m.obj"))
    assert len(sc.triangles) == 3 and sc.material_names == ["lambertian_grey", "diffuse_light_top"]
    assert np.allclose(sc.materials["baseColor"][0], [0.3, 0.4, 0.5]) and sc.materials["emissiveFactor"][1].tolist() == [2, 2, 2]
    assert sc.triangles["v2"]["texCoords"][1].tolist() == [0, 1]
    # extension: the legacy class each name would have got from obj_loader.h:65-96
    (tmp_path / "n.mtl").write_text("newmtl lambertian_a\nKa 0.3 0.4 0.5\nnewmtl metal_b\nKa 0.9 0.8 0.7\nNs 0.25\n"
                                    "newmtl dielectric_c\nNi 1.5\nnewmtl diffuse_light_d\nKd 4 5 6\nnewmtl other\nKd 0.1 0.2 0.3\n")
    (tmp_path / "n.obj").write_text("mtllib n.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\n" +
                                    "".join("usemtl %s\nf 1 2 3\n" % m for m in ("lambertian_a", "metal_b", "dielectric_c", "diffuse_light_d", "other")))
    lm = pt.SceneLoader().load(str(tmp_path / "n.obj")).legacyMaterials()
    assert lm["kind"].tolist() == [1, 2, 3, 4, 0]
    assert np.allclose(lm["albedo"][0], [0.3, 0.4, 0.5]) and np.allclose(lm["albedo"][1], [0.9, 0.8, 0.7]) and np.allclose(lm["albedo"][3], [4, 5, 6])
    assert np.isclose(lm["param"][1], 0.25) and np.isclose(lm["param"][2], 1.5)
    assert sc.legacyMaterials()["kind"].tolist() == [1, 4]
    terr = pt.make_terrain_scene(16)
    assert len(terr.triangles) == 2 * 16 * 16 + 2 and np.allclose(terr.materials["emissiveFactor"][1], 0.25)
    assert not terr.legacyMaterials()["kind"].any()
    t2 = pt.make_terrain_scene(16)
    assert terr.triangles.tobytes() == t2.triangles.tobytes()        # deterministic
    info, _, links, _ = pt.host_bvh_build(terr.triangles)
    assert info["max_stack"] <= 64 and links[:, 3].sum() == len(terr.triangles)


def test_render_without_gpu_fails_loudly(pt, box_scene):
    """No CPU fallback: on a box without a GPU creating the engine is an error, not a slow path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    st = pt.capi.lib().pt_device_create(0, C.byref(h))
    assert st == 8 and b"no CPU fallback" in pt.capi.lib().pt_last_error_string()


def _minimal_glb(json_obj, bin_bytes=b""):
    import json as _json
    import struct
    js = _json.dumps(json_obj).encode()
    js += b" " * (-len(js) % 4)
    bin_bytes = bin_bytes + b"\x00" * (-len(bin_bytes) % 4)
    chunks = struct.pack("<II", len(js), 0x4E4F534A) + js
    if bin_bytes:
        chunks += struct.pack("<II", len(bin_bytes), 0x004E4942) + bin_bytes
    return b"glTF" + struct.pack("<II", 2, 12 + len(chunks)) + chunks


def test_glb_loader_handcrafted_files(pt, tmp_path):
    """Node hierarchy with nested TRS, matrix nodes, non-indexed primitives, default material,
    negative-determinant transforms (winding flip) and malformed input (errors, not crashes)."""
    import struct
    pos = struct.pack("<9f", 0, 0, 0, 1, 0, 0, 0, 1, 0)
    base = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}],
        "buffers": [{"byteLength": len(pos)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(pos)}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}],
    }
    # child node under a translated + scaled parent; non-indexed triangle; no material -> DefaultMaterial
    j = dict(base, nodes=[{"translation": [10, 0, 0], "scale": [2, 2, 2], "children": [1]}, {"mesh": 0, "translation": [0, 1, 0]}])
    p = tmp_path / "a.glb"; p.write_bytes(_minimal_glb(j, pos))
    sc = pt.SceneLoader().load(str(p))
    t = sc.triangles[0]
    assert t["v0"]["position"].tolist() == [10, 2, 0] and t["v1"]["position"].tolist() == [12, 2, 0] and t["v2"]["position"].tolist() == [10, 4, 0]
    assert sc.material_names == ["DefaultMaterial"] and sc.materials["baseColor"][0].tolist() == [1, 1, 1]
    # mirrored node (negative determinant): winding is reversed so the geometric normal stays consistent
    j = dict(base, nodes=[{"mesh": 0, "scale": [-1, 1, 1]}])
    p.write_bytes(_minimal_glb(j, pos))
    t = pt.SceneLoader().load(str(p)).triangles[0]
    assert t["v0"]["position"].tolist() == [0, 1, 0] and t["v2"]["position"].tolist() == [0, 0, 0]
    # column-major matrix node: translate by (5, 6, 7)
    j = dict(base, nodes=[{"mesh": 0, "matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 5, 6, 7, 1]}])
    p.write_bytes(_minimal_glb(j, pos))
    assert pt.SceneLoader().load(str(p)).triangles[0]["v0"]["position"].tolist() == [5, 6, 7]
    # degenerate triangle (two identical vertices) is dropped -> no triangles -> error
    deg = struct.pack("<9f", 0, 0, 0, 0, 0, 0, 0, 1, 0)
    p.write_bytes(_minimal_glb(dict(base, nodes=[{"mesh": 0}]), deg))
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load(str(p))
    # triangle strip / fan of 5 vertices: 3 triangles each, strip triangle 1 re-ordered to keep the orientation
    five = struct.pack("<15f", 0, 0, 0, 1, 0, 0, 0, 1, 0, 1, 1, 0, 0, 2, 0)
    v = [[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 2, 0]]
    sf = dict(base, nodes=[{"mesh": 0}], buffers=[{"byteLength": len(five)}],
              bufferViews=[{"buffer": 0, "byteOffset": 0, "byteLength": len(five)}],
              accessors=[{"bufferView": 0, "componentType": 5126, "count": 5, "type": "VEC3"}])
    for mode, want in ((5, [(0, 1, 2), (2, 1, 3), (2, 3, 4)]), (6, [(0, 1, 2), (0, 2, 3), (0, 3, 4)])):
        sf["meshes"] = [{"primitives": [{"attributes": {"POSITION": 0}, "mode": mode}]}]
        p.write_bytes(_minimal_glb(sf, five))
        tr = pt.SceneLoader().load(str(p)).triangles
        got = [tuple(v.index(t[k]["position"].tolist()) for k in ("v0", "v1", "v2")) for t in tr]
        assert got == want, (mode, got)
    # sparse accessor: zero-initialised positions (no bufferView) with two vertices supplied, and a base view
    # with one vertex replaced
    spi = struct.pack("<2H", 1, 2)
    spv = struct.pack("<6f", 3, 0, 0, 0, 4, 0)
    blob = pos + spi + spv
    sp = dict(base, nodes=[{"mesh": 0}], buffers=[{"byteLength": len(blob)}],
              bufferViews=[{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 4},
                           {"buffer": 0, "byteOffset": 40, "byteLength": 24}])
    sparse = {"count": 2, "indices": {"bufferView": 1, "componentType": 5123}, "values": {"bufferView": 2}}
    sp["accessors"] = [{"componentType": 5126, "count": 3, "type": "VEC3", "sparse": sparse}]
    p.write_bytes(_minimal_glb(sp, blob))
    t = pt.SceneLoader().load(str(p)).triangles[0]
    assert [t[k]["position"].tolist() for k in ("v0", "v1", "v2")] == [[0, 0, 0], [3, 0, 0], [0, 4, 0]]
    sp["accessors"] = [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3",
                        "sparse": {"count": 1, "indices": {"bufferView": 1, "componentType": 5123}, "values": {"bufferView": 2}}}]
    p.write_bytes(_minimal_glb(sp, blob))
    t = pt.SceneLoader().load(str(p)).triangles[0]
    assert [t[k]["position"].tolist() for k in ("v0", "v1", "v2")] == [[0, 0, 0], [3, 0, 0], [0, 1, 0]]
    sp["accessors"][0]["sparse"]["indices"]["byteOffset"] = 2                    # index 2, then the first value
    p.write_bytes(_minimal_glb(sp, blob))
    assert pt.SceneLoader().load(str(p)).triangles[0]["v2"]["position"].tolist() == [3, 0, 0]
    sp["accessors"][0]["sparse"]["count"] = 9                                    # beyond the index view
    p.write_bytes(_minimal_glb(sp, blob))
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load(str(p))
    # points and lines are dropped (SortByPType), leaving nothing to render
    sf["meshes"] = [{"primitives": [{"attributes": {"POSITION": 0}, "mode": 1}]}]
    p.write_bytes(_minimal_glb(sf, five))
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load(str(p))
    # malformed: accessor beyond the buffer, index out of range, truncated chunk, garbage JSON
    bad = dict(base, nodes=[{"mesh": 0}]); bad["accessors"] = [{"bufferView": 0, "componentType": 5126, "count": 300, "type": "VEC3"}]
    p.write_bytes(_minimal_glb(bad, pos))
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load(str(p))
    idx = struct.pack("<3H", 0, 1, 7) + b"\x00\x00"
    bad = dict(base, nodes=[{"mesh": 0}])
    bad["buffers"] = [{"byteLength": len(pos) + len(idx)}]
    bad["bufferViews"] = base["bufferViews"] + [{"buffer": 0, "byteOffset": len(pos), "byteLength": 6}]
    bad["accessors"] = base["accessors"] + [{"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}]
    bad["meshes"] = [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}]
    p.write_bytes(_minimal_glb(bad, pos + idx))
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load(str(p))
    # sizes whose products wrap 64 bits, or that are not sizes at all (found by tools/loader_fuzz/fuzz_gltf.py: count x stride
    # wrapped to a small number and the accessor read far outside its buffer)
    for acc, view in (({"count": 2 ** 47}, {"byteStride": 2 ** 20}), ({"count": 2 ** 44 + 1}, {"byteStride": 2 ** 20}),
                      ({"count": 1e308}, {}), ({"count": -3}, {}), ({"byteOffset": 2 ** 64}, {}), ({"byteOffset": -12}, {}),
                      ({}, {"byteOffset": 2 ** 63}), ({}, {"byteLength": 2 ** 64 + 36}), ({"count": "3"}, {})):
        bad = dict(base, nodes=[{"mesh": 0}])
        bad["accessors"] = [dict(base["accessors"][0], **acc)]
        bad["bufferViews"] = [dict(base["bufferViews"][0], **view)]
        p.write_bytes(_minimal_glb(bad, pos))
        with pytest.raises(RuntimeError):
            pt.SceneLoader().load(str(p))
    # the node hierarchy must be a strict tree: cycles and shared children (which instantiate exponentially) are refused
    for nodes in ([{"mesh": 0, "children": [0]}], [{"mesh": 0, "children": [1]}, {"children": [0]}],
                  [{"mesh": 0, "children": [1] * 40}] + [{"mesh": 0, "children": [k + 2] * 40} for k in range(6)] + [{"mesh": 0}]):
        p.write_bytes(_minimal_glb(dict(base, nodes=nodes), pos))
        with pytest.raises(RuntimeError):
            pt.SceneLoader().load(str(p))
    whole = _minimal_glb(dict(base, nodes=[{"mesh": 0}]), pos)
    for cut in (10, 20, len(whole) - 5):
        p.write_bytes(whole[:cut])
        with pytest.raises(RuntimeError):
            pt.SceneLoader().load(str(p))
    p.write_bytes(b"glTF" + struct.pack("<II", 2, 40) + struct.pack("<II", 20, 0x4E4F534A) + b"{not json at all!!}}")
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load(str(p))


def test_capi_argument_checking(pt, box_scene):
    """Null pointers and bad sizes come back as PT_ERR_INVALID with a message, never a crash."""
    lib = pt.capi.lib()
    out = C.c_void_p()
    assert lib.pt_scene_load_glb(None, C.byref(out)) == 1 and lib.pt_last_error_string()
    assert lib.pt_scene_make_terrain(0, C.byref(out)) == 1
    assert lib.pt_tasks_equal_row(0, 10, 10, None) == 1
    assert lib.pt_tasks_equal_layout(2, 0, 10, 10, None) == 1
    n = C.c_int()
    assert lib.pt_tasks_row_bands(10, 10, 3, 1, 0, None, 0, C.byref(n)) == 1          # odd band height
    assert lib.pt_host_bvh_build(None, 0, None, None, None, None) == 1
    assert lib.pt_write_ppm(b"/nonexistent-dir/x.ppm", 2, 2, pt.capi.np_ptr(np.zeros(12, np.uint8))) == 5
    assert b"cannot write" in lib.pt_last_error_string()
    # no C++ exception crosses the C ABI: a triangle count whose working arrays exceed what a vector can hold makes
    # std::vector throw before anything is read or allocated; it comes back as PT_ERR_NOMEM (9) with a message
    one = np.zeros(1, pt.capi.TRIANGLE_DTYPE)
    assert lib.pt_host_bvh_build(pt.capi.np_ptr(one), 1 << 60, None, None, None, None) == 9
    assert b"allocation failed" in lib.pt_last_error_string()
    assert lib.pt_host_sah_build(pt.capi.np_ptr(one), 1 << 60, None, None, None) == 9
    assert pt.capi.STATUS_NAMES[9] == "PT_ERR_NOMEM" and pt.capi.STATUS_NAMES[10] == "PT_ERR_INTERNAL"
    # device entry points with a null device never touch the GPU
    assert lib.pt_device_sync(None) == 1 and lib.pt_counters_reset(None) == 1
    assert lib.pt_render_task_async(None, None, None, 1, 1, None) == 1
    assert lib.pt_device_destroy(None) == 0


def _png(width, height, color_type, rows, bit_depth=8, palette=None, trns=None, filters=None):
    """Tiny PNG writer for tests: rows = list of per-row raw sample bytes; filters = per-row filter
    type (the row bytes are filtered here with the requested PNG filter)."""
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    spp = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    bpp = max(1, spp * bit_depth // 8)
    raw = b""
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = filters[y] if filters else 0
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
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, bit_depth, color_type, 0, 0, 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    half = len(raw) // 2 or 1
    comp = zlib.compress(raw)
    data += chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:])     # split IDAT
    return data + chunk(b"IEND", b"")


def _glb_with_texture(png_bytes):
    import struct
    pos = struct.pack("<9f", 0, 0, 0, 1, 0, 0, 0, 1, 0)
    uv = struct.pack("<6f", 0, 0, 1, 0, 0, 1)
    blob = pos + uv + png_bytes
    j = {
        "asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "buffers": [{"byteLength": len(blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 24},
                        {"buffer": 0, "byteOffset": 60, "byteLength": len(png_bytes)}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5126, "count": 3, "type": "VEC2"}],
        "images": [{"bufferView": 2, "mimeType": "image/png"}], "textures": [{"source": 0}],
        "materials": [{"name": "m", "emissiveFactor": [1, 1, 1], "emissiveTexture": {"index": 0},
                       "pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "baseColorFactor": [0.5, 0.25, 1, 1]}}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "material": 0}]}],
    }
    return _minimal_glb(j, blob)


def test_embedded_png_variants(pt, tmp_path):
    """Embedded textures as stb_image (req_comp = 0) + the reference's 3-bytes-per-pixel copy loop
    (HostScene.cpp:19-46) would deliver them: RGB, palette (+ every PNG filter, split IDAT), 16-bit,
    RGBA read with the reference's wrong stride, grey refused (the reference reads out of bounds)."""
    rng = np.random.default_rng(3)
    p = tmp_path / "t.glb"
    W, H = 5, 4
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    # 8-bit RGB, all five filter types
    p.write_bytes(_glb_with_texture(_png(W, H, 2, [bytes(r.reshape(-1)) for r in rgb], filters=[0, 1, 2, 3][:H - 1] + [4])))
    sc = pt.SceneLoader().load(str(p))
    (tex,) = sc.textures
    assert tex.shape == (H, W, 3) and np.array_equal(tex.astype(np.uint8), rgb)
    assert sc.materials["baseColorTextureIdx"][0] == 0 and sc.materials["emissiveTextureIdx"][0] == 0      # one image, deduplicated
    assert sc.triangles["v2"]["texCoords"][0].tolist() == [0, 0]                                          # v flipped: 1 - 1
    # palette, 4 bits per index
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    idx = rng.integers(0, 16, (H, 6), dtype=np.uint8)
    rows = [bytes((int(r[2 * k]) << 4) | int(r[2 * k + 1]) for k in range(3)) for r in idx]
    p.write_bytes(_glb_with_texture(_png(6, H, 3, rows, bit_depth=4, palette=pal.reshape(-1))))
    (tex,) = pt.SceneLoader().load(str(p)).textures
    assert np.array_equal(tex.astype(np.uint8), pal[idx])
    # 16-bit RGB: high byte is kept
    hi = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    rows16 = [bytes(b for v in r.reshape(-1) for b in (int(v), 7)) for r in hi]
    p.write_bytes(_glb_with_texture(_png(W, H, 2, rows16, bit_depth=16)))
    (tex,) = pt.SceneLoader().load(str(p)).textures
    assert np.array_equal(tex.astype(np.uint8), hi)
    # RGBA: the reference copies width*height*3 bytes out of a 4-byte-per-pixel buffer (mis-strided)
    rgba = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    p.write_bytes(_glb_with_texture(_png(W, H, 6, [bytes(r.reshape(-1)) for r in rgba])))
    (tex,) = pt.SceneLoader().load(str(p)).textures
    assert np.array_equal(tex.astype(np.uint8).reshape(-1), rgba.reshape(-1)[:W * H * 3])
    # palette + tRNS -> 4 channels -> same mis-stride
    p.write_bytes(_glb_with_texture(_png(6, H, 3, rows, bit_depth=4, palette=pal.reshape(-1), trns=[0, 128])))
    (tex,) = pt.SceneLoader().load(str(p)).textures
    alpha = np.full((16,), 255, np.uint8); alpha[0] = 0; alpha[1] = 128
    expanded = np.concatenate([pal[idx], alpha[idx][..., None]], axis=2)
    assert np.array_equal(tex.astype(np.uint8).reshape(-1), expanded.reshape(-1)[:6 * H * 3])
    # grey (1 channel): the reference would read past the decoded buffer -> refused
    grey = rng.integers(0, 256, (H, W), dtype=np.uint8)
    p.write_bytes(_glb_with_texture(_png(W, H, 0, [bytes(r) for r in grey])))
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load(str(p))
    # not a PNG at all (e.g. JPEG bytes)
    p.write_bytes(_glb_with_texture(b"\xff\xd8\xff\xe0" + bytes(64)))
    with pytest.raises(RuntimeError):
        pt.SceneLoader().load(str(p))


@pytest.mark.parametrize("images,buffer", [("file", "file"), ("data", "data"), ("view", "file")])
def test_gltf_text_container_equals_glb(pt, duck_scene, box_scene, tmp_path, images, buffer):
    """.gltf (JSON text, UTF-8 BOM) with the buffer as a percent-encoded file name or a base64 data: URI and the
    images as files, data: URIs or bufferViews loads to the same HostScene as the .glb it was made from."""
    import pngkit
    for model, ref in (("cornell_duck.glb", duck_scene), ("cornell_box.glb", None)):
        d = tmp_path / model.replace(".", "_")
        d.mkdir()
        path = pngkit.glb_to_gltf(os.path.join(ROOT, "models", model), str(d), images=images, buffer=buffer)
        sc = pt.SceneLoader().load(path)
        want = ref if ref is not None else pt.SceneLoader().load(os.path.join(ROOT, "models", model))
        assert np.array_equal(sc.triangles, want.triangles) and np.array_equal(sc.materials, want.materials)
        assert len(sc.textures) == len(want.textures)
        for a, b in zip(sc.textures, want.textures):
            assert np.array_equal(a, b)
    with pytest.raises(RuntimeError):
        (tmp_path / "bad.gltf").write_text('{"asset": {"version": "2.0"}, "buffers": [{"uri": "missing.bin", "byteLength": 4}]}')
        pt.SceneLoader().load(str(tmp_path / "bad.gltf"))


def test_sah_tree_invariants_and_determinism(pt, duck_scene):
    """Opt-in SAH tree (host build, threaded for big inputs): every triangle sits in exactly one leaf
    of <= 4, every child box (inflated) contains its triangles, child links form a tree in which
    pair 0 is the root, and the result does not depend on the thread schedule."""
    for tris in (duck_scene.triangles, pt.make_terrain_scene(300).triangles):      # 4224 and 180002 triangles
        info, pairs, order = pt.host_sah_build(tris)
        info2, pairs2, order2 = pt.host_sah_build(tris)
        assert np.array_equal(pairs.view(np.uint32), pairs2.view(np.uint32)) and np.array_equal(order, order2)
        n = len(tris)
        assert sorted(order.tolist()) == list(range(n)) and info["max_leaf"] <= 4 and info["max_stack"] <= 64
        links = pairs[:, 12:16].view(np.uint32)
        boxes = pairs[:, :12]
        pos = np.stack([tris[v]["position"] for v in ("v0", "v1", "v2")], axis=1)[order]      # leaf order
        seen_pairs = np.zeros(len(pairs), np.int32); seen_pairs[0] = 1
        covered = np.zeros(n, np.int32)
        for p in range(len(pairs)):
            for side in (0, 1):
                link, cnt = int(links[p, 2 * side]), int(links[p, 2 * side + 1])
                b = boxes[p, 6 * side: 6 * side + 6]
                if cnt == 0:
                    assert link > p                      # pre-order: children come later
                    seen_pairs[link] += 1
                elif np.isfinite(b).all():
                    covered[link:link + cnt] += 1
                    pts = pos[link:link + cnt].reshape(-1, 3)
                    assert (pts[:, 0] >= b[0]).all() and (pts[:, 0] <= b[1]).all()
                    assert (pts[:, 1] >= b[2]).all() and (pts[:, 1] <= b[3]).all()
                    assert (pts[:, 2] >= b[4]).all() and (pts[:, 2] <= b[5]).all()
        assert (covered == 1).all() and (seen_pairs == 1).all()


def test_sah_tree_four_wide_device_form(tmp_path):
    """The 4-wide device form of the SAH tree (collapse_sah_quads, what the default SAH kernel walks): built by plain g++
    from the product's host sources, checked on the host -- every node reachable once, every leaf slot covered once,
    well-formed links, absent children that the slab test can never accept (an inverted box of infinities WOULD be
    accepted), the stack bound matching the worst walk order, and the walk itself (restated in the checker) finding
    the brute-force closest hit within that bound.  The GPU suite checks the real walk bit for bit."""
    import subprocess
    csrc = os.path.join(ROOT, "multi-gpu-path-tracer_amd", "csrc")
    exe = str(tmp_path / "quad_collapse_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", csrc, "-I", os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "native", "quad_collapse_check.cpp"),
                    os.path.join(csrc, "pt_bvh.cpp"), os.path.join(csrc, "pt_common.cpp"), "-pthread"], check=True)
    for n, seed in ((1, 1), (2, 2), (3, 3), (4, 4), (5, 5), (9, 6), (17, 7), (37, 8), (500, 9), (3000, 10), (40000, 11)):
        r = subprocess.run([exe, str(n), str(seed)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok:"), (n, seed, r.stdout, r.stderr)
    # the same with the tree's top levels moved to the front (the layout behind PTAMD_SAH_TOP)
    for n, seed, top in ((1, 1, 8), (9, 6, 2), (500, 9, 16), (3000, 10, 64), (40000, 11, 256)):
        r = subprocess.run([exe, str(n), str(seed), str(top)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok:"), (n, seed, top, r.stdout, r.stderr)


def test_missing_native_library_fails_loudly(pt, monkeypatch):
    """No silent fallback: without libptamd.so the binding raises at first use."""
    monkeypatch.setattr(pt.capi, "_lib", None)
    monkeypatch.setattr(pt.capi, "LIB_PATH", "/nonexistent/libptamd.so")
    with pytest.raises(ImportError) as e:
        pt.capi.lib()
    assert "no fallback" in str(e.value)


def test_native_exchange_layout_sizes_match_the_python_path(pt):
    """pt_fb_packed_size (host-only arithmetic of the native gather's layout) against scheduling.packed_size for row
    bands, and: the rectangles of any partition of a frame own every byte of RGB + YUV420 exactly once."""
    import ctypes as C
    lib = pt.capi.lib()

    def size(W, H, tasks):
        arr = (pt.capi.Task * max(len(tasks), 1))()
        for i, t in enumerate(tasks):
            arr[i] = pt.capi.Task(t.width, t.height, t.offset_x, t.offset_y)
        n = C.c_size_t()
        pt.capi.check(lib.pt_fb_packed_size(W, H, arr, len(tasks), C.byref(n)))
        return n.value

    for (W, H, band, world) in [(1920, 1080, 8, 8), (64, 36, 8, 3), (48, 50, 4, 3), (200, 76, 8, 2)]:
        total = 0
        for rank in range(world):
            tasks = pt.row_band_tasks(W, H, band, world, rank)
            assert size(W, H, tasks) == pt.packed_size(W, H, pt.band_row_ranges(H, band, world, rank))
            total += size(W, H, tasks)
        assert total == W * H * 3 + W * H + 2 * (W * H // 4)
    for n_tasks, in_row in [(4, 2), (8, 2), (6, 3), (16, 4)]:                  # FSFL rectangles (TaskGenerator.h:58-80)
        layout = pt.get_task_layout(n_tasks, in_row)
        tasks = pt.TaskGenerator().generateEqualTasks(n_tasks, layout, 1918, 1078)
        assert sum(size(1918, 1078, [t]) for t in tasks) == 1918 * 1078 * 3 + 1918 * 1078 + 2 * (1918 * 1078 // 4)
    assert size(64, 64, []) == 0


def _as_reference_records(pt, tris, fill=0xA5):
    """The scene's triangles as the reference's TRUE 80-byte records (HostScene.h:28-47: float2 is 8-byte aligned under
    nvcc and hipcc, so Vertex is 24 bytes with texCoords at 16), every padding byte set to `fill`."""
    dt = pt.capi.REFERENCE_TRIANGLE_DTYPE
    assert dt.itemsize == 80 and dt.fields["v1"][1] == 24 and dt.fields["textureIdx"][1] == 72
    assert dt.fields["v0"][0].itemsize == 24 and dt.fields["v0"][0].fields["texCoords"][1] == 16
    raw = np.full(len(tris) * 80, fill, np.uint8)
    rec = raw.view(dt)
    for v in ("v0", "v1", "v2"):
        rec[v]["position"] = tris[v]["position"]
        rec[v]["texCoords"] = tris[v]["texCoords"]
    rec["textureIdx"], rec["materialIdx"] = tris["textureIdx"], tris["materialIdx"]
    return rec


def test_reference_triangle_layout_goes_through_the_strided_boundary(pt, duck_scene):
    """the reference's Triangle is 80 bytes, not pt_triangle's 68 -- a cast would shift every vertex.
    The documented route (pt_triangle_layout) reads each field at its own offset."""
    import ctypes as C
    capi = pt.capi
    tris = duck_scene.triangles                      # has real UVs and per-triangle texture / material indices
    for fill in (0xA5, 0x00, 0xFF):                  # padding bytes are never read
        rec = _as_reference_records(pt, tris, fill)
        lay = capi.TriangleLayout.of(rec.dtype)
        assert (lay.stride, list(lay.position), list(lay.texCoords), lay.textureIdx, lay.materialIdx) == \
            (80, [0, 24, 48], [16, 40, 64], 72, 76)                                   # == PT_TRIANGLE_LAYOUT_REFERENCE
        sc = pt.HostScene.from_arrays(rec, duck_scene.materials, duck_scene.textures)
        assert sc.triangles.tobytes() == tris.tobytes()
    # the packed layout through the same entry point, and the conversion on its own at an odd (unaligned) base address
    packed_lay = capi.TriangleLayout.of(capi.TRIANGLE_DTYPE)
    assert (packed_lay.stride, list(packed_lay.position), list(packed_lay.texCoords)) == (68, [0, 20, 40], [12, 32, 52])
    buf = np.zeros(len(rec) * 80 + 1, np.uint8)
    buf[1:] = rec.view(np.uint8)
    out = np.zeros(len(tris), capi.TRIANGLE_DTYPE)
    capi.check(capi.lib().pt_triangles_repack(C.c_void_p(buf.ctypes.data + 1), len(tris), C.byref(lay), capi.np_ptr(out)))
    assert out.tobytes() == tris.tobytes()
    # what the cast would have done: 68-byte stride over 80-byte records is NOT the scene
    wrong = np.frombuffer(rec.view(np.uint8)[:len(tris) * 68].tobytes(), capi.TRIANGLE_DTYPE)
    assert wrong[0]["v0"].tobytes() != tris[0]["v0"].tobytes() and wrong[1].tobytes() != tris[1].tobytes()
    # a field outside the record is refused, not read
    bad = capi.TriangleLayout.of(rec.dtype)
    bad.materialIdx = 78
    with pytest.raises(capi.PtError) as e:
        capi.check(capi.lib().pt_triangles_repack(capi.np_ptr(rec), len(tris), C.byref(bad), capi.np_ptr(out)))
    assert e.value.status == 1
    bad = capi.TriangleLayout.of(rec.dtype)
    bad.stride = 0
    with pytest.raises(capi.PtError):
        capi.check(capi.lib().pt_triangles_repack(capi.np_ptr(rec), len(tris), C.byref(bad), capi.np_ptr(out)))


def test_bench_without_a_launcher_starts_its_own_ranks():
    """VERDICT r2 missing 2: `python bench.py --gpus 2` (no torchrun) used to read WORLD_SIZE = 1 and measure ONE GPU.
    It now starts torch.distributed.run as a child before it touches torch; in this GPU-less container both ranks
    must come up and refuse ("needs an MI355X"), and the launcher must leave with their failure."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU box: the real thing runs in tests/test_gpu_pipeline.py")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    out = r.stdout + r.stderr
    # (the launcher stops the other rank as soon as one has failed: the second refusal may or may not get printed)
    assert out.count("needs an MI355X") >= 1 and "torch.distributed" in out and "FAILED" in out, out[-2000:]
    # and a launcher that disagrees with --gpus is an error, not a silent N = 1
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stdout + r.stderr), (r.stdout + r.stderr)[-2000:]


def test_band_queue_of_the_dqrb_policy(tmp_path):
    """csrc/host/BandQueue.h (VERDICT r2 missing 3): full-width bands of 8-row multiples, 64..128 per frame
    where the height allows, every band pulled exactly once by concurrent workers, dearest-first order from last frame's
    times.  Plain g++ against the host headers; no GPU call is made."""
    import subprocess
    csrc = os.path.join(ROOT, "multi-gpu-path-tracer_amd", "csrc")
    exe = str(tmp_path / "band_queue_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", csrc, "-o", exe, os.path.join(ROOT, "tests", "native", "band_queue_check.cpp"),
                    "-L", os.path.join(ROOT, "multi-gpu-path-tracer_amd"), "-lptamd", "-Wl,-rpath," + os.path.join(ROOT, "multi-gpu-path-tracer_amd"),
                    "-Wl,-rpath,/opt/rocm/lib", "-pthread"], check=True)
    for w, h, t in ((1920, 1080, 8), (3840, 2160, 16), (400, 400, 3), (64, 36, 4), (16, 8, 2)):
        r = subprocess.run([exe, str(w), str(h), str(t)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok:"), (w, h, t, r.stdout, r.stderr)


def test_stream_thread_teardown_is_deterministic_and_race_free(tmp_path):
    """VERDICT r3 item 4 / ADVICE r3: the worker teardown race fixed in 6d7caea (a worker that tested `shouldTerminate`
    after the frame-END barrier, as /root/reference/src/StreamThread.h:94-101 does, could leave while RenderManager::reset()
    waited for it on the frame-START barrier) pinned on the CPU: the product's RenderManager.h / StreamThread.h / barrier.h
    compiled against a stub of the C ABI (no HIP), worker 0 held for 300 ms right behind the frame-end barrier while
    ~RenderManager() runs -- the interleaving that hung.  The current header must terminate (also under ThreadSanitizer,
    with no report); the same program with the old check re-created in the header's test seam must hang (timeout)."""
    import subprocess
    csrc = os.path.join(ROOT, "multi-gpu-path-tracer_amd", "csrc")
    src = [os.path.join(ROOT, "tests", "native", "stream_thread_teardown_check.cpp"), os.path.join(csrc, "pt_common.cpp"), os.path.join(csrc, "pt_bvh.cpp")]
    base = ["g++", "-O1", "-g", "-std=c++17", "-I", csrc, "-pthread"]
    new, old, tsan = str(tmp_path / "teardown_new"), str(tmp_path / "teardown_old"), str(tmp_path / "teardown_tsan")
    subprocess.run(base + ["-o", new] + src, check=True)
    subprocess.run(base + ["-DOLD_CHECK_AFTER_FRAME_END", "-o", old] + src, check=True)
    subprocess.run(base + ["-fsanitize=thread", "-o", tsan] + src, check=True)
    for workers, frames in ((8, 3), (2, 1), (16, 2)):
        r = subprocess.run([new, str(workers), str(frames)], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and r.stdout.startswith("teardown ok"), (workers, frames, r.stdout, r.stderr)
    r = subprocess.run([tsan, "8", "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ThreadSanitizer" not in (r.stdout + r.stderr), (r.stdout, r.stderr[-3000:])
    with pytest.raises(subprocess.TimeoutExpired):          # the pre-fix ordering deadlocks in this interleaving, every time
        subprocess.run([old, "8", "3"], capture_output=True, text=True, timeout=5)
