"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle.

Bar: bit-exact.  The render path's floating-point contract makes every
operation a single IEEE op or an explicit fmaf, so RGB/YUV bytes and the work counters must
be IDENTICAL to the oracle's, not merely close.
"""
import numpy as np
import pytest

from conftest import BOX_CAMERA, DUCK_CAMERA

pytestmark = pytest.mark.gpu


def _render_gpu(pt, scene, camera, W, H, spp, depth, tasks=None, device=0, mode=0, coop=-1):
    cam = pt.CameraConfig(camera[0], camera[1], camera[2], camera[3])
    fb = pt.Framebuffer(pt.Resolution(W, H), device="cuda:%d" % device)
    tracer = pt.DevicePathTracer(device, spp, depth, (8, 8), scene, fb, cam)
    tracer.setTraversalMode(mode)
    tracer.setCooperativeMode(coop)
    tracer.resetCounters()
    for t in (tasks or [pt.RenderTask(W, H, 0, 0)]):
        tracer.renderTaskAsync(t)
    tracer.waitForRenderTask()
    rgb, yuv, cnt = fb.rgb_host().copy(), fb.yuv_host().copy(), tracer.readCounters()
    tracer.close()
    return rgb, yuv, cnt


def _assert_same(gpu, ref):
    g_rgb, g_yuv, g_cnt = gpu
    r_rgb, r_yuv, r_cnt = ref
    diff = np.argwhere(g_rgb != r_rgb)
    assert diff.size == 0, "RGB differs at %d bytes, first %s: gpu %s ref %s" % (
        len(diff), diff[0], g_rgb[tuple(diff[0])], r_rgb[tuple(diff[0])])
    assert np.array_equal(g_yuv, r_yuv), "YUV differs"
    for k in ("paths", "segments", "box_tests", "tri_tests", "light_tests"):
        assert g_cnt[k] == r_cnt[k], (k, g_cnt[k], r_cnt[k])


def test_device_arithmetic_is_ieee_and_matches_oracle(pt, ob, box_scene, gpu_device):
    """sin/cos polynomial, /, sqrt, double-divide-by-pi and XORWOW: device == host, bitwise."""
    fb = pt.Framebuffer(pt.Resolution(8, 8))
    tr = pt.DevicePathTracer(0, 1, 1, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    rng = np.random.default_rng(7)
    # sincos over the path's domain (0, 2*pi]
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 20000), [0.0, np.pi / 4, np.pi / 2, np.pi, 2 * np.pi, 6.2831855]]).astype(np.float32)
    s, c = tr.katArith(0, x)
    ref = np.array([ob.sincos(float(v)) for v in x], np.float32)
    assert np.array_equal(s.view(np.uint32), ref[:, 0].view(np.uint32))
    assert np.array_equal(c.view(np.uint32), ref[:, 1].view(np.uint32))
    assert np.max(np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64)))) < 3e-7
    assert np.max(np.abs(c.astype(np.float64) - np.cos(x.astype(np.float64)))) < 3e-7
    # division / sqrt: correctly rounded, incl. denormal results and huge ratios
    a = np.concatenate([rng.standard_normal(50000) * 10.0 ** rng.integers(-30, 30, 50000), [1.0, 1.0, 0.0, 1e-40]]).astype(np.float32)
    b = np.concatenate([rng.standard_normal(50000) * 10.0 ** rng.integers(-30, 30, 50000), [3.0, 0.0, 0.0, 1e30]]).astype(np.float32)
    with np.errstate(all="ignore"):
        q, _ = tr.katArith(1, a, b)
        assert np.array_equal(q.view(np.uint32), (a / b).view(np.uint32))
        r, _ = tr.katArith(2, np.abs(a))
        assert np.array_equal(r.view(np.uint32), np.sqrt(np.abs(a)).view(np.uint32))
        d, _ = tr.katArith(3, a)
        assert np.array_equal(d.view(np.uint32), (a.astype(np.float64) / np.pi).astype(np.float32).view(np.uint32))
    # XORWOW: uniforms #0..#7 for the seeds of pixel 0, pixel 1 and the last pixel of 1080p
    for seed in (1984, 1985, 1984 + 1920 * 1080 - 1):
        _, ref_u = ob.xorwow_uniforms(seed, 8)
        seeds = np.full(8, seed, np.uint32).view(np.float32)
        skips = np.arange(8, dtype=np.uint32).view(np.float32)
        got, _ = tr.katArith(4, seeds, skips)
        assert np.array_equal(got.view(np.uint32), ref_u.view(np.uint32))
    tr.close()


def test_short_division_forms_equal_ieee_for_every_float(pt, box_scene, gpu_device):
    """The kernels' short reciprocal (v_rcp + Newton step + v_div_fixup, IEEE sequence outside
    2^-126..2^126), x/pi (double multiply + FMA residual) and sqrt (v_rsq + residual step, IEEE sequence
    below 2^-100) forms against the IEEE expressions they
    replace, over ALL 2^32 float bit patterns: not one differing bit (pt_math.h, pt_kat_exhaustive).
    The IEEE expressions themselves are pinned to the host's by the sampled test above."""
    fb = pt.Framebuffer(pt.Resolution(8, 8))
    tr = pt.DevicePathTracer(0, 1, 1, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    for op in (0, 1, 2, 4):
        bad, examples = tr.katExhaustive(op, 0, 1 << 32, 8)
        assert bad == 0, (op, bad, ["%08x" % b for b in examples])
    # the checker itself must be able to fail: the unguarded core is wrong for |x| > 2^126 (op 3)
    bad, _ = tr.katExhaustive(3, 0x7f000000, 1 << 23, 0)
    assert bad > 0
    tr.close()


@pytest.mark.parametrize("name", ["box", "duck"])
def test_device_bvh_equals_oracle_bvh(pt, ob, box_scene, duck_scene, box_oracle, duck_oracle, gpu_device, name):
    scene, osc = (box_scene, box_oracle) if name == "box" else (duck_scene, duck_oracle)
    fb = pt.Framebuffer(pt.Resolution(8, 8))
    tr = pt.DevicePathTracer(0, 1, 1, (8, 8), scene, fb, pt.CameraConfig((0, 0, 0), (0, 0, -1)))
    b, l, i, lights = tr.bvhCopy()
    ob_b, ob_l, ob_i, ob_lights = osc.bvh()
    assert np.array_equal(b.view(np.uint32), ob_b.view(np.uint32))
    assert np.array_equal(l, ob_l) and np.array_equal(i, ob_i) and np.array_equal(lights, ob_lights)
    tr.close()


@pytest.mark.parametrize("name,n", [("box", 4000), ("duck", 600)])
def test_closest_hit_matches_oracle_on_random_rays(pt, box_scene, duck_scene, box_oracle, duck_oracle, gpu_device, name, n):
    scene, osc, cam = (box_scene, box_oracle, BOX_CAMERA) if name == "box" else (duck_scene, duck_oracle, DUCK_CAMERA)
    fb = pt.Framebuffer(pt.Resolution(8, 8))
    tr = pt.DevicePathTracer(0, 1, 1, (8, 8), scene, fb, pt.CameraConfig(cam[0], cam[1]))
    rng = np.random.default_rng(3)
    tris = scene.triangles
    lo = np.min([tris[v]["position"].min(axis=0) for v in ("v0", "v1", "v2")], axis=0)
    hi = np.max([tris[v]["position"].max(axis=0) for v in ("v0", "v1", "v2")], axis=0)
    org = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    dirs = rng.standard_normal((n, 3)).astype(np.float32)
    dirs[::7, 0] = 0.0          # axis-parallel components: 1/0 = inf in the slab test
    dirs[::11, 1] = 0.0
    t, face = tr.katClosestHit(org, dirs)
    mats = tris["materialIdx"]
    for k in range(n):
        h, rt, rm = osc.closest_hit(org[k], dirs[k])
        if h:
            assert face[k] >= 0 and np.float32(rt).view(np.uint32) == t[k].view(np.uint32), (k, rt, t[k])
            assert mats[face[k]] == rm
        else:
            assert face[k] == -1
    tr.close()


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("W,H,spp,depth", [(64, 64, 4, 4), (200, 120, 8, 8), (50, 34, 3, 1)])
def test_cornell_box_image_bit_exact(pt, box_scene, box_oracle, gpu_device, W, H, spp, depth, mode):
    """Every traversal schedule (1 = per-lane two-phase walk, 2 / 3 = wave-uniform walks, 4 = mask walk with
    dealt sparse leaves) must give the oracle's bytes AND the oracle's exact test counts."""
    ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    _assert_same(_render_gpu(pt, box_scene, BOX_CAMERA, W, H, spp, depth, mode=mode), ref)


def test_wave_uniform_mode_is_refused_for_big_trees(pt, duck_scene, gpu_device):
    fb = pt.Framebuffer(pt.Resolution(8, 8))
    tr = pt.DevicePathTracer(0, 1, 1, (8, 8), duck_scene, fb, pt.CameraConfig(DUCK_CAMERA[0], DUCK_CAMERA[1]))
    with pytest.raises(pt.capi.PtError) as e:
        tr.setTraversalMode(2)          # 332 inner nodes > 64
    assert e.value.status == 6
    tr.setTraversalMode(3)              # fine for the stack walk
    tr.close()


def test_cornell_box_config1_400x400_16spp_4bounce(pt, box_scene, box_oracle, gpu_device):
    """BASELINE.json configs[0]: the reference's own CPU-runnable case, at full size."""
    ref = box_oracle.render(BOX_CAMERA, 400, 400, 16, 4, threads=8)
    _assert_same(_render_gpu(pt, box_scene, BOX_CAMERA, 400, 400, 16, 4), ref)


@pytest.mark.parametrize("mode,coop", [(1, -1), (3, 0), (3, 1), (5, -1), (5, 1), (0, -1)])
@pytest.mark.parametrize("W,H,spp,depth", [(64, 36, 2, 4), (96, 54, 3, 8)])
def test_cornell_duck_image_bit_exact(pt, duck_scene, duck_oracle, gpu_device, W, H, spp, depth, mode, coop):
    """(Mode 5: the stack walk with dealt sparse leaves, the duck's default; 0: whatever the library picks.)
    Per-lane walk, wave-uniform stack walk with one wave per tile, and the cooperative kernel (four
    waves per tile, each testing a quarter of every leaf, closest hits merged by the reference's visit
    order): bytes and all counters equal the oracle's."""
    ref = duck_oracle.render(DUCK_CAMERA, W, H, spp, depth, threads=8)
    _assert_same(_render_gpu(pt, duck_scene, DUCK_CAMERA, W, H, spp, depth, mode=mode, coop=coop), ref)


def test_task_partition_invariance(pt, box_scene, box_oracle, gpu_device):
    """RNG state is never written back (DevicePathTracer.h:80): the image is independent of
    how the frame is cut into RenderTasks -- ragged rectangles included."""
    W, H, spp, depth = 100, 76, 4, 5
    whole = _render_gpu(pt, box_scene, BOX_CAMERA, W, H, spp, depth)
    tasks = [pt.RenderTask(37, 30, 0, 0), pt.RenderTask(63, 30, 37, 0), pt.RenderTask(100, 1, 0, 30),
             pt.RenderTask(100, 45, 0, 31), pt.RenderTask(0, 10, 5, 5)]          # width 0 = no-op (:195)
    parts = _render_gpu(pt, box_scene, BOX_CAMERA, W, H, spp, depth, tasks=tasks)
    _assert_same(parts, whole)
    _assert_same(whole, box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8))


def test_edge_cases(pt, box_scene, box_oracle, gpu_device):
    # depth 0: ray_color returns black without tracing; spp 1; single-pixel-wide task
    for (W, H, spp, depth) in [(16, 16, 2, 0), (16, 16, 1, 2), (2, 2, 5, 3)]:
        ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth)
        got = _render_gpu(pt, box_scene, BOX_CAMERA, W, H, spp, depth)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    # error behaviour at the boundary
    fb = pt.Framebuffer(pt.Resolution(16, 16))
    tr = pt.DevicePathTracer(0, 1, 1, (8, 8), box_scene, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    with pytest.raises(pt.capi.PtError) as e:
        tr.renderTaskAsync(pt.RenderTask(17, 4, 0, 0))        # leaves the framebuffer
    assert e.value.status == 1
    tr.close()
    # a scene without emissive faces is undefined behaviour in the reference -> explicit error
    tris = box_scene.triangles
    mats = box_scene.materials.copy()
    mats["emissiveFactor"] = 0
    dark = pt.HostScene.from_arrays(tris, mats)
    with pytest.raises(pt.capi.PtError) as e:
        pt.DevicePathTracer(0, 1, 1, (8, 8), dark, fb, pt.CameraConfig(BOX_CAMERA[0], BOX_CAMERA[1]))
    assert e.value.status == 3


@pytest.mark.parametrize("mode", [1, 3, 5])
def test_synthetic_terrain_bit_exact(pt, ob, gpu_device, mode):
    """BASELINE.json configs[3] in miniature: the deterministic terrain generator at grid 48
    (4610 triangles, leaves of hundreds of triangles, traversal stack 10+ deep) against the oracle,
    through both the per-lane walk (what the 8M-triangle scene uses) and the stack walk."""
    terr = pt.make_terrain_scene(48)
    cam = ((0.0, 220.0, 620.0), (0.0, -0.35, -1.0), 45.0, 45.0)
    osc = ob.OracleScene(terr.triangles, terr.materials, terr.textures)
    W, H, spp, depth = 64, 36, 2, 6
    ref = osc.render(cam, W, H, spp, depth, threads=8)
    assert ref[2]["max_stack"] >= 5
    got = _render_gpu(pt, terr, cam, W, H, spp, depth, mode=mode)
    _assert_same(got, ref)
    assert got[0].max() > 0          # the terrain is lit


def _textured_scene(pt):
    """A hand-made scene that exercises what the Cornell models do not: a base-colour texture AND
    an emissive texture, UVs outside [0,1) (fmod + clamp paths, Texture.h:38-58), the row-`height`
    read (Texture.h:68), and the texture-pointer inheritance between materials
    (DevicePathTracer.h:269-279: material 2 has no texture of its own and inherits material 1's)."""
    rng = np.random.default_rng(5)
    tex0 = rng.integers(0, 256, (8, 16, 3)).astype(np.float32)           # base colour, 16x8
    tex1 = np.full((4, 4, 3), 255.0, np.float32)                          # emissive mask: one dark texel.  A path
    tex1[1, 2] = 0.0                                                      # that lands on it scatters off the light's own
    # plane, gets pdf 0 / 0 and poisons its pixel with NaN (-> byte 0): the reference's arithmetic, kept
    mats = np.zeros(4, pt.capi.MATERIAL_DTYPE)
    mats["baseColorTextureIdx"] = -1; mats["emissiveTextureIdx"] = -1
    mats[0]["baseColor"] = (0.8, 0.8, 0.8)                                # plain floor
    mats[1]["baseColor"] = (1.0, 0.9, 0.8); mats[1]["baseColorTextureIdx"] = 0
    mats[2]["baseColor"] = (0.5, 1.0, 0.5)                                # inherits texture 0 (quirk)
    mats[3]["baseColor"] = (0.0, 0.0, 0.0); mats[3]["emissiveFactor"] = (1.0, 0.8, 0.6); mats[3]["emissiveTextureIdx"] = 1
    quads = [   # (corner, edge u, edge v, material, uv scale)
        ((-4, 0, -4), (8, 0, 0), (0, 0, 8), 0, 1.0),                      # floor y=0, normal up?  (see winding below)
        ((-4, 0, -4), (0, 5, 0), (8, 0, 0), 1, 2.5),                      # back wall z=-4, UVs up to 2.5 (wrap)
        ((-4, 0, 4), (0, 5, 0), (0, 0, -8), 2, -1.5),                     # left wall x=-4, negative UVs
        ((-1.5, 4.9, -1.5), (0, 0, 3), (3, 0, 0), 3, 1.0),                # textured light under the ceiling
    ]
    tris = []
    for (c, eu, ev, m, s) in quads:
        c, eu, ev = np.array(c, np.float32), np.array(eu, np.float32), np.array(ev, np.float32)
        p = [c, c + eu, c + eu + ev, c + ev]
        uv = [(0, 0), (s, 0), (s, s), (0, s)]
        for (a, b, d) in ((0, 2, 1), (0, 3, 2)):
            t = np.zeros((), pt.capi.TRIANGLE_DTYPE)
            for name, k in (("v0", a), ("v1", b), ("v2", d)):
                t[name]["position"] = p[k]; t[name]["texCoords"] = uv[k]
            t["materialIdx"] = m
            tris.append(t)
    return np.array(tris, pt.capi.TRIANGLE_DTYPE), mats, [tex0, tex1]


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5])
def test_textures_emissive_texture_and_inheritance_quirk(pt, ob, gpu_device, mode):
    tris, mats, texs = _textured_scene(pt)
    scene = pt.HostScene.from_arrays(tris, mats, texs)
    osc = ob.OracleScene(tris, mats, texs)
    cam = ((0.5, 2.0, 7.0), (0.0, 0.0, -1.0), 60.0, 60.0)
    W, H, spp, depth = 72, 48, 6, 5
    ref = osc.render(cam, W, H, spp, depth, threads=8)
    got = _render_gpu(pt, scene, cam, W, H, spp, depth, mode=mode)
    _assert_same(got, ref)
    assert len(np.unique(ref[0].reshape(-1, 3), axis=0)) > 50         # textured, lit, not flat


@pytest.mark.parametrize("parts", [4, 8])
def test_textured_scene_with_parts_that_walk_by_lanes(pt, ob, gpu_device, monkeypatch, parts):
    """The textured scene under the dealt mask walk with its dearest tiles split: their parts test all child boxes at once
    (part_walk_by_lanes) and the winner's barycentrics are recomputed for the texture lookup.  Frames 1-4 (raster order, cost
    order, split tiles) equal the oracle's, counters included."""
    monkeypatch.setenv("PTAMD_SPLIT_PCT", "10")
    monkeypatch.setenv("PTAMD_DEAL_SPLIT", str(parts))
    tris, mats, texs = _textured_scene(pt)
    scene = pt.HostScene.from_arrays(tris, mats, texs)
    osc = ob.OracleScene(tris, mats, texs)
    cam = ((0.5, 2.0, 7.0), (0.0, 0.0, -1.0), 60.0, 60.0)
    W, H, spp, depth = 200, 168, 3, 5
    ref = osc.render(cam, W, H, spp, depth, threads=8)
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), scene, fb, pt.CameraConfig(cam[0], cam[1], cam[2], cam[3]))
    tr.setTraversalMode(4)
    seen = 0
    for frame in range(4):
        fb.rgb.zero_(); fb.yuv.zero_()
        tr.resetCounters()
        tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
        assert "dealt<" in tr.kernelName()
        _assert_same((fb.rgb_host(), fb.yuv_host(), tr.readCounters()), ref)
        seen = max(seen, tr.splitTiles())
    assert seen > 0
    tr.close()


@pytest.mark.parametrize("parts,cap", [(4, 64), (2, 64), (8, 768)])
def test_unsplit_root_scene_with_split_tiles(pt, ob, gpu_device, monkeypatch, parts, cap):
    """Two triangles: the reference's root stays a leaf and is its own two children (bvh.h:26), so every ray lists its
    triangles twice.  With the dearest tiles split, parts of 16 pixels and a 64-entry item list the list is exactly full
    (16 rays x 2 triangles x 2) and the parts walk by lanes; parts of 32 pixels would overflow it and keep the pair walk.
    Frames 1-4 equal the oracle's, counters included."""
    monkeypatch.setenv("PTAMD_SPLIT_PCT", "10")
    monkeypatch.setenv("PTAMD_DEAL_SPLIT", str(parts))
    monkeypatch.setenv("PTAMD_DEAL_CAP", str(cap))
    tris = np.zeros(2, pt.capi.TRIANGLE_DTYPE)
    for k, (a, b, c, m) in enumerate([((-6, 0, -6), (6, 0, 6), (6, 0, -6), 0), ((-2, 4, -2), (2, 4, -2), (0, 4, 2), 1)]):
        for name, pos in (("v0", a), ("v1", b), ("v2", c)):
            tris[k][name]["position"] = pos
        tris[k]["materialIdx"] = m
    mats = np.zeros(2, pt.capi.MATERIAL_DTYPE)
    mats["baseColorTextureIdx"] = -1; mats["emissiveTextureIdx"] = -1
    mats[0]["baseColor"] = (0.7, 0.6, 0.5)
    mats[1]["emissiveFactor"] = (1.0, 0.9, 0.8)
    scene = pt.HostScene.from_arrays(tris, mats, [])
    osc = ob.OracleScene(tris, mats, [])
    cam = ((0.0, 2.0, 9.0), (0.0, -0.1, -1.0), 60.0, 60.0)
    W, H, spp, depth = 200, 168, 3, 5
    ref = osc.render(cam, W, H, spp, depth, threads=8)
    assert ref[2]["tri_tests"] % 4 == 0 and ref[2]["tri_tests"] > 0      # both triangles, through both "children", per box hit
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), scene, fb, pt.CameraConfig(cam[0], cam[1], cam[2], cam[3]))
    tr.setTraversalMode(4)
    for frame in range(4):
        fb.rgb.zero_(); fb.yuv.zero_()
        tr.resetCounters()
        tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
        _assert_same((fb.rgb_host(), fb.yuv_host(), tr.readCounters()), ref)
    tr.close()


def test_c5_parameters_small_frame(pt, box_scene, box_oracle, gpu_device):
    """BASELINE.json configs[4] parameters (1024 spp, 16 bounces) on a small frame."""
    ref = box_oracle.render(BOX_CAMERA, 32, 16, 1024, 16, threads=8)
    _assert_same(_render_gpu(pt, box_scene, BOX_CAMERA, 32, 16, 1024, 16), ref)


def test_depth_zero_counts_camera_rays_and_renders_black(pt, box_scene, box_oracle, gpu_device):
    """recursionDepth 0: camera rays are generated (paths = pixels x spp), nothing is traced, the frame is
    black (camera.h:49-83 returns (0,0,0)).  Found by tools/fuzz_parity.py: the kernels used to report 0 paths."""
    ref = box_oracle.render(BOX_CAMERA, 24, 16, 3, 0)
    assert ref[2]["paths"] == 24 * 16 * 3 and ref[2]["segments"] == 0 and not ref[0].any()
    for mode in (0, 1, 2, 4, 5):
        _assert_same(_render_gpu(pt, box_scene, BOX_CAMERA, 24, 16, 3, 0, mode=mode), ref)


def _random_scene(pt, seed):
    """Random triangle soup in [-10,10]^3 with random materials, 1-3 emissive triangles, random UVs
    and (sometimes) a texture: nothing about it is tuned to any walk's fast path."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(6, 400)) if seed < 8 else int(rng.integers(6, 90))      # seeds >= 8: small trees (mask walk)
    tris = np.zeros(n, pt.capi.TRIANGLE_DTYPE)
    centre = rng.uniform(-8, 8, (n, 3)).astype(np.float32)
    size = rng.uniform(0.3, 5.0, (n, 1, 1)).astype(np.float32)
    verts = centre[:, None, :] + (rng.standard_normal((n, 3, 3)).astype(np.float32) * size)
    if seed % 3 == 0:                                   # some exactly axis-aligned / coplanar geometry
        verts[: n // 3, :, 1] = np.float32(-6.0)
    for k, name in enumerate(("v0", "v1", "v2")):
        tris[name]["position"] = verts[:, k, :]
        tris[name]["texCoords"] = rng.uniform(-1.5, 2.5, (n, 2)).astype(np.float32)
    n_mats = int(rng.integers(2, 6))
    mats = np.zeros(n_mats, pt.capi.MATERIAL_DTYPE)
    mats["baseColor"] = rng.uniform(0.1, 0.9, (n_mats, 3)).astype(np.float32)
    mats["baseColorTextureIdx"] = -1; mats["emissiveTextureIdx"] = -1
    mats[0]["emissiveFactor"] = rng.uniform(0.2, 1.0, 3).astype(np.float32)
    texs = []
    if seed % 2 == 0:
        texs = [rng.integers(0, 256, (int(rng.integers(1, 9)), int(rng.integers(1, 9)), 3)).astype(np.float32)]
        mats[n_mats - 1]["baseColorTextureIdx"] = 0
    tris["materialIdx"] = rng.integers(1, n_mats, n)
    lights = rng.choice(n, int(rng.integers(1, 4)), replace=False)
    tris["materialIdx"][lights] = 0
    for k, name in enumerate(("v0", "v1", "v2")):          # lights: big triangles around the middle
        tris[name]["position"][lights] = (rng.standard_normal((len(lights), 3)) * 6.0).astype(np.float32)
    eye = rng.standard_normal(3); eye = eye / np.linalg.norm(eye) * rng.uniform(14, 30)
    front = -eye / np.linalg.norm(eye) + rng.standard_normal(3) * 0.05
    cam = (tuple(eye.astype(np.float32).tolist()), tuple(front.astype(np.float32).tolist()),
           float(rng.uniform(30, 90)), float(rng.uniform(30, 90)))
    return tris, mats, texs, cam


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_scenes_bit_exact_in_every_walk(pt, ob, gpu_device, seed):
    tris, mats, texs, cam = _random_scene(pt, seed)
    scene = pt.HostScene.from_arrays(tris, mats, texs)
    osc = ob.OracleScene(tris, mats, texs)
    W, H, spp, depth = 40, 24, 3, 6
    ref = osc.render(cam, W, H, spp, depth, threads=8)
    info = pt.host_bvh_build(tris)[0]
    modes = [1, 3, 5] + ([2, 4] if info["nodes"] - info["leaves"] <= 64 else [])
    for mode in modes:
        _assert_same(_render_gpu(pt, scene, cam, W, H, spp, depth, mode=mode, coop=0), ref)
    _assert_same(_render_gpu(pt, scene, cam, W, H, spp, depth, mode=3, coop=1), ref)      # cooperative tiles, small leaves too


# ---- extension: legacy material classes + spheres (pt_set_legacy_scene) -----------------

def _legacy_records(pt, kinds):
    rec = np.zeros(len(kinds), pt.capi.LEGACY_MATERIAL_DTYPE)
    for i, (kind, albedo, param) in enumerate(kinds):
        rec[i]["kind"] = kind; rec[i]["albedo"] = albedo; rec[i]["param"] = param
    return rec


def _render_gpu_legacy(pt, scene, legacy, spheres, camera, W, H, spp, depth):
    fb = pt.Framebuffer(pt.Resolution(W, H))
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), scene, fb, pt.CameraConfig(camera[0], camera[1], camera[2], camera[3]))
    tr.setLegacyScene(legacy, spheres)
    tr.resetCounters()
    tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
    got = (fb.rgb_host().copy(), fb.yuv_host().copy(), tr.readCounters())
    # switching the extension off again restores the plain path
    tr.setLegacyScene(None, None)
    tr.resetCounters()
    tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
    plain = (fb.rgb_host().copy(), fb.yuv_host().copy(), tr.readCounters())
    tr.close()
    return got, plain


def test_legacy_materials_in_the_cornell_box_match_the_oracle(pt, ob, box_scene, box_oracle, gpu_device):
    """Cornell box with the walls re-declared as legacy lambertian / metal / dielectric where their names
    say so, one glass and one mirror sphere added; the light stays the live emissive material, so universal
    and legacy materials meet on the same paths.  Kernel == oracle, bytes and counters."""
    W, H, spp, depth = 96, 64, 8, 8
    n = len(box_scene.materials)
    kinds = []
    for i in range(n):
        name = box_scene.material_names[i]
        emissive = box_scene.materials["emissiveFactor"][i].max() > 0
        if emissive: kinds.append((0, (0, 0, 0), 0.0))                               # the light: live material
        elif i % 3 == 0: kinds.append((1, tuple(box_scene.materials["baseColor"][i]), 0.0))   # lambertian
        elif i % 3 == 1: kinds.append((2, (0.8, 0.85, 0.9), 0.15))                   # fuzzy metal
        else: kinds.append((0, (0, 0, 0), 0.0))                                      # stays universal
    legacy = _legacy_records(pt, kinds)
    # add glass + mirror materials? spheres reuse existing material slots: pick a metal and make slot 0.. dielectric
    di = next(i for i, k in enumerate(kinds) if k[0] == 1)
    legacy[di]["kind"] = 3; legacy[di]["param"] = 1.5                               # one lambertian slot becomes glass
    me = next(i for i, k in enumerate(kinds) if k[0] == 2)
    spheres = np.zeros(2, pt.capi.SPHERE_DTYPE)
    spheres[0] = ((-180.0, 120.0, 250.0), 90.0, di)
    spheres[1] = ((-380.0, 300.0, 180.0), 70.0, me)
    box_oracle.set_legacy(legacy, spheres)
    try:
        ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    finally:
        box_oracle.set_legacy(None, None)
    plain_ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    got, plain = _render_gpu_legacy(pt, box_scene, legacy, spheres, BOX_CAMERA, W, H, spp, depth)
    _assert_same(got, ref)
    _assert_same(plain, plain_ref)
    assert (ref[0] != plain_ref[0]).mean() > 0.2                                      # it is a different picture
    assert ref[0].mean() > 3


def test_pure_legacy_scene_with_spheres_matches_the_oracle(pt, ob, gpu_device):
    """No live material at all: lambertian ground, diffuse_light quad, metal / glass / lambertian spheres
    (a hollow glass sphere via a negative radius).  Accepted without any light-sampled face."""
    tris = np.zeros(4, pt.capi.TRIANGLE_DTYPE)
    quad = lambda k, a, b, c, d, m: [  # noqa: E731  (wound so that the geometric normal of the ground points up)
        (k, a, c, b, m), (k + 1, a, d, c, m)]
    for k, a, b, c, m in quad(0, (-30, 0, -30), (30, 0, -30), (30, 0, 30), (-30, 0, 30), 0) + \
            quad(2, (-7, 9, -7), (7, 9, -7), (7, 9, 7), (-7, 9, 7), 1):
        tris[k]["v0"]["position"] = a; tris[k]["v1"]["position"] = b; tris[k]["v2"]["position"] = c
        tris[k]["materialIdx"] = m
    mats = np.zeros(5, pt.capi.MATERIAL_DTYPE)
    mats["baseColorTextureIdx"] = -1; mats["emissiveTextureIdx"] = -1
    legacy = _legacy_records(pt, [(1, (0.5, 0.6, 0.4), 0.0), (4, (4.0, 3.7, 3.4), 0.0), (2, (0.9, 0.7, 0.3), 0.0),
                                  (3, (0, 0, 0), 1.5), (1, (0.2, 0.3, 0.8), 0.0)])
    spheres = np.zeros(4, pt.capi.SPHERE_DTYPE)
    spheres[0] = ((-3.0, 1.5, 0.0), 1.5, 2)
    spheres[1] = ((0.5, 1.5, 1.0), 1.5, 3)
    spheres[2] = ((0.5, 1.5, 1.0), -1.3, 3)
    spheres[3] = ((3.5, 1.0, -1.0), 1.0, 4)
    cam = ((0.0, 4.0, 12.0), (0.0, -0.25, -1.0), 40.0, 55.0)
    W, H, spp, depth = 80, 48, 16, 10
    scene = pt.HostScene.from_arrays(tris, mats, [])
    osc = ob.OracleScene(tris, mats, [])
    osc.set_legacy(legacy, spheres)
    ref = osc.render(cam, W, H, spp, depth, threads=8)
    fb = pt.Framebuffer(pt.Resolution(W, H))
    cc = pt.CameraConfig(cam[0], cam[1], cam[2], cam[3])
    # without the extension this scene has no light-sampled face: refused
    with pytest.raises(pt.capi.PtError):
        pt.DevicePathTracer(0, spp, depth, (8, 8), scene, fb, cc)
    tr = pt.DevicePathTracer(0, spp, depth, (8, 8), scene, fb, cc, legacyMaterials=legacy, spheres=spheres)
    tr.resetCounters()
    tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
    _assert_same((fb.rgb_host().copy(), fb.yuv_host().copy(), tr.readCounters()), ref)
    assert ref[0].mean() > 5 and len(np.unique(ref[0].reshape(-1, 3), axis=0)) > 100
    # the extension needs the reference tree, and the uniform walks are not compiled for it
    with pytest.raises(pt.capi.PtError):
        tr.setBvhMode(1)
    tr.setBvhMode(0)
    with pytest.raises(pt.capi.PtError):
        tr.setTraversalMode(2)
    tr.close()


def test_per_lane_test_counters_fold_before_they_can_wrap(pt, ob, duck_scene, duck_oracle, gpu_device, monkeypatch):
    """The per-lane walks count box / triangle tests in 32-bit per-lane registers; spp x depth x tests per segment
    passes 2^32 at C5-class sample counts on big reference trees, so the kernels fold them into wider counters at
    2^31.  Here the fold point is lowered to 2^10 (PTAMD_FOLD_BITS, read at pt_device_create), which makes every
    lane fold hundreds of times in a small frame: the totals must still be the oracle's, in the per-lane walk of
    the reference tree (per-lane high words) and in the SAH kernel (wave-uniform spill words)."""
    monkeypatch.setenv("PTAMD_FOLD_BITS", "10")
    W, H, spp, depth = 64, 36, 2, 4
    ref = duck_oracle.render(DUCK_CAMERA, W, H, spp, depth, threads=8)
    assert ref[2]["tri_tests"] > (W * H) << 12                      # every lane passes the fold point many times
    _assert_same(_render_gpu(pt, duck_scene, DUCK_CAMERA, W, H, spp, depth, mode=1), ref)
    terr = pt.make_terrain_scene(48)
    cam = ((0.0, 220.0, 620.0), (0.0, -0.35, -1.0), 45.0, 45.0)
    fb = pt.Framebuffer(pt.Resolution(W, H))
    plain = pt.DevicePathTracer(0, 64, 8, (8, 8), terr, fb, pt.CameraConfig(*cam), bvhMode=1)
    monkeypatch.delenv("PTAMD_FOLD_BITS")
    fb2 = pt.Framebuffer(pt.Resolution(W, H))
    wide = pt.DevicePathTracer(0, 64, 8, (8, 8), terr, fb2, pt.CameraConfig(*cam), bvhMode=1)
    got = []
    for tr in (plain, wide):
        tr.resetCounters()
        tr.renderTaskAsync(pt.RenderTask(W, H, 0, 0)); tr.waitForRenderTask()
        got.append(tr.readCounters())
        tr.close()
    assert got[0] == got[1] and got[0]["box_tests"] > (W * H) << 11
    assert np.array_equal(fb.rgb_host(), fb2.rgb_host())


@pytest.mark.parametrize("deal_min,deal_cap", [(65, 768), (65, 64), (20, 768), (0, 768)])
def test_dealt_walk_is_exact_whatever_is_dealt(pt, ob, box_scene, box_oracle, gpu_device, monkeypatch, deal_min, deal_cap):
    """The dealt walk (mode 4) moves the triangle tests of leaves hit by fewer than PTAMD_DEAL_MIN lanes onto the
    wave's item list and merges their closest hits by (t, visit rank); leaves above the threshold, and leaves whose
    items do not fit the list (PTAMD_DEAL_CAP), are tested in place.  Every split must give the oracle's bytes and
    counters: everything dealt (65), everything dealt into a list that overflows at once (cap 64), a low threshold,
    nothing dealt (0 = the plain mask walk's tests run through this kernel) -- on the box, on a textured scene
    (winner's barycentrics recomputed) and on a random soup with exact ties."""
    monkeypatch.setenv("PTAMD_DEAL_MIN", str(deal_min))
    monkeypatch.setenv("PTAMD_DEAL_CAP", str(deal_cap))
    W, H, spp, depth = 104, 56, 6, 8                         # ragged right edge: lanes without a pixel help in the rounds
    ref = box_oracle.render(BOX_CAMERA, W, H, spp, depth, threads=8)
    _assert_same(_render_gpu(pt, box_scene, BOX_CAMERA, W, H, spp, depth, mode=4), ref)
    tris, mats, texs = _textured_scene(pt)
    cam = ((0.5, 2.0, 7.0), (0.0, 0.0, -1.0), 60.0, 60.0)
    ref = ob.OracleScene(tris, mats, texs).render(cam, 72, 48, 6, 5, threads=8)
    _assert_same(_render_gpu(pt, pt.HostScene.from_arrays(tris, mats, texs), cam, 72, 48, 6, 5, mode=4), ref)
    for seed in (9, 10):                                      # small random soups, coplanar geometry -> equal-t ties
        tris, mats, texs, cam = _random_scene(pt, seed)
        ref = ob.OracleScene(tris, mats, texs).render(cam, 40, 24, 3, 6, threads=8)
        _assert_same(_render_gpu(pt, pt.HostScene.from_arrays(tris, mats, texs), cam, 40, 24, 3, 6, mode=4), ref)
