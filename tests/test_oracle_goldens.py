"""
Pins the CPU oracle against the reference's own golden images:
rend3-test/tests/results/**.png (copied to tests/golden/rend3-test/),
examples/src/cube/screenshot.png (tests/golden/cube-screenshot.png) and
examples/src/static_gltf/{data.glb,screenshot.png} (tests/golden/static_gltf-*).

Each test rebuilds the scene of the reference test it cites and renders it through the oracle's
restatement of BaseRenderGraph::add_to_graph.  `Threshold::Mean(0.0)` goldens are compared
pixel-exactly; the lit ones within the tolerance the reference test itself uses (FLIP is not
available offline, so a per-channel LSB bound stands in for it and is stated in each test).
The MSAA goldens (msaa/four.png, msaa/sample-coverage-4.png, both Threshold::Mean(0.0)) pin row N4: sample
positions, per-sample coverage and the box resolve.
"""
import math
import os

import numpy as np
import pytest
from PIL import Image

import scenes
from oracle import host as hm
from oracle.world import OracleRenderer, material_record as mk

GOLD = os.path.join(os.path.dirname(__file__), "golden")
f32 = np.float32


def load(path):
    return np.array(Image.open(os.path.join(GOLD, path)).convert("RGBA"))


def raw_identity_camera(r):
    r.set_camera_data(hm.identity(), ("raw", hm.identity()))


def test_empty():
    """rend3-test/tests/simple.rs:7-26"""
    r = OracleRenderer(hm.LEFT)
    raw_identity_camera(r)
    out = r.render(64, 64)
    assert np.array_equal(out["rgba8"], load("rend3-test/simple/empty.png"))


@pytest.mark.parametrize(
    "handedness,winding_ccw,visible",
    [(hm.LEFT, False, True), (hm.LEFT, True, False), (hm.RIGHT, False, False), (hm.RIGHT, True, True)],
)
def test_triangle(handedness, winding_ccw, visible):
    """rend3-test/tests/simple.rs:28-84: winding x handedness truth table, top-left rule, OETF."""
    r = OracleRenderer(handedness)
    if winding_ccw:
        pos = [(0.5, -0.5, 0.0), (0.0, 0.5, 0.0), (-0.5, -0.5, 0.0)]
        mh = hm.RIGHT
    else:
        pos = [(0.5, -0.5, 0.0), (-0.5, -0.5, 0.0), (0.0, 0.5, 0.0)]
        mh = hm.LEFT
    mesh = r.add_mesh(pos, mesh_handedness=mh)
    mat = scenes.unlit(r, mk, (0.25, 0.5, 0.75, 1.0))
    r.add_object(mesh, mat, hm.identity())
    raw_identity_camera(r)
    out = r.render(64, 64)
    name = "triangle.png" if visible else "triangle-backface.png"
    assert np.array_equal(out["rgba8"], load("rend3-test/simple/" + name))


COORD_TESTS = [
    ("NegZ", (1, 0, 0), (0, 1, 0), (0, 0, -1)),
    ("Z", (-1, 0, 0), (0, 1, 0), (0, 0, 1)),
    ("NegY", (1, 0, 0), (0, 0, -1), (0, -1, 0)),
    ("Y", (1, 0, 0), (0, 0, 1), (0, 1, 0)),
    ("NegX", (0, 0, -1), (0, 1, 0), (-1, 0, 0)),
    ("X", (0, 0, 1), (0, 1, 0), (1, 0, 0)),
]


def test_coordinate_space():
    """rend3-test/tests/simple.rs:86-141: six triangles, one visible per look direction; one renderer
    renders all six frames, so the two-pass temporal state is exercised too."""
    r = OracleRenderer(hm.LEFT)
    for _name, right, up, camv in COORD_TESTS:
        right, up, camv = (np.array(v, dtype=f32) for v in (right, up, camv))
        pos = [f32(0.5) * right + f32(-0.5) * up, f32(-0.5) * right + f32(-0.5) * up, f32(0.0) * right + f32(0.5) * up]
        mesh = r.add_mesh(pos, mesh_handedness=hm.LEFT)
        neg = bool((camv < 0).any())
        color = camv * f32(-0.25) if neg else camv
        mat = scenes.unlit(r, mk, (color[0], color[1], color[2], 1.0))
        r.add_object(mesh, mat, hm.identity())
    for name, _right, up, camv in COORD_TESTS:
        r.set_camera_data(hm.look_at_lh(camv, (0, 0, 0), up), ("raw", hm.identity()))
        out = r.render(64, 64)
        assert np.array_equal(out["rgba8"], load(f"rend3-test/simple/coordinate-space-{name}.png")), name


def srt(s, t):
    return hm.mat4_mul(hm.translation(t), hm.scale(s))


def test_duplicate_object_retain():
    """rend3-test/tests/object.rs:9-59: deferred removal (enabled == 0) + new object in the residual pass."""
    r = OracleRenderer(hm.LEFT)
    raw_identity_camera(r)
    mat = scenes.unlit(r, mk, (1, 1, 1, 1))
    mesh = scenes.plane_mesh(r)
    o1 = r.add_object(mesh, mat, srt((-0.25, 0.25, 0.25), (-0.5, 0, 0)))
    out = r.render(64, 64)
    assert np.array_equal(out["rgba8"], load("rend3-test/object/duplicate-object-retain-left.png"))
    r.add_object(mesh, mat, srt((-0.25, 0.25, 0.25), (0.5, 0, 0)))
    r.remove_object(o1)
    out = r.render(64, 64)
    assert np.array_equal(out["rgba8"], load("rend3-test/object/duplicate-object-retain-right.png"))


def test_multi_frame_add():
    """rend3-test/tests/object.rs:61-109: object buffer grows 16 -> 32 between frames."""
    r = OracleRenderer(hm.LEFT)
    mat = scenes.unlit(r, mk, (1, 1, 1, 1))
    base = hm.mat4_mul(hm.translation((0.5, 0.5, 0.0)), hm.scale((0.5, 1.0, 1.0)))
    r.set_camera_data(hm.identity(), ("raw", hm.orthographic_lh(0.0, 2.0, 16.0, 0.0, 0.0, 1.0)))
    mesh = scenes.plane_mesh(r)
    for x in range(2):
        for y in range(16):
            r.add_object(mesh, mat, hm.mat4_mul(hm.translation((x, y, 0.0)), base))
        out = r.render(64, 64)
        assert np.array_equal(out["rgba8"], load(f"rend3-test/object/multi-frame-add-{x}.png")), x


def build_msaa_triangle(r, hm, mk):
    """rend3-test/tests/msaa.rs:6-39"""
    mesh = r.add_mesh([(0.5, -0.5, 0.0), (-0.5, -0.5, 0.0), (0.0, 0.5, 0.0)], mesh_handedness=hm.LEFT)
    r.add_object(mesh, scenes.unlit(r, mk, (0.25, 0.5, 0.75, 1.0)), hm.identity())
    r.set_camera_data(hm.identity(), ("raw", hm.identity()))


def test_msaa_four():
    """rend3-test/tests/msaa.rs:6-39, SampleCount::Four, Threshold::Mean(0.0): edge pixels hold k/4 of the colour."""
    r = OracleRenderer(hm.LEFT)
    build_msaa_triangle(r, hm, mk)
    out = r.render(64, 64, samples=4)
    assert np.array_equal(out["rgba8"], load("rend3-test/msaa/four.png"))


def build_sample_coverage(r, hm, mk):
    """rend3-test/tests/msaa.rs:41-82: 64x64 planes, plane (x, y) covers (1 - x/63) x (1 - y/63) of its pixel."""
    mat = scenes.unlit(r, mk, (1, 1, 1, 1))
    base = hm.mat4_mul(hm.translation((0.5, 0.5, 0.0)), hm.scale((0.5, 0.5, 1.0)))
    mesh = scenes.plane_mesh(r)
    for x in range(64):
        for y in range(64):
            sx = f32(1.0) - (f32(x) / f32(63.0))
            sy = f32(1.0) - (f32(y) / f32(63.0))
            m = hm.mat4_mul(hm.mat4_mul(hm.translation((x, y, 0.0)), hm.scale((sx, sy, 1.0))), base)
            r.add_object(mesh, mat, m)
    r.set_camera_data(hm.identity(), ("raw", hm.orthographic_lh(0.0, 64.0, 64.0, 0.0, 0.0, 1.0)))


def test_sample_coverage_4():
    """rend3-test/tests/msaa.rs:41-82 at 4 spp, Threshold::Mean(0.0): pins the four sample positions (each plane
    covers the samples inside its shrinking rectangle), the multisample flag of the cull (no sub-pixel
    rejection, cull.wgsl:292) and the resolve."""
    r = OracleRenderer(hm.LEFT)
    build_sample_coverage(r, hm, mk)
    out = r.render(64, 64, samples=4)
    gold = load("rend3-test/msaa/sample-coverage-4.png")
    # the reference compares RGB only (runner.rs:244-246 FlipImageRgb8): RGB exact.  Alpha of a half-covered pixel is
    # 0.5 -> 127.5: the GPU that made the golden stored 127, `e * 255 + 0.5` gives 128 (a float -> unorm tie)
    assert np.array_equal(out["rgba8"][..., :3], gold[..., :3])
    assert np.abs(out["rgba8"][..., 3].astype(int) - gold[..., 3].astype(int)).max() <= 1


def test_sample_coverage_1():
    """rend3-test/tests/msaa.rs:41-82 at 1 spp: 64x64 planes of shrinking size -- pins the sub-pixel
    cull (cull.wgsl:292-298) against the rasteriser's pixel-centre rule."""
    r = OracleRenderer(hm.LEFT)
    mat = scenes.unlit(r, mk, (1, 1, 1, 1))
    base = hm.mat4_mul(hm.translation((0.5, 0.5, 0.0)), hm.scale((0.5, 0.5, 1.0)))
    mesh = scenes.plane_mesh(r)
    for x in range(64):
        for y in range(64):
            sx = f32(1.0) - (f32(x) / f32(63.0))
            sy = f32(1.0) - (f32(y) / f32(63.0))
            m = hm.mat4_mul(hm.mat4_mul(hm.translation((x, y, 0.0)), hm.scale((sx, sy, 1.0))), base)
            r.add_object(mesh, mat, m)
    r.set_camera_data(hm.identity(), ("raw", hm.orthographic_lh(0.0, 64.0, 64.0, 0.0, 0.0, 1.0)))
    out = r.render(64, 64)
    assert np.array_equal(out["rgba8"], load("rend3-test/msaa/sample-coverage-1.png"))


def _shadow_scene():
    r = OracleRenderer(hm.LEFT)
    r.add_directional_light(color=(1, 1, 1), intensity=1.0, direction=(-1.0, -1.0, 1.0), distance=5.0, resolution=256)
    m1 = scenes.lit(r, mk, (0.25, 0.5, 0.75, 1.0))
    r.add_object(scenes.plane_mesh(r), m1, hm.rotation_x(-math.pi / 2))
    r.set_camera_data(hm.look_at_lh((0.0, 1.0, -1.0), (0, 0, 0), (0, 1, 0)), ("orthographic", (2.5, 2.5, 5.0)))
    return r


def test_shadow_plane():
    """rend3-test/tests/shadow.rs:9-37.  Reference tolerance: FLIP 50th percentile <= 0.04; here: same
    lit-pixel set exactly, every lit pixel within 1 LSB of the golden's [61,86,104]."""
    r = _shadow_scene()
    out = r.render(256, 256)
    gold = load("rend3-test/shadow/plane.png")
    lit_gold = gold[..., :3].any(axis=2)
    lit_ours = out["rgba8"][..., :3].any(axis=2)
    assert lit_gold.sum() == 29376
    assert np.array_equal(lit_gold, lit_ours)
    diff = np.abs(out["rgba8"].astype(int) - gold.astype(int))
    assert diff.max() <= 1, diff.max()
    # every lit pixel is [61, 85, 104] here against the golden's [61, 86, 104]: the exact code of G is 85.47 and no rounding of the
    # Rgba16Float store reaches 86 -- the golden adapter's fixed-function sRGB encode did (profiles/r05_lit_plane_lsb.md).  Pinned:
    # exactly the lit pixels differ, by one LSB, in G only.
    assert (diff.max(axis=2) == 1).sum() == 29376 and diff[..., 0].max() == 0 and diff[..., 2].max() == 0 and diff[..., 3].max() == 0


def test_shadow_cube():
    """rend3-test/tests/shadow.rs:39-54 (same runner: second frame adds the cube)."""
    r = _shadow_scene()
    r.render(256, 256)
    m2 = scenes.lit(r, mk, (0.75, 0.5, 0.25, 1.0))
    cube = scenes.cube_mesh(r)
    r.add_object(cube, m2, srt((0.25, 0.25, 0.25), (0.25, 0.25, -0.25)))
    out = r.render(256, 256)
    gold = load("rend3-test/shadow/cube.png")
    diff = np.abs(out["rgba8"].astype(int) - gold.astype(int)).max(axis=2)
    # reference: 50th percentile of FLIP error <= 0.04.  Here (the bounds are what the restatement achieves): EVERY
    # pixel within 1 LSB, and the 1-LSB pixels pinned by count -- 26 108, of which 25 985 are the lit plane's G channel (85 vs the
    # golden's 86: tools/lit_plane_lsb.py) and 123 penumbra / silhouette pixels.  A regression of the shading, the PCF or the
    # rasteriser's coverage moves these counts.
    assert diff.max() <= 1, diff.max()
    d3 = np.abs(out["rgba8"].astype(int) - gold.astype(int))
    plane_g = (d3[..., 1] == 1) & (d3[..., 0] == 0) & (d3[..., 2] == 0)
    assert abs(int((diff == 1).sum()) - 26108) <= 8, int((diff == 1).sum())
    assert abs(int(plane_g.sum()) - 25985) <= 8, int(plane_g.sum())
    assert np.median(diff) == 0


def test_cube_example():
    """examples/src/cube/mod.rs:70-135,189-200 at 1280x720 (reference threshold: FLIP mean 0.01)."""
    w, h = 1280, 720
    r = OracleRenderer(hm.LEFT, aspect_ratio=f32(w) / f32(h))
    mesh = scenes.cube_mesh(r)
    mat = r.add_material(mk(albedo=(0.5, 0.5, 0.5, 1.0), albedo_mode="value"), scenes.OPAQUE)
    r.add_object(mesh, mat, hm.identity())
    view = hm.mat4_mul(hm.from_euler_xyz(-0.55, 0.5, 0.0), hm.translation((-3.0, -3.0, 5.0)))
    r.set_camera_data(view, ("perspective", 60.0, 0.1))
    r.add_directional_light(color=(1, 1, 1), intensity=1.0, direction=(-1.0, -4.0, 2.0), distance=400.0, resolution=2048)
    r.add_point_light((0.1, 1.2, -1.5), (1.0, 0.0, 0.0), 4.0, 2.0)
    r.add_point_light((1.5, 1.2, -0.1), (0.0, 1.0, 0.0), 4.0, 2.0)
    out = r.render(w, h, clear_color=(0.10, 0.05, 0.10, 1.0))
    gold = np.array(Image.open(os.path.join(GOLD, "cube-screenshot.png")).convert("RGBA"))
    assert tuple(out["rgba8"][0, 0]) == (89, 63, 89, 255)
    bg = np.array([89, 63, 89, 255])
    cov_gold = (gold != bg).any(axis=2)
    cov_ours = (out["rgba8"] != bg).any(axis=2)
    # silhouettes agree except for a handful of edge pixels (float edge placement vs the GPU's snapping)
    assert (cov_gold != cov_ours).sum() <= 4, (cov_gold != cov_ours).sum()  # measured: 2 of 921 600
    diff = np.abs(out["rgba8"].astype(int) - gold.astype(int)).max(axis=2)
    # bounds at what the restatement achieves (mean 0.035 LSB, 99.998 % within 1 LSB, 16 silhouette pixels beyond)
    assert diff.mean() <= 0.05, diff.mean()
    assert (diff <= 1).mean() >= 0.9999, (diff <= 1).mean()
    assert (diff > 1).sum() <= 24, (diff > 1).sum()


def build_static_gltf(r, hm, mk):
    """examples/src/static_gltf/mod.rs:5-41,70-107: a real asset (5 188 vertices, 3 476 triangles, smooth normals) read
    from the reference's data.glb with the product's GLB reader (rend3_amd/gltf.py, row N1)."""
    from rend3_amd.gltf import Gltf
    g = Gltf(os.path.join(GOLD, "static_gltf-data.glb"))
    p = g.primitive(0, 0)
    idx = p["indices"].reshape(-1, 3)[:, ::-1].reshape(-1)  # MeshBuilder::with_flip_winding_order (rend3-types/src/lib.rs:879-887)
    mesh = r.add_mesh(p["positions"], idx, normals=p["normals"], tangents=p["tangents"])
    mat = r.add_material(mk(albedo=g.base_color_factor(p["material"]), albedo_mode="value"), scenes.OPAQUE)
    r.add_object(mesh, mat, hm.scale((1.0, 1.0, -1.0)))
    r.set_camera_data(hm.mat4_mul(hm.from_euler_xyz(-0.55, 0.5, 0.0), hm.translation((-3.0, -3.0, 5.0))), ("perspective", 60.0, 0.1))
    r.add_directional_light(color=(1, 1, 1), intensity=4.0, direction=(-1.0, -4.0, 2.0), distance=20.0, resolution=2048)


def golden_stats(ours, path):
    gold = np.array(Image.open(os.path.join(GOLD, path)).convert("RGBA"))
    diff = np.abs(ours.astype(int) - gold.astype(int)).max(axis=2)
    return gold, diff


def test_static_gltf_example():
    """examples/src/static_gltf/mod.rs:137-148 (reference threshold: FLIP mean <= 0.01) at 1280x720."""
    w, h = 1280, 720
    r = OracleRenderer(hm.LEFT, aspect_ratio=f32(w) / f32(h))
    build_static_gltf(r, hm, mk)
    out = r.render(w, h, clear_color=(0.10, 0.05, 0.10, 1.0))
    gold, diff = golden_stats(out["rgba8"], "static_gltf-screenshot.png")
    bg = np.array([89, 63, 89, 255])
    cov_gold, cov_ours = (gold != bg).any(axis=2), (out["rgba8"] != bg).any(axis=2)
    # measured: silhouettes differ in 2 of 27 773 pixels, 99.998 % of all pixels within 1 LSB, mean |diff| 0.001 LSB
    # (0.019 / 99.90 % with the homogeneous depth quotient of rounds 1-3: self-shadowing speckle)
    assert (cov_gold != cov_ours).sum() <= 16, (cov_gold != cov_ours).sum()
    assert diff.mean() <= 0.005, diff.mean()
    assert (diff <= 1).mean() >= 0.9999, (diff <= 1).mean()


def build_textured_quad(r, hm, mk, resolution=(1280, 720)):
    """examples/src/textured_quad/mod.rs:11-113: a 300-unit quad with checker.png (300x300, Rgba8UnormSrgb, one mip)
    as unlit albedo texture, SampleType::Nearest, orthographic camera sized to the resolution: one texel per pixel."""
    size = 300.0
    pos = [(-size * 0.5, size * 0.5, 0.0), (size * 0.5, size * 0.5, 0.0), (size * 0.5, -size * 0.5, 0.0), (-size * 0.5, -size * 0.5, 0.0)]
    uv = [(0.0, 0.0), (1.0, 0.0), (1.0, 1.0), (0.0, 1.0)]
    mesh = r.add_mesh(pos, indices=[0, 1, 2, 2, 3, 0], mesh_handedness=hm.LEFT, uv0=uv)
    img = np.array(Image.open(os.path.join(GOLD, "textured_quad-checker.png")).convert("RGBA"))
    tex = r.add_texture_2d(img, srgb=True, mip_count=1, mip_source="uploaded")
    mat = r.add_material(mk(albedo_mode="texture", albedo_texture=tex, unlit=True, nearest=True))
    r.add_object(mesh, mat, hm.identity())
    view = hm.mat4_mul(hm.from_euler_xyz(0.0, 0.0, 0.0), hm.translation((0.0, 0.0, 1.0)))
    r.set_camera_data(view, ("orthographic", (float(resolution[0]), float(resolution[1]), 10.0)))


def test_textured_quad_example():
    """examples/src/textured_quad/mod.rs:181-192, Threshold::Mean(0.0) at 1280x720: pins the texture path of row N2
    (uv attribute, perspective-correct interpolation, texel addressing, sRGB decode, nearest sampler)."""
    w, h = 1280, 720
    r = OracleRenderer(hm.LEFT, aspect_ratio=f32(w) / f32(h))
    build_textured_quad(r, hm, mk)
    out = r.render(w, h, clear_color=(0.10, 0.05, 0.10, 1.0))
    gold = load("textured_quad-screenshot.png")
    assert np.array_equal(out["rgba8"][..., :3], gold[..., :3])


def build_skinning_example(r, hm, mk, t=0.0):
    """examples/src/skinning/mod.rs:28-110 at time t (the screenshot is t = 0): RiggedSimple.glb through the GLB
    reader + scene instancing (rend3_amd/gltf.py, row N1)."""
    from rend3_amd.gltf import Gltf, instance_scene
    g = Gltf(os.path.join(GOLD, "skinning-RiggedSimple.glb"))
    r.set_camera_data(hm.mat4_mul(hm.from_euler_xyz(0.0, 0.0, 0.0), hm.translation((0.0, 0.0, 10.0))), ("perspective", 60.0, 0.1))
    inst = instance_scene(g, r, hm, mk)
    set_skinning_pose(r, hm, inst, t)
    r.add_directional_light(color=(1, 1, 1), intensity=10.0, direction=(-1.0, -4.0, 2.0), distance=400.0, resolution=2048)
    return inst


def set_skinning_pose(r, hm, inst, t):
    """examples/src/skinning/mod.rs:143-160: joint transforms {T(0,0,-4.18), Rx(30 deg * sin(5 t))} x inverse bind
    matrices (Skeleton::compute_joint_matrices, rend3-types/src/lib.rs:1233-1239)."""
    ibm = inst["inverse_bind_matrices"][0]
    rot = f32(30.0) * f32(math.sin(5.0 * t))
    glob = [hm.translation((0.0, 0.0, -4.18)),
            hm.mat4_mul(hm.translation((0.0, 0.0, 0.0)), hm.rotation_x(rot * f32(0.017453292519943295)))]
    for sk in inst["skeletons"]:
        r.set_skeleton_joint_matrices(sk, np.array([hm.mat4_mul(glob[i], ibm[i]) for i in range(2)], dtype=f32))


def test_skinning_example():
    """examples/src/skinning/mod.rs:181-192 (reference threshold: FLIP mean <= 0.01) at 1280x720: pins the skinning
    restatement (row S1) and the glTF instancing (row N1) on the reference's own screenshot.

    Two renders.  (1) The frame as the reference graph orders it.  The silhouette differs from the screenshot in 1 of
    46 953 pixels; since round 4 (depth interpolated as the plane through the window-space vertices, oracle/r3o.c
    setup_triangle) 99.99 % of all pixels are within 1 LSB and the mean |diff| is 0.005 LSB.  Until round 3 the depth was
    the homogeneous quotient sum(E_i z_i) / det, and the lit flank carried a speckle of self-shadowed PCF taps the
    screenshot does not show (97.6 % within 1 LSB, mean 0.52): with no depth bias in the reference a lit surface compares
    against its own rasterised depth, and only a rasterised depth that agrees with the vertex depths to the last bits
    reproduces the reference (test_depth_interpolation_experiment).  (2) The same frame with the
    shadow draw skipped (test probe), which isolates skinning + instancing + shading: 99.87 % of all pixels within
    1 LSB, mean 0.015 LSB."""
    w, h = 1280, 720
    bg = np.array([89, 63, 89, 255])
    for skip, max_xor, max_mean, min_le1 in ((False, 8, 0.02, 0.999), (True, 8, 0.03, 0.998)):
        r = OracleRenderer(hm.LEFT, aspect_ratio=f32(w) / f32(h))
        r.skip_shadow_draw = skip
        build_skinning_example(r, hm, mk)
        out = r.render(w, h, clear_color=(0.10, 0.05, 0.10, 1.0))
        gold, diff = golden_stats(out["rgba8"], "skinning-screenshot.png")
        cov_gold, cov_ours = (gold != bg).any(axis=2), (out["rgba8"] != bg).any(axis=2)
        assert (cov_gold != cov_ours).sum() <= max_xor, (cov_gold != cov_ours).sum()
        assert diff.mean() <= max_mean, (skip, diff.mean())
        assert (diff <= 1).mean() >= min_le1, (skip, (diff <= 1).mean())


def build_animation_example(r, hm, mk):
    """examples/src/animation/mod.rs:43-106: scene.gltf (a 34-joint skinned character, one JPEG texture) and cube_3.gltf
    (an animated node), left-handed, one directional light; returns [(instance, animations)] for the poser.
    Assets: tests/golden/animation-*.glb (CC-BY, animation-LICENSE.txt), packed from the reference's example resources by
    tests/golden/make_animation_fixture.py."""
    from rend3_amd.gltf import Gltf, instance_scene, load_animations
    r.set_camera_data(hm.mat4_mul(hm.from_euler_xyz(0.0, 0.0, 0.0), hm.translation((0.0, -1.5, 5.0))), ("perspective", 60.0, 0.1))
    out = []
    for name in ("animation-character.glb", "animation-cube.glb"):
        g = Gltf(os.path.join(GOLD, name))
        out.append((instance_scene(g, r, hm, mk), load_animations(g)))
    r.add_directional_light(color=(1, 1, 1), intensity=5.0, direction=(-1.0, -4.0, 2.0), distance=400.0, resolution=2048)
    return out


def test_animation_example():
    """examples/src/animation/mod.rs:168-178 (reference threshold: FLIP mean <= 0.01) at 1280x720.  The example test runs
    one redraw with delta_t = 0 (examples/src/tests.rs:79): pose_animation_frame(animation 0, time 0) on both scenes, then
    the frame.  This pins rend3-anim's restatement (oracle/anim.py: channel sampling at t = 0, bind components, the
    34-joint hierarchy, joint matrices), the animated node transform, glTF animation / skin loading and the JPEG
    texture path on the reference's own screenshot.

    Measured: the silhouette is IDENTICAL (0 of 100 031 covered pixels differ).  Colours, with the shadow pass as the graph
    orders it: 99.7 % of all pixels within 1 LSB, mean 0.022 LSB (until round 3, with the homogeneous depth quotient: 96 % /
    1.6 LSB -- see test_skinning_example and test_depth_interpolation_experiment); with the shadow draw skipped (test probe)
    the character's half of the image is at mean 0.24 LSB, the rest being the pixels that are genuinely in shadow."""
    from oracle import anim as oa
    w, h = 1280, 720
    bg = np.array([89, 63, 89, 255])
    for skip in (False, True):
        r = OracleRenderer(hm.LEFT, aspect_ratio=f32(w) / f32(h))
        r.skip_shadow_draw = skip
        for inst, anims in build_animation_example(r, hm, mk):
            oa.pose_animation_frame(r, inst, anims, 0, 0.0)
        out = r.render(w, h, clear_color=(0.10, 0.05, 0.10, 1.0))
        gold, diff = golden_stats(out["rgba8"], "animation-screenshot.png")
        cov_gold, cov_ours = (gold != bg).any(axis=2), (out["rgba8"] != bg).any(axis=2)
        assert cov_gold.sum() > 90000 and (cov_gold != cov_ours).sum() <= 8, (cov_gold != cov_ours).sum()
        if not skip:
            assert diff.mean() <= 0.05 and (diff <= 1).mean() >= 0.995, (diff.mean(), (diff <= 1).mean())
        else:
            assert diff[:, 640:].mean() <= 0.4, diff[:, 640:].mean()


def test_tonemap_output_formats():
    """TonemappingRoutine's two fragment entry points (tonemapping.rs:44, blit.wgsl:19-31): *Srgb targets store the exact
    OETF, plain unorm targets get math/color.wgsl's srgb_scene_to_display (exponent 0.4166); Bgra8* targets hold blue
    first.  KATs: SURVEY section 8c's sRGB(0.25, 0.5, 0.75) -> [137, 188, 225] on the exact path; the manual path
    differs from it by at most 1 code anywhere in [0, 1] and is monotone."""
    from oracle import lib as olib
    o = olib.get()
    px = np.array([[0.25, 0.5, 0.75, 1.0]], dtype=np.float16).view(np.uint16)
    outs = {}
    for fmt in range(4):
        out8 = np.zeros((1, 4), np.uint8)
        o.r3o_tonemap_format(o.ptr(px), 1, None, o.ptr(out8), fmt)
        outs[fmt] = out8[0].tolist()
    assert outs[0] == [137, 188, 225, 255] and outs[1] == [225, 188, 137, 255]
    assert outs[3] == outs[2][2::-1] + [255] and max(abs(a - b) for a, b in zip(outs[0], outs[2])) <= 1
    halves = np.arange(0x3C01, dtype=np.uint16)   # [0, 1]
    hdr = np.stack([halves, halves, halves, halves], axis=1).copy()
    exact, manual = np.zeros((len(halves), 4), np.uint8), np.zeros((len(halves), 4), np.uint8)
    o.r3o_tonemap_format(o.ptr(hdr), len(halves), None, o.ptr(exact), 0)
    o.r3o_tonemap_format(o.ptr(hdr), len(halves), None, o.ptr(manual), 2)
    assert (np.diff(manual[:, 0].astype(int)) >= 0).all() and manual[0, 0] == 0 and manual[-1, 0] == 255
    assert np.abs(exact[:, 0].astype(int) - manual[:, 0].astype(int)).max() <= 1
    assert (exact[:, 0] != manual[:, 0]).any()  # 0.4166 is not 1 / 2.4


def test_depth_interpolation_experiment():
    """the one measured oracle-vs-reference gap was the self-shadowed example screenshots.  The oracle's
    experiment switch (r3o_set_depth_mode, tools/depth_mode_experiment.py prints the full table, profiles/r04_depth_modes.md
    holds it) renders them with eight formulations of the rasteriser's depth interpolation.  Mean |LSB| against the reference's
    1280x720 screenshots (animation / skinning / static_gltf):
      sum(E_i z_i) / det (the contract of rounds 1-3)                    1.622 / 0.516 / 0.019
      barycentric weights first, plane from the edge coefficients          the same to three digits
      plane through the window-space vertices (x / w, y / w, z / w)        0.022 / 0.005 / 0.001
      the same gradients from the edge coefficients, anchored at vertex 0  0.038 / 0.006 / 0.001
      ... on vertices snapped to 8 sub-pixel bits                          0.103 / 0.006 / 0.001
    What matters is the ANCHOR: a rasterised depth that reproduces the vertices' own z / w lets a lit surface pass the
    unbiased comparison against itself; the homogeneous forms are mathematically the same plane but carry the rounding of
    |C_i z_i| / det ~ the window size.  The anchored plane is the contract since round 4 (oracle AND kernels:
    device_math.h setup_triangle); snapping makes it worse, so the contract stays unsnapped.  Pinned here on the skinning
    example: the former contract, the new one, and the new one with snapping."""
    from oracle.lib import get as ol
    w, h = 1280, 720
    stats = {}
    try:
        for mode in (7, 0, 3):
            ol().r3o_set_depth_mode(mode)
            r = OracleRenderer(hm.LEFT, aspect_ratio=f32(w) / f32(h))
            build_skinning_example(r, hm, mk)
            out = r.render(w, h, clear_color=(0.10, 0.05, 0.10, 1.0))
            _gold, diff = golden_stats(out["rgba8"], "skinning-screenshot.png")
            stats[mode] = (float(diff.mean()), float((diff <= 1).mean()))
    finally:
        ol().r3o_set_depth_mode(0)
    assert 0.4 <= stats[7][0] <= 0.6 and stats[7][1] <= 0.98, stats   # what rounds 1-3 measured
    assert stats[0][0] <= 0.02 and stats[0][1] >= 0.999, stats         # the contract
    assert stats[3][0] <= 0.05 and stats[3][0] >= stats[0][0], stats   # snapping does not help


def test_op_tally_build_is_the_same_oracle_and_counts():
    """oracle/tally.h ("exact count from the oracle's op tally"): r3o.c compiled with a counting f32 gives
    bit-identical frames, and the resolve's count per shaded pixel is where the survey's estimate puts it (80 + 130 per light for the
    untextured formula, a few hundred more with three trilinear maps and the tangent frame)."""
    import scenes
    from oracle import host as oh
    from oracle import lib as olib
    from oracle.world import OracleRenderer, material_record
    tally_lib = olib.OracleLib(tally=True)
    frames = []
    for lib_ in (None, tally_lib):
        r = OracleRenderer(oh.LEFT, np.float32(160) / np.float32(96), lib=lib_)
        scenes.build_textured_scene(r, oh, material_record, 40, 0x7A11, lights=2)
        r.set_camera_data(oh.look_at_lh((0.0, 2.0, -6.0), (0.0, 0.5, 6.0), (0, 1, 0)), ("perspective", 60.0, 0.1))
        frames.append((r.render(160, 96, ambient=(0.1, 0.1, 0.1, 1.0), clear_color=(0.0, 0.0, 0.0, 1.0)), r))
    (a, _), (b, rt) = frames
    for k in ("vis", "hdr16", "rgba8", "pass", "residual"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["atlas"].view(np.uint32), b["atlas"].view(np.uint32))
    shaded = int((b["vis"] != 0).sum())
    t = rt.stage_tally["shade"]
    per_px = t["fs_main"]["flops"] / max(shaded, 1)
    assert shaded > 2000 and 300 < per_px < 3000, (shaded, per_px, t)
    assert t["fs_main"]["sqrt"] > 0 and t["fs_main"]["div"] > 0 and t["vs_main"]["flops"] > 0 and t["fixed_function"]["flops"] > 0
    assert t["flops"] == t["fs_main"]["flops"] + t["vs_main"]["flops"] + t["fixed_function"]["flops"]
