"""CPU-only: (1) librend3_amd.so loads and exports every symbol include/r3n.h declares; (2) the product's host
mirror (rend3_amd/csrc/host.cpp) agrees bit-for-bit with the oracle's numpy restatement (oracle/host.py);
(3) the reference's own CPU known-answer tests for the path: the shadow-atlas known-answer cases of
rend3/src/managers/directional/shadow_alloc.rs:225-319."""
import math
import re

import numpy as np
import pytest

import scenes
from oracle import host as oh
from oracle.lib import get as oracle_lib
from rend3_amd import _ffi
from rend3_amd import host as ph

f32 = np.float32


def test_library_exports_every_declared_symbol():
    import os
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "r3n.h")).read()
    declared = set(re.findall(r"\b(r3n_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"r3n_ctx"}
    lib = _ffi.lib()
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _ffi.lib()
    assert not lib.r3n_create(0, None)
    assert b"HIP device" in lib.r3n_create_error() or lib.r3n_create_error()


def rand_mats(n, seed):
    rng = scenes.Pcg32(seed)
    out = []
    for _ in range(n):
        m = oh.mat4_mul(oh.mat4_mul(oh.translation((rng.uniform(-50, 50), rng.uniform(-5, 5), rng.uniform(-50, 50))),
                                    scenes.random_rotation(rng, oh)),
                        oh.scale((rng.uniform(0.2, 4), rng.uniform(0.2, 4), rng.uniform(0.2, 4))))
        out.append(m)
    return out


def eq(a, b):
    return np.array_equal(np.asarray(a, dtype=f32).view(np.uint32), np.asarray(b, dtype=f32).view(np.uint32))


def test_matrix_ops_bit_exact():
    ms = rand_mats(64, 0xABCD)
    for a, b in zip(ms[:-1], ms[1:]):
        assert eq(oh.mat4_mul(a, b), ph.mat4_mul(a, b))
        assert eq(oh.mat4_inverse(a), ph.mat4_inverse(a))
    for eye, c, up in [((3, 3, -5), (0, 0, 0), (0, 1, 0)), ((0, 1, -1), (0, 0, 0), (0, 1, 0)), ((-17.2, 3.7, -4.6), (1, 2, 3), (0, 1, 0))]:
        assert eq(oh.look_at_lh(eye, c, up), ph.look_at_lh(eye, c, up))
        assert eq(oh.look_at_rh(eye, c, up), ph.look_at_rh(eye, c, up))
    assert eq(oh.orthographic_lh(0, 64, 64, 0, 0, 1), ph.orthographic_lh(0, 64, 64, 0, 0, 1))


@pytest.mark.parametrize("hand", [oh.LEFT, oh.RIGHT])
def test_camera_state_frustum_uniform_blocks_bit_exact(hand):
    ol = oracle_lib()
    for proj in [("perspective", 60.0, 0.1), ("orthographic", (2.5, 2.5, 5.0)), ("raw", oh.orthographic_lh(0, 2, 16, 0, 0, 1))]:
        view = oh.mat4_mul(oh.from_euler_xyz(-0.55, 0.5, 0.0), oh.translation((-3, -3, 5)))
        a = oh.CameraState(view, proj, hand, f32(1280) / f32(720))
        b = ph.CameraState(view, proj, hand, f32(1280) / f32(720))
        for k in ("proj", "view_proj", "inv_view", "origin_view_proj", "location"):
            assert eq(getattr(a, k), getattr(b, k)), (proj[0], k)
        assert eq(oh.camera_header(a, None, (1280, 720), 1, 16, ol), ph.camera_header(b, None, (1280, 720), 1, 16))
        assert eq(oh.camera_header(a, 2, (256, 256), 1, 32, ol), ph.camera_header(b, 2, (256, 256), 1, 32))
        assert eq(oh.frame_uniforms(a, (0.1, 0.1, 0.1, 1), (1280, 720), ol), ph.frame_uniforms(b, (0.1, 0.1, 0.1, 1), (1280, 720)))


def test_shadow_camera_and_light_buffers_bit_exact():
    view = oh.mat4_mul(oh.from_euler_xyz(-0.04430086, -4.6065736, 0.0), oh.translation((17.174278, -3.715882, 4.631997)))
    for hand in (oh.LEFT, oh.RIGHT):
        a = oh.CameraState(view, ("perspective", 60.0, 0.1), hand, 16 / 9)
        b = ph.CameraState(view, ("perspective", 60.0, 0.1), hand, 16 / 9)
        lights = [dict(color=(1, 1, 1), intensity=15.0, direction=(1, -5, -1), distance=100.0, resolution=2048),
                  dict(color=(1, 0.5, 0.2), intensity=2.0, direction=(-1, -2, 3), distance=50.0, resolution=1024),
                  None,
                  dict(color=(0.3, 1, 1), intensity=1.0, direction=(0.2, -1, 0.1), distance=20.0, resolution=512)]
        sa, sha, ba = oh.evaluate_directional_lights(lights, a)
        sb, shb, bb = ph.evaluate_directional_lights(lights, b)
        assert sa == sb
        assert ba == bb.tobytes()
        for x, y in zip(sha, shb):
            assert x["offset"] == y["offset"] and x["size"] == y["size"] and x["handle"] == y["handle"]
            assert eq(x["camera"].view_proj, y["camera"].view_proj)
    pts = [dict(position=(0.1, 1.2, -1.5), color=(1, 0, 0), intensity=4.0, radius=2.0), None]
    assert oh.point_light_buffer(pts) == ph.point_light_buffer(pts).tobytes()


def test_bounding_sphere_and_normals_bit_exact():
    pos, idx, _ = scenes.icosphere(2)
    ca, ra = oh.bounding_sphere_from_mesh(pos)
    cb, rb = ph.bounding_sphere_from_mesh(pos)
    assert eq(ca, cb) and eq([ra], [rb])
    for m in rand_mats(16, 77):
        c1, r1 = oh.bounding_sphere_apply_transform(ca, ra, m)
        c2, r2 = ph.bounding_sphere_apply_transform(cb, rb, m)
        assert eq(c1, c2) and eq([r1], [r2])
    for lh in (True, False):
        assert eq(oh.calculate_normals(scenes.CUBE_POS, scenes.CUBE_IDX, lh), ph.calculate_normals(scenes.CUBE_POS, scenes.CUBE_IDX, lh))
        assert eq(oh.calculate_normals(pos, idx, lh), ph.calculate_normals(pos, idx, lh))


def test_frustum_planes_and_sphere_test_match_oracle():
    ol = oracle_lib()
    view = oh.look_at_lh((0, 5, -20), (0, 0, 0), (0, 1, 0))
    cam = oh.CameraState(view, ("perspective", 60.0, 0.1), oh.LEFT, 16 / 9)
    pa, pb = oh.frustum_planes(cam.view_proj, ol), ph.frustum_planes(cam.view_proj)
    assert eq(pa, pb)
    rng = scenes.Pcg32(5)
    lib = _ffi.lib()
    for _ in range(2000):
        c = np.array([rng.uniform(-60, 60), rng.uniform(-30, 30), rng.uniform(-60, 60)], dtype=f32)
        r = f32(rng.uniform(0, 8))
        assert ol.r3o_frustum_contains_sphere(ol.ptr(pa), ol.ptr(c), r) == lib.r3n_host_frustum_contains_sphere(_ffi.ptr(pb), _ffi.ptr(c), r)


# ---- rend3/src/managers/directional/shadow_alloc.rs:225-319: the reference's own known-answer tests for
# allocate_shadow_atlas, restated verbatim (expected map ORDER included).  The other five tests of that module
# (chunk_subdivision_*, :146-223) pin the private ShadowNode tree, which both restatements rebuild internally.
ATLAS_KATS = [
    # allocate_single
    ([(0, 16)], 16, (16, 16), [((0, 0), 16, 0)]),
    # allocate_single_level_single_row
    ([(0, 16), (1, 16), (2, 16)], 48, (48, 16), [((0, 0), 16, 0), ((16, 0), 16, 1), ((32, 0), 16, 2)]),
    # allocate_single_level_double_row
    ([(0, 16), (1, 16), (2, 16)], 32, (32, 32), [((0, 0), 16, 0), ((16, 0), 16, 1), ((0, 16), 16, 2)]),
    # allocate_single_level_double_row_extra_space
    ([(0, 16), (1, 16), (2, 16), (3, 16), (4, 16)], 64, (48, 32),
     [((0, 0), 16, 0), ((16, 0), 16, 1), ((32, 0), 16, 2), ((0, 16), 16, 3), ((16, 16), 16, 4)]),
    # allocate_multiple_level
    ([(0, 16), (1, 8), (2, 8), (3, 4), (4, 4), (5, 4)], 32, (32, 16),
     [((0, 0), 16, 0), ((16, 0), 8, 1), ((24, 0), 8, 2), ((16, 8), 4, 3), ((20, 8), 4, 4), ((16, 12), 4, 5)]),
]


@pytest.mark.parametrize("maps,maxdim,dims,expect", ATLAS_KATS)
def test_shadow_atlas_known_answers(maps, maxdim, dims, expect):
    for impl in (oh.allocate_shadow_atlas, ph.allocate_shadow_atlas):
        got_dims, got = impl(maps, maxdim)
        assert tuple(got_dims) == dims
        assert [(tuple(o), s, h) for o, s, h in got] == expect


def test_shadow_atlas_empty_and_zero_dimension():
    for impl in (oh.allocate_shadow_atlas, ph.allocate_shadow_atlas):
        assert impl([], 32) is None
        assert impl([(0, 16)], 0) is None


def test_shadow_atlas_random_agree():
    rng = scenes.Pcg32(99)
    for _ in range(50):
        n = 1 + rng.randint(9)
        maps = [(i, 1 << (4 + rng.randint(6))) for i in range(n)]
        a = oh.allocate_shadow_atlas(maps, 2048)
        b = ph.allocate_shadow_atlas(maps, 2048)
        assert a[0] == b[0] and a[1] == b[1]


@pytest.mark.parametrize("hand", [oh.LEFT, oh.RIGHT])
def test_evaluate_frame_matches_the_oracle_host_math(hand):
    """r3n_host_evaluate_frame (host.cpp) -- the CPU half of a frame behind one C call -- against the oracle's numpy restatement
    of the same reference code: CameraState (camera.rs:23-85), DirectionalLightManager::evaluate (directional.rs:99-157: atlas,
    shadow cameras, light buffer), FrameUniforms::new (uniforms.rs:28-48), PerCameraUniform headers (culler.rs:477-502).
    Every output byte is compared; ctypes struct layouts are the ones layouts.h pins."""
    import ctypes as ct
    lib = _ffi.lib()
    ol = oracle_lib()
    lights = [dict(color=(1.0, 0.9, 0.8), intensity=15.0, direction=(1.0, -5.0, -1.0), distance=100.0, resolution=2048),
              None,
              dict(color=(0.2, 0.4, 1.0), intensity=3.5, direction=(-0.3, -1.0, 0.2), distance=40.0, resolution=512),
              dict(color=(1.0, 1.0, 1.0), intensity=1.0, direction=(0.1, -1.0, 0.0), distance=400.0, resolution=1024),
              dict(color=(0.5, 0.5, 0.5), intensity=2.0, direction=(2.0, -1.0, 0.5), distance=25.0, resolution=2048)]
    cases = [(("perspective", 60.0, 0.1), f32(16 / 9), 1), (("orthographic", (2.5, 2.5, 5.0)), None, 4),
             (("raw", oh.orthographic_lh(0, 2, 16, 0, 0, 1)), f32(1.25), 1)]
    for (proj, aspect, samples), view in zip(cases, rand_mats(3, 0x51DE)):
        view = oh.mat4_inverse(view)
        for use in (lights, lights[:1], []):
            cam = _ffi.HostCamera144()
            ct.memmove(cam.view, np.ascontiguousarray(view, dtype=f32).ctypes.data, 64)
            cam.handedness = 1 if hand == oh.RIGHT else 0
            cam.aspect_ratio = 0.0 if aspect is None else float(aspect)
            cam.projection_kind = {"orthographic": 0, "perspective": 1, "raw": 2}[proj[0]]
            params = np.zeros(16, dtype=f32)
            if proj[0] == "perspective":
                params[:2] = proj[1:]
            elif proj[0] == "orthographic":
                params[:3] = proj[1]
            else:
                params[:] = proj[1]
            ct.memmove(cam.projection_params, params.ctypes.data, 64)
            la = np.zeros((max(len(use), 1), 12), dtype=f32)
            for i, l in enumerate(use):
                if l is not None:
                    la[i, 0:3], la[i, 3], la[i, 4:7], la[i, 7] = l["color"], l["intensity"], l["direction"], l["distance"]
                    la[i, 8:9].view(np.uint32)[0] = l["resolution"]
            amb = (ct.c_float * 4)(0.1, 0.2, 0.3, 1.0)
            fr = _ffi.HostFrame()
            assert lib.r3n_host_evaluate_frame(ct.byref(cam), la.ctypes.data, len(use), 16384, amb, 1920, 1080, samples, 4096, ct.byref(fr)) == 0
            ocam = oh.CameraState(view, proj, hand, aspect)
            size, shadows, dir_buf = oh.evaluate_directional_lights(use, ocam)
            assert (fr.shadow_atlas_width, fr.shadow_atlas_height) == tuple(size) and fr.n_shadow_views == len(shadows)
            assert bytes(fr.directional_buffer)[: fr.directional_bytes] == bytes(dir_buf) and fr.directional_bytes == len(bytes(dir_buf))
            assert bytes(fr.uniforms) == oh.frame_uniforms(ocam, (0.1, 0.2, 0.3, 1.0), (1920, 1080), ol).tobytes()
            assert bytes(fr.viewport_header) == oh.camera_header(ocam, None, (1920, 1080), samples, 4096, ol).tobytes()
            assert eq(fr.camera_location[:], ocam.location)
            for k, sh in enumerate(shadows):
                v = fr.shadow_views[k]
                assert (v.x, v.y, v.size, fr.shadow_handles[k]) == (sh["offset"][0], sh["offset"][1], sh["size"], sh["handle"])
                assert bytes(v.header) == oh.camera_header(sh["camera"], k, (sh["size"], sh["size"]), 1, 4096, ol).tobytes()

