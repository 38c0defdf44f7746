"""A material whose transparency key changes between frames.

The reference allows it: Renderer::update_material (rend3/src/renderer/mod.rs:256-266) -> MaterialManager::update
(rend3/src/managers/material.rs:163-188) checks only the material TYPE; PbrMaterial::key() is the transparency
(rend3-routine/src/pbr/material.rs:497-499), read again by batch_objects every frame (culling/batching.rs:153).  What the frame of
the change draws follows from three facts:
  (1) the first pass draws LAST frame's predicted triangles through LAST frame's DrawCallSet (forward.rs:224-232): ranges per
      material key as they were then (forward.rs:286, culler.rs:623-640); the pipeline -- cutout discard or not, pbr/routine.rs:61-83
      -- belongs to the range, the material record it reads is the current one (forward.rs:257);
  (2) only atomic-capable objects (opaque + cutout, Sorting::OPAQUE) write predicted triangles (cull.wgsl:361-371); a blend object
      writes residual entries only (cull.wgsl:372-378) -- but its culling RESULTS are saved like everybody's (cull.wgsl:381-388);
  (3) this frame's cull drops a passing triangle from the residual list when the object's previous result says it passed last
      frame (cull.wgsl:365-369; the previous invocation is looked up by object handle, batching.rs:226, whatever its key was).
So, in the frame after the change:
  opaque -> blend : drawn TWICE -- last frame's triangles by the opaque pipeline in the first pass, every passing triangle again by
                    the blend pipeline (depth GreaterEqual: an equal depth passes) over its own opaque image;
  blend -> opaque : a HOLE -- nothing predicted (2), and the triangles that passed last frame are not residual (3): only newly
                    passing ones are drawn; whole again one frame later;
  opaque -> cutout: last frame's triangles drawn WITHOUT the discard in the first pass (1), the discard applies from the next frame;
  cutout -> opaque: the discard still compiled into the first pass, against the new record's alpha_cutout (0.0: nothing fails).
The CPU test pins oracle/world.py on exactly that; the GPU tests compare EVERY frame of the HIP path with it."""
import math

import numpy as np
import pytest

import scenes
from oracle import host as oh
from oracle.world import OracleRenderer
from oracle.world import material_record as omk

f32 = np.float32
W, H = 96, 64
KW = dict(ambient=(0.2, 0.2, 0.2, 1.0), clear_color=(0.0, 0.0, 0.0, 1.0))


def _scene(r, mk, key, alpha=0.5, cutout=None):
    """A quad that fills the middle of the view, material `m` (the one that flips), in front of an opaque red quad."""
    r.set_camera_data(oh.identity(), ("perspective", 60.0, 0.1))
    plane = scenes.plane_mesh(r)
    back = scenes.unlit(r, mk, (1.0, 0.0, 0.0, 1.0))
    m = r.add_material(mk(albedo=(0.0, 1.0, 0.0, alpha), albedo_mode="value", unlit=True, cutout=cutout), key)
    turn = oh.rotation_y(math.pi)  # (the helper's quad faces +z)
    r.add_object(plane, back, oh.mat4_mul(oh.translation((0.0, 0.0, 6.0)), oh.mat4_mul(turn, oh.scale((3.0, 3.0, 1.0)))))
    front = r.add_object(plane, m, oh.mat4_mul(oh.translation((0.0, 0.0, 3.0)), turn))
    return m, front


def _front_pixels(frame, front):
    """pixels whose nearest opaque fragment belongs to object `front`"""
    slot = (frame["vis"] & np.uint64(0xFFFFFFFF)).astype(np.int64) - 1
    tb = frame["tri_base"]
    return int(((slot >= int(tb[front])) & (slot < int(tb[front]) + 2)).sum())


def test_oracle_flip_frames_follow_the_reference():
    o = OracleRenderer(oh.LEFT, f32(W) / f32(H))
    m, front = _scene(o, omk, scenes.OPAQUE)
    rec = o.materials[m][0]
    f0, f1 = o.render(W, H, **KW), o.render(W, H, **KW)
    area = _front_pixels(f1, front)
    assert area > 200 and _front_pixels(f0, front) == area and f1["residual"].sum() == 0  # (second frame: everything predicted)
    # opaque -> blend: drawn twice in the frame of the change
    o.update_material(m, rec, key=scenes.BLEND)
    a0 = o.render(W, H, **KW)
    assert _front_pixels(a0, front) == area, "the first pass still draws last frame's triangles through the opaque range"
    assert len(a0["blend_list"][0]) == 2, "and the blend pass draws the object as well"
    a1 = o.render(W, H, **KW)
    assert _front_pixels(a1, front) == 0 and len(a1["blend_list"][0]) == 2, "one frame later it is a blend object only"
    cy, cx = H // 2, W // 2
    assert not np.array_equal(a0["hdr16"][cy, cx], a1["hdr16"][cy, cx]), "blended over itself vs. over the red quad"
    # blend -> opaque: a hole in the frame of the change
    o.update_material(m, rec, key=scenes.OPAQUE)
    b0 = o.render(W, H, **KW)
    assert _front_pixels(b0, front) == 0 and len(b0["blend_list"][0]) == 0, "neither predicted (blend objects write none) nor residual (it passed last frame)"
    assert b0["pass"][int(b0["tri_base"][front]): int(b0["tri_base"][front]) + 2].all() and b0["residual"].sum() == 0
    b1 = o.render(W, H, **KW)
    assert _front_pixels(b1, front) == area, "whole again one frame later"
    # opaque -> cutout (alpha 0.5 < cutout 0.75): the first pass of the frame of the change has no discard
    o.update_material(m, omk(albedo=(0.0, 1.0, 0.0, 0.5), albedo_mode="value", unlit=True, cutout=0.75), key=scenes.CUTOUT)
    c0 = o.render(W, H, **KW)
    assert _front_pixels(c0, front) == area, "last frame's triangles sit in the opaque range: no discard"
    c1 = o.render(W, H, **KW)
    assert _front_pixels(c1, front) == 0, "from the next frame on every fragment fails the cutout"
    # cutout -> opaque: the predicted triangles of a cutout object are its passing TRIANGLES (the discard is per fragment), drawn
    # through the cutout range against the new record's alpha_cutout = 0.0
    o.update_material(m, rec, key=scenes.OPAQUE)
    d0 = o.render(W, H, **KW)
    assert _front_pixels(d0, front) == area


@pytest.fixture(scope="module")
def r3():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import rend3_amd
    return rend3_amd


@pytest.mark.gpu
@pytest.mark.parametrize("samples", [1, 4])
def test_flip_frames_bit_exact_small(r3, samples):
    """The schedule of the CPU test through the HIP path, every frame compared (sets, keys, HDR)."""
    from test_gpu_parity import compare_frames
    o, p = OracleRenderer(oh.LEFT, f32(W) / f32(H)), r3.Renderer(oh.LEFT, f32(W) / f32(H))
    (m, _), (m2, _) = _scene(o, omk, scenes.OPAQUE), _scene(p, r3.material_record, scenes.OPAQUE)
    assert m == m2
    steps = [(scenes.BLEND, None), (scenes.OPAQUE, None), (scenes.CUTOUT, 0.75), (scenes.OPAQUE, None), (scenes.CUTOUT, 0.25),
             (scenes.BLEND, None), (scenes.CUTOUT, 0.75), (scenes.BLEND, None)]
    for r in (o, p):
        r.render(W, H, samples=samples, **KW)
    for i, (key, cut) in enumerate(steps):
        for r, mk in ((o, omk), (p, r3.material_record)):
            r.update_material(m, mk(albedo=(0.0, 1.0, 0.0, 0.5), albedo_mode="value", unlit=True, cutout=cut), key=key)
        for f in range(2):
            compare_frames(o.render(W, H, samples=samples, **KW), p.render(W, H, samples=samples, **KW), f"step {i} (key {key}) frame {f}")


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [0xF11B, 0x51DE])
def test_flip_frames_bit_exact_scene(r3, seed):
    """Key changes in a lit scene with shadows, blend objects and camera motion: every frame compared, the frame of each change
    included (tests/test_gpu_parity.py::test_material_key_flip_between_frames compared only the third frame after it)."""
    from test_gpu_parity import compare_frames
    o, p = OracleRenderer(oh.LEFT, f32(320) / f32(192)), r3.Renderer(oh.LEFT, f32(320) / f32(192))
    for r, mk in ((o, omk), (p, r3.material_record)):
        scenes.build_random_scene(r, oh, mk, 80, seed, lights=1, with_cutout=True)
        scenes.add_blend_objects(r, oh, mk, seed ^ 0xB1E2E)
    rng = scenes.Pcg32(seed)
    n_mat = len(o.materials)
    frame = 0
    for step in range(8):
        flips = [(rng.randint(n_mat), rng.randint(3)) for _ in range(3)]
        for r in (o, p):
            for m, key in flips:
                r.update_material(m, r.materials[m][0], key=key)
        for f in range(2):
            ang = 0.07 * frame
            for r in (o, p):
                r.set_camera_data(oh.look_at_lh((-2.0 + 3.0 * np.sin(ang), 1.0, -3.0), (0.0, 0.5, 8.0), (0, 1, 0)), ("perspective", 60.0, 0.1))
            kw = dict(ambient=(0.1, 0.1, 0.1, 1.0), clear_color=(0.02, 0.03, 0.05, 1.0))
            compare_frames(o.render(320, 192, **kw), p.render(320, 192, **kw), f"step {step} flips {flips} frame {f}")
            frame += 1
