"""
Scene builders shared by the oracle tests and the GPU parity tests.  Every builder takes a
`renderer` exposing the Renderer-shaped API (add_mesh/add_material/add_object/...): both
oracle.world.OracleRenderer and rend3_amd.Renderer implement it, so the same scene runs
through both.  `hm` is the host-math module matching the renderer (oracle.host or rend3_amd.host).

The scenes restate the reference's own tests:
  rend3-test/tests/simple.rs, object.rs, msaa.rs, shadow.rs; rend3-test/src/helpers.rs;
  examples/src/cube/mod.rs.
"""
import math

import numpy as np

f32 = np.float32
LEFT, RIGHT = 0, 1
OPAQUE, CUTOUT, BLEND = 0, 1, 2


def plane_mesh(r):
    """rend3-test/src/helpers.rs:55-74"""
    pos = [(-1, -1, 0), (-1, 1, 0), (1, 1, 0), (1, -1, 0)]
    return r.add_mesh(pos, [0, 2, 1, 0, 3, 2], mesh_handedness=LEFT)


CUBE_POS = [
    (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1),
    (-1, 1, -1), (1, 1, -1), (1, -1, -1), (-1, -1, -1),
    (1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1),
    (-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (-1, -1, -1),
    (1, 1, -1), (-1, 1, -1), (-1, 1, 1), (1, 1, 1),
    (1, -1, 1), (-1, -1, 1), (-1, -1, -1), (1, -1, -1),
]
CUBE_IDX = [0, 1, 2, 2, 3, 0, 4, 5, 6, 6, 7, 4, 8, 9, 10, 10, 11, 8, 12, 13, 14, 14, 15, 12,
            16, 17, 18, 18, 19, 16, 20, 21, 22, 22, 23, 20]


def cube_mesh(r):
    """rend3-test/src/helpers.rs:77-130 == examples/src/cube/mod.rs:5-52"""
    return r.add_mesh(CUBE_POS, CUBE_IDX, mesh_handedness=LEFT)


def unlit(r, mk, color):
    return r.add_material(mk(albedo=color, albedo_mode="value", unlit=True), OPAQUE)


def lit(r, mk, color):
    return r.add_material(mk(albedo=color, albedo_mode="value", unlit=False), OPAQUE)


from rend3_amd.scenes import Pcg32, box, grid_plane, icosphere  # noqa: E402,F401  (input generators, shared with bench.py)


def random_rotation(rng, hm):
    return hm.mat4_mul(hm.mat4_mul(hm.rotation_y(rng.uniform(0, 2 * math.pi)), hm.rotation_x(rng.uniform(0, 2 * math.pi))),
                       hm.rotation_z(rng.uniform(0, 2 * math.pi)))


def build_random_scene(r, hm, mk, n_objects, seed, extent=(30.0, 8.0, 30.0), n_materials=8, handedness=LEFT,
                       lights=1, shadow_res=256, shadow_distance=60.0, with_cutout=False):
    """Small 'scifi-like' scene (SURVEY section 8d cfg 2 shape, reduced): instanced procedural meshes with random
    transforms inside a box around the origin, camera at the box centre."""
    rng = Pcg32(seed)
    meshes = []
    for sub in (0, 1, 2):
        p, i, n = icosphere(sub)
        if handedness == LEFT:
            i = i.reshape(-1, 3)[:, ::-1].reshape(-1)  # outward faces CW for LH front-face
        meshes.append(r.add_mesh(p, i, normals=n))
    p, i, n = box()
    if handedness == RIGHT:
        i = i.reshape(-1, 3)[:, ::-1].reshape(-1)
    meshes.append(r.add_mesh(p, i, normals=n))
    mats = []
    for k in range(n_materials):
        col = (rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), rng.uniform(0.2, 1.0), 1.0)
        key = OPAQUE
        cutout = None
        if with_cutout and k % 4 == 3:
            key, cutout = CUTOUT, 0.5
            col = col[:3] + (0.25 if k % 8 == 3 else 0.75,)
        rec = mk(albedo=col, albedo_mode="value", roughness=rng.uniform(0.2, 0.9),
                 metallic=1.0 if rng.uniform() < 0.2 else 0.0, cutout=cutout)
        mats.append(r.add_material(rec, key))
    handles = []
    for _ in range(n_objects):
        pos = (rng.uniform(-extent[0], extent[0]), rng.uniform(-extent[1], extent[1]), rng.uniform(-extent[2], extent[2]))
        s = math.exp(rng.uniform(math.log(0.25), math.log(4.0)))
        xf = hm.mat4_mul(hm.mat4_mul(hm.translation(pos), random_rotation(rng, hm)), hm.scale((s, s, s)))
        handles.append(r.add_object(meshes[rng.randint(len(meshes))], mats[rng.randint(len(mats))], xf))
    for k in range(lights):
        ang = 2 * math.pi * k / max(lights, 1)
        r.add_directional_light(color=(1, 1, 1), intensity=3.0,
                                direction=(math.cos(ang) - 0.3, -2.0, math.sin(ang) + 0.2),
                                distance=shadow_distance, resolution=shadow_res)
    return handles
