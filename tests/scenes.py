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


# ------------------------------------------------------------------ procedural meshes for synthetic scenes
def icosphere(subdiv):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache = {}
        nf = []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    pos = np.array(v, dtype=f32)
    idx = np.array(f, dtype=np.uint32).reshape(-1)
    return pos, idx, pos.copy()  # unit sphere: normal == position


def box(sx=1.0, sy=1.0, sz=1.0):
    pos = np.array(CUBE_POS, dtype=f32) * np.array([sx, sy, sz], dtype=f32)
    nrm = np.repeat(np.array([(0, 0, 1), (0, 0, -1), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0)], dtype=f32), 4, 0)
    return pos, np.array(CUBE_IDX, dtype=np.uint32), nrm


def grid_plane(n, size=1.0):
    """(n x n) quads in the XZ plane facing +Y, CW-from-above winding for a LH renderer."""
    xs = np.linspace(-size, size, n + 1, dtype=f32)
    pos = np.array([(x, 0.0, z) for z in xs for x in xs], dtype=f32)
    idx = []
    for j in range(n):
        for i in range(n):
            a = j * (n + 1) + i
            b = a + 1
            c = a + n + 1
            d = c + 1
            idx += [a, c, b, b, c, d]
    nrm = np.tile(np.array([0, 1, 0], dtype=f32), (len(pos), 1))
    return pos, np.array(idx, dtype=np.uint32), nrm


class Pcg32:
    """PCG32 (O'Neill), the generator SURVEY.md section 8d names for the synthetic configs."""

    def __init__(self, seed, seq=54):
        self.state = 0
        self.inc = ((seq << 1) | 1) & 0xFFFFFFFFFFFFFFFF
        self.next_u32()
        self.state = (self.state + seed) & 0xFFFFFFFFFFFFFFFF
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & 0xFFFFFFFFFFFFFFFF
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * (self.next_u32() / 4294967296.0)

    def randint(self, n):
        return self.next_u32() % n


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
