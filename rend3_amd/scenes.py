"""Synthetic input generators for the named BASELINE.json configs (the real assets -- Bistro, sci-fi base,
Emerald Square -- are git-ignored downloads in the reference, .gitignore:12-13 / build.bash:34-38, and there
is no network here).  Parameters follow SURVEY.md section 8(d).  Generators only produce inputs (meshes,
transforms, materials, lights, camera); they use the product host mirror for matrix maths.
"""
import math

import numpy as np

f32 = np.float32
LEFT, RIGHT = 0, 1
OPAQUE, CUTOUT, BLEND = 0, 1, 2

_CUBE_POS = [
    (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1),
    (-1, 1, -1), (1, 1, -1), (1, -1, -1), (-1, -1, -1),
    (1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1),
    (-1, -1, 1), (-1, 1, 1), (-1, 1, -1), (-1, -1, -1),
    (1, 1, -1), (-1, 1, -1), (-1, 1, 1), (1, 1, 1),
    (1, -1, 1), (-1, -1, 1), (-1, -1, -1), (1, -1, -1),
]
_CUBE_IDX = [0, 1, 2, 2, 3, 0, 4, 5, 6, 6, 7, 4, 8, 9, 10, 10, 11, 8, 12, 13, 14, 14, 15, 12,
             16, 17, 18, 18, 19, 16, 20, 21, 22, 22, 23, 20]

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
    pos = np.array(_CUBE_POS, dtype=f32) * np.array([sx, sy, sz], dtype=f32)
    nrm = np.repeat(np.array([(0, 0, 1), (0, 0, -1), (1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0)], dtype=f32), 4, 0)
    return pos, np.array(_CUBE_IDX, dtype=np.uint32), nrm


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




def subdivided_box(n, sx=1.0, sy=1.0, sz=1.0):
    """Box whose 6 faces are n x n quads (12 n^2 triangles), wound like the reference cube
    (examples/src/cube/mod.rs:39-46): front-facing from outside for a left-handed renderer."""
    faces = [((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (-1, 0, 0), (0, 1, 0)), ((1, 0, 0), (0, 0, -1), (0, 1, 0)),
             ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (1, 0, 0), (0, 0, -1)), ((0, -1, 0), (1, 0, 0), (0, 0, 1))]
    pos, nrm, idx = [], [], []
    ts = np.linspace(-1.0, 1.0, n + 1)
    for nvec, u, v in faces:
        nvec, u, v = (np.array(a, dtype=np.float64) for a in (nvec, u, v))
        base = len(pos)
        for b in ts:
            for a in ts:
                pos.append(nvec + a * u + b * v)
                nrm.append(nvec)
        # orientation: (u x v) . n > 0 means a,b,c order (a -> a+1 -> a+n+1) is CCW seen from outside
        ccw = float(np.dot(np.cross(u, v), nvec)) > 0
        for j in range(n):
            for i in range(n):
                a0 = base + j * (n + 1) + i
                b0, c0, d0 = a0 + 1, a0 + n + 1, a0 + n + 2
                tri = [(a0, b0, c0), (b0, d0, c0)] if ccw else [(a0, c0, b0), (b0, c0, d0)]
                for t in tri:
                    idx += [t[0], t[1], t[2]]
    pos = np.array(pos, dtype=f32) * np.array([sx, sy, sz], dtype=f32)
    return pos, np.array(idx, dtype=np.uint32), np.array(nrm, dtype=f32)


def _flip(idx):
    return np.ascontiguousarray(idx.reshape(-1, 3)[:, ::-1].reshape(-1))


def bistro_like(r, hm, mk, n_objects=3000, target_tris=2_800_000, n_materials=130, seed=0xB157, shadow_res=2048,
                n_lights=4):
    """BASELINE.json configs[2] stand-in (SURVEY.md section 8d cfg 3): street canyon with real occlusion, ~3 000 objects,
    ~2.8 M triangles (log-normal per object), 130 untextured PBR materials (roughness U[0.2,0.9], metallic in {0,1}
    p=0.2), 4 directional lights with 2048^2 shadow views, distance 100, ambient 0.1 (applied by the caller),
    Bistro test camera of examples/src/scene_viewer/mod.rs:727-751.  Right-handed like scene_viewer (:435).
    Returns dict(objects, triangles, camera=(view, projection))."""
    rng = Pcg32(seed)
    rh = r.handedness == RIGHT
    fix = (lambda i: _flip(i)) if rh else (lambda i: i)  # generators emit LH-front-facing winding

    # ---- mesh library: triangle counts from 12 to 20 480
    lib = []
    for n in (1, 2, 4, 8, 16, 32):
        p, i, nr = subdivided_box(n)
        lib.append(("box", 12 * n * n, r.add_mesh(p, fix(i), normals=nr)))
    for sub in (1, 2, 3, 4, 5):
        p, i, nr = icosphere(sub)
        lib.append(("sphere", 20 * 4 ** sub, r.add_mesh(p, fix(i), normals=nr)))
    boxes = [m for m in lib if m[0] == "box"]
    spheres = [m for m in lib if m[0] == "sphere"]

    mats = []
    for _ in range(n_materials):
        col = (rng.uniform(0.15, 0.95), rng.uniform(0.15, 0.95), rng.uniform(0.15, 0.95), 1.0)
        mats.append(r.add_material(mk(albedo=col, albedo_mode="value", roughness=rng.uniform(0.2, 0.9),
                                      metallic=1.0 if rng.uniform() < 0.2 else 0.0), OPAQUE))

    # ---- camera (scene_viewer/mod.rs:739-741, view = euler XYZ(-pitch,-yaw,0) * T(-loc), :640-641)
    loc = (-17.174278, 3.715882, -4.631997)
    pitch, yaw = 0.04430086, 4.6065736
    view = hm.mat4_mul(hm.from_euler_xyz(-pitch, -yaw, 0.0), hm.translation((-loc[0], -loc[1], -loc[2])))
    inv = hm.mat4_inverse(view)
    fwd = -inv[8:11] if rh else inv[8:11]          # camera looks down -Z (RH) / +Z (LH) in view space
    fx, fz = float(fwd[0]), float(fwd[2])
    nrm_ = math.hypot(fx, fz)
    fx, fz = fx / nrm_, fz / nrm_
    sx, sz = -fz, fx                                # street-right direction (XZ plane)
    yaw_street = math.atan2(fx, fz)                 # rotation about Y taking +Z onto the street direction

    def place(along, side, up):
        return (loc[0] + fx * along + sx * side, up, loc[2] + fz * along + sz * side)

    def xf(pos, scale, rot_y=0.0):
        return hm.mat4_mul(hm.mat4_mul(hm.translation(pos), hm.rotation_y(yaw_street + rot_y)), hm.scale(scale))

    objs = []  # (mesh entry, transform)
    street_half, length = 6.0, 160.0
    # ground: tiles of a finely subdivided thin box (reaches behind the camera: near-plane crossing triangles)
    tile = 8.0
    for a in np.arange(-16.0, length, 2 * tile):
        for s in np.arange(-5 * tile, 5 * tile + 0.1, 2 * tile):
            objs.append((boxes[4], xf(place(a + tile, s, -0.1), (tile, 0.1, tile))))
    # three rows of buildings each side: the back rows are occluded by the front row
    for row, dist in enumerate((street_half + 6.0, street_half + 22.0, street_half + 38.0)):
        for side in (-1.0, 1.0):
            a = -10.0
            while a < length:
                w = rng.uniform(5.0, 9.0)
                h = rng.uniform(8.0, 22.0) + 4.0 * row
                mesh = boxes[5] if row == 0 else boxes[3]
                objs.append((mesh, xf(place(a + w, side * dist, h), (w, h, 6.0))))
                a += 2 * w + rng.uniform(0.3, 1.5)
    # the far end of the street is closed by a block (so the canyon has a back wall)
    objs.append((boxes[5], xf(place(length + 8.0, 0.0, 14.0), (40.0, 14.0, 6.0))))
    n_fixed = len(objs)

    # ---- props: log-normal triangle counts, scattered on the street, behind facades and inside the back rows
    remaining = max(n_objects - n_fixed, 0)
    fixed_tris = sum(m[1] for m, _ in objs)
    budget = max(target_tris - fixed_tris, 0)
    mu = math.log(max(budget / max(remaining, 1), 12.0)) - 0.5 * 1.2 ** 2
    for _ in range(remaining):
        want = math.exp(mu + 1.2 * _gauss(rng))
        cands = spheres + boxes
        mesh = min(cands, key=lambda m: abs(math.log(m[1]) - math.log(max(want, 12.0))))
        zone = rng.uniform()
        along = rng.uniform(-8.0, length)
        if zone < 0.45:      # on the street / pavements
            side = rng.uniform(-street_half - 1.0, street_half + 1.0)
            s = rng.uniform(0.15, 0.9)
            up = s
        elif zone < 0.6:     # lamps / signs hanging above the street
            side = rng.uniform(-street_half, street_half)
            s = rng.uniform(0.1, 0.5)
            up = rng.uniform(3.0, 9.0)
        else:                # between / behind building rows: mostly occluded
            side = (1.0 if rng.uniform() < 0.5 else -1.0) * rng.uniform(street_half + 13.0, street_half + 45.0)
            s = rng.uniform(0.3, 2.0)
            up = rng.uniform(0.3, 12.0)
        objs.append((mesh, xf(place(along, side, up), (s, s, s), rng.uniform(0.0, 2 * math.pi))))

    total = 0
    for mesh, m in objs:
        r.add_object(mesh[2], mats[rng.randint(len(mats))], m)
        total += mesh[1]

    # ---- lights (scene_viewer/mod.rs:736-737 + 3 rotated copies), shadow distance 100, resolution 2048
    d0 = np.array([1.0, -5.0, -1.0])
    for k in range(n_lights):
        ang = 0.5 * math.pi * k
        d = (d0[0] * math.cos(ang) + d0[2] * math.sin(ang), d0[1], -d0[0] * math.sin(ang) + d0[2] * math.cos(ang))
        r.add_directional_light(color=(1.0, 0.96, 0.9) if k == 0 else (0.6, 0.7, 1.0), intensity=15.0 if k == 0 else 4.0,
                                direction=d, distance=100.0, resolution=shadow_res)
    projection = ("perspective", 60.0, 0.1)
    r.set_camera_data(view, projection)
    return dict(objects=len(objs), triangles=total, camera=(view, projection))


def _gauss(rng):
    u1 = max(rng.uniform(), 1e-12)
    u2 = rng.uniform()
    return math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)
