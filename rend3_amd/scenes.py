"""Synthetic input generators for the named BASELINE.json configs (the real assets -- Bistro, sci-fi base,
Emerald Square -- are git-ignored downloads in the reference, .gitignore:12-13 / build.bash:34-38, and there
is no network here).  Parameters follow SURVEY.md section 8(d).  Generators only produce inputs (meshes,
transforms, materials, lights, camera); they use the product host mirror for matrix maths.
"""
import math
import os

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




def subdivided_box(n, sx=1.0, sy=1.0, sz=1.0, share=True):
    """Box whose 6 faces are n x n quads (12 n^2 triangles), wound like the reference cube
    (examples/src/cube/mod.rs:39-46): front-facing from outside for a left-handed renderer.
    share=False: every quad owns its four vertices (24 n^2 vertices, two per triangle -- the vertex : triangle ratio of
    meshes with per-face attributes / UV seams) instead of sharing the (n + 1)^2 grid vertices of its face."""
    faces = [((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (-1, 0, 0), (0, 1, 0)), ((1, 0, 0), (0, 0, -1), (0, 1, 0)),
             ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (1, 0, 0), (0, 0, -1)), ((0, -1, 0), (1, 0, 0), (0, 0, 1))]
    pos, nrm, idx = [], [], []
    ts = np.linspace(-1.0, 1.0, n + 1)
    for nvec, u, v in faces:
        nvec, u, v = (np.array(a, dtype=np.float64) for a in (nvec, u, v))
        # orientation: (u x v) . n > 0 means a,b,c order (a -> a+1 -> a+n+1) is CCW seen from outside
        ccw = float(np.dot(np.cross(u, v), nvec)) > 0
        grid = nvec[None, None, :] + ts[None, :, None] * u[None, None, :] + ts[:, None, None] * v[None, None, :]  # [b][a]
        base = sum(len(p_) for p_ in pos)
        if share:
            pos.append(grid.reshape(-1, 3))
            j, i = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
            a0 = (base + j * (n + 1) + i).reshape(-1)
            b0, c0, d0 = a0 + 1, a0 + n + 1, a0 + n + 2
        else:
            quads = np.stack([grid[:-1, :-1], grid[:-1, 1:], grid[1:, :-1], grid[1:, 1:]], axis=2)  # [j][i][corner a,b,c,d]
            pos.append(quads.reshape(-1, 3))
            a0 = base + 4 * np.arange(n * n)
            b0, c0, d0 = a0 + 1, a0 + 2, a0 + 3
        tri = np.stack([a0, b0, c0, b0, d0, c0], axis=1) if ccw else np.stack([a0, c0, b0, b0, c0, d0], axis=1)
        idx.append(tri.reshape(-1))
        nrm.append(np.tile(nvec, (len(pos[-1]), 1)))
    pos = np.concatenate(pos).astype(f32) * np.array([sx, sy, sz], dtype=f32)
    return pos, np.concatenate(idx).astype(np.uint32), np.concatenate(nrm).astype(f32)


def _flip(idx):
    return np.ascontiguousarray(idx.reshape(-1, 3)[:, ::-1].reshape(-1))


def planar_uv_tangent(positions, normals, tiles):
    """Box-projected texture coordinates (the two axes other than the dominant normal axis, `tiles` repeats per unit)
    and a unit tangent along the first of those axes, re-orthogonalised against the normal."""
    p = np.asarray(positions, dtype=f32)
    n = np.asarray(normals, dtype=f32)
    dom = np.argmax(np.abs(n), axis=1)
    ua = np.where(dom == 0, 2, 0)
    va = np.where(dom == 1, 2, 1)
    rows = np.arange(len(p))
    uv = np.stack([p[rows, ua], p[rows, va]], axis=1).astype(f32) * f32(tiles)
    t = np.zeros_like(p)
    t[rows, ua] = 1.0
    t = t - n * (t * n).sum(axis=1, keepdims=True)
    t /= np.maximum(np.linalg.norm(t, axis=1, keepdims=True), f32(1e-20))
    return uv.astype(f32), t.astype(f32)


def procedural_textures(n_albedo=16, n_normal=8, n_orm=8, size=1024, seed=0x7E57):
    """RGBA8 stand-ins for a scanned material set: tinted brick / plaster albedo (sRGB), tangent-space bump maps and
    packed AO / roughness / metallic maps (linear).  Deterministic (numpy Generator)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    out = {"albedo": [], "normal": [], "orm": []}

    def smooth_noise(cells):
        g = rng.random((cells + 1, cells + 1)).astype(np.float32)
        fx, fy = xx * cells / size, yy * cells / size
        ix, iy = fx.astype(int), fy.astype(int)
        tx, ty = fx - ix, fy - iy
        a = g[iy, ix] * (1 - tx) + g[iy, ix + 1] * tx
        b = g[iy + 1, ix] * (1 - tx) + g[iy + 1, ix + 1] * tx
        return a * (1 - ty) + b * ty

    for k in range(n_albedo):
        base = rng.uniform(0.35, 0.95, 3)
        bricks_x, bricks_y = 4 << (k % 3), 8 << (k % 3)
        row = (yy * bricks_y / size).astype(int)
        mortar = (((xx * bricks_x / size + 0.5 * (row % 2)) % 1.0) < 0.06) | (((yy * bricks_y / size) % 1.0) < 0.1)
        tone = 0.75 + 0.25 * smooth_noise(16) + 0.1 * rng.random((size, size)).astype(np.float32)
        img = np.empty((size, size, 4), dtype=np.uint8)
        for c in range(3):
            img[..., c] = np.clip(np.where(mortar, 0.55, base[c]) * tone * 255.0, 0, 255).astype(np.uint8)
        img[..., 3] = 255
        out["albedo"].append(img)
    for k in range(n_normal):
        hgt = smooth_noise(32 << (k % 2)) + 0.3 * smooth_noise(128)
        dx = np.roll(hgt, -1, axis=1) - np.roll(hgt, 1, axis=1)
        dy = np.roll(hgt, -1, axis=0) - np.roll(hgt, 1, axis=0)
        nx, ny, nz = -dx * 6.0, -dy * 6.0, np.ones_like(dx)
        ln = np.sqrt(nx * nx + ny * ny + nz * nz)
        img = np.empty((size, size, 4), dtype=np.uint8)
        img[..., 0] = ((nx / ln) * 0.5 + 0.5) * 255.0
        img[..., 1] = ((ny / ln) * 0.5 + 0.5) * 255.0
        img[..., 2] = ((nz / ln) * 0.5 + 0.5) * 255.0
        img[..., 3] = 255
        out["normal"].append(img)
    for k in range(n_orm):
        img = np.empty((size // 2, size // 2, 4), dtype=np.uint8)
        sub = (slice(None, None, 2), slice(None, None, 2))
        img[..., 0] = (0.7 + 0.3 * smooth_noise(8)[sub]) * 255.0
        img[..., 1] = np.clip(0.35 + 0.6 * smooth_noise(24)[sub], 0, 1) * 255.0
        img[..., 2] = np.where(smooth_noise(6)[sub] > 0.8, 255, 0) if k % 4 == 0 else 0
        img[..., 3] = 255
        out["orm"].append(img)
    return out


_V2_TEXTURES = {}  # tex_size -> the BC7 chains of bistro_like's v2 texture sets
_BC7_W4 = np.array([0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64], dtype=np.float32)


def bc7_encode(rgba8):
    """(H, W, 4) u8, H and W multiples of 4 -> BC7 blocks (bytes), every block in MODE 6 (one subset, 7-bit RGBA endpoints + a
    p-bit each, 4-bit indices): endpoints = the block's per-channel bounding box, indices by projection on its diagonal.  A
    plain encoder for synthetic assets -- the product and the oracle DECODE the blocks (texture_decode.hip / oracle/bcn.c, both
    pinned on Pillow's decoder), so whatever it writes reads back the same on both sides.  Vectorised over the image's blocks."""
    h, w = rgba8.shape[:2]
    assert h % 4 == 0 and w % 4 == 0 and rgba8.shape[2] == 4
    px = rgba8.reshape(h // 4, 4, w // 4, 4, 4).transpose(0, 2, 1, 3, 4).reshape(-1, 16, 4).astype(np.float32)  # blocks x pixels x channels
    lo8, hi8 = px.min(axis=1), px.max(axis=1)
    # the bounding box's diagonal that follows the data: a channel that falls while the widest channel rises swaps its ends
    dev = px - px.mean(axis=1, keepdims=True)
    major = np.take_along_axis(dev, (hi8 - lo8).argmax(axis=1)[:, None, None].repeat(16, 1), axis=2)
    anti = (dev * major).sum(axis=1) < 0.0
    lo8, hi8 = np.where(anti, hi8, lo8), np.where(anti, lo8, hi8)

    def quant(e):  # 8-bit value -> 7 bits + a p-bit shared by the endpoint's four channels
        e = e.astype(np.uint32)
        p = ((e & 1).sum(axis=1) >= 2).astype(np.uint32)
        q = np.clip((e.astype(np.int32) - p[:, None].astype(np.int32) + 1) >> 1, 0, 127).astype(np.uint32)
        return q, p, ((q << 1) | p[:, None]).astype(np.float32)

    q0, p0, e0 = quant(lo8)
    q1, p1, e1 = quant(hi8)
    d = e1 - e0
    den = (d * d).sum(axis=1)
    t = ((px - e0[:, None, :]) * d[:, None, :]).sum(axis=2) / np.maximum(den, 1.0)[:, None]
    idx = np.abs(np.clip(t, 0.0, 1.0)[:, :, None] * 64.0 - _BC7_W4[None, None, :]).argmin(axis=2).astype(np.uint64)
    swap = idx[:, 0] >= 8  # the anchor index (pixel 0) stores three bits: its top bit must be clear
    idx[swap] = 15 - idx[swap]
    q0s, q1s = np.where(swap[:, None], q1, q0).astype(np.uint64), np.where(swap[:, None], q0, q1).astype(np.uint64)
    p0s, p1s = np.where(swap, p1, p0).astype(np.uint64), np.where(swap, p0, p1).astype(np.uint64)
    lo = np.full(len(px), 1 << 6, dtype=np.uint64)  # mode 6: six zero bits, then a one
    for c in range(4):
        lo |= (q0s[:, c] << np.uint64(7 + 14 * c)) | (q1s[:, c] << np.uint64(14 + 14 * c))
    lo |= p0s << np.uint64(63)
    hi = p1s | (idx[:, 0] << np.uint64(1))
    for k in range(1, 16):
        hi |= idx[:, k] << np.uint64(4 * k)
    return np.stack([lo, hi], axis=1).astype("<u8").tobytes()


def bc7_mip_chain(rgba8):
    """BC7 levels of an image (power-of-two extents >= 4), largest first, down to 4 x 4: box-filtered levels, each encoded."""
    levels, img = [], rgba8
    while True:
        levels.append(bc7_encode(img))
        if min(img.shape[:2]) <= 4:
            return levels
        img = ((img[0::2, 0::2].astype(np.uint16) + img[1::2, 0::2] + img[0::2, 1::2] + img[1::2, 1::2] + 2) >> 2).astype(np.uint8)


def foliage_cards(n_cards, rng_np):
    """A clump of `n_cards` crossed, double-sided quads inside the unit cube (two triangles per side: 4 per card), with texture
    coordinates over the whole leaf atlas: the alpha-tested foliage of a scanned city asset.  Returns positions, indices, normals,
    uv, tangents (LH-front-facing winding like the other generators; both windings are present)."""
    c = rng_np.uniform(-0.8, 0.8, (n_cards, 3)).astype(f32)
    ang = rng_np.uniform(0.0, math.pi, n_cards).astype(f32)
    half = rng_np.uniform(0.12, 0.3, n_cards).astype(f32)
    ux = np.stack([np.cos(ang), np.zeros(n_cards, f32), np.sin(ang)], axis=1) * half[:, None]
    uy = np.stack([np.zeros(n_cards, f32), np.ones(n_cards, f32), np.zeros(n_cards, f32)], axis=1) * half[:, None]
    corners = np.stack([c - ux - uy, c + ux - uy, c + ux + uy, c - ux + uy], axis=1)  # (n, 4, 3)
    nrm = np.cross(ux, uy)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    tan = ux / half[:, None]
    # front and back copies own their vertices (opposite normals)
    pos = np.concatenate([corners, corners], axis=1).reshape(-1, 3).astype(f32)
    normals = np.concatenate([np.repeat(nrm[:, None], 4, 1), np.repeat(-nrm[:, None], 4, 1)], axis=1).reshape(-1, 3).astype(f32)
    tangents = np.concatenate([np.repeat(tan[:, None], 4, 1), np.repeat(-tan[:, None], 4, 1)], axis=1).reshape(-1, 3).astype(f32)
    uv1 = np.array([[0, 1], [1, 1], [1, 0], [0, 0]], dtype=f32)
    uv = np.tile(np.concatenate([uv1, uv1]), (n_cards, 1)).astype(f32)
    base = (np.arange(n_cards, dtype=np.uint32) * 8)[:, None]
    idx = np.concatenate([base + np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32), base + np.array([4, 6, 5, 4, 7, 6], dtype=np.uint32)], axis=1).reshape(-1)
    return pos, idx.astype(np.uint32), normals, uv, tangents


def leaf_atlas(size, seed):
    """RGBA8 leaf texture: green blades on a transparent ground (alpha 0 / 255 with a soft edge): what an alpha cutout tests."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    a = np.zeros((size, size), dtype=np.float32)
    for _ in range(40):
        cx, cy, ang, ln, wd = rng.uniform(0.1, 0.9), rng.uniform(0.1, 0.9), rng.uniform(0, math.pi), rng.uniform(0.08, 0.22), rng.uniform(0.015, 0.05)
        dx, dy = xx - cx, yy - cy
        u, v = dx * math.cos(ang) + dy * math.sin(ang), -dx * math.sin(ang) + dy * math.cos(ang)
        a = np.maximum(a, np.clip(1.5 - np.sqrt((u / ln) ** 2 + (v / wd) ** 2) * 1.5, 0.0, 1.0))
    img = np.empty((size, size, 4), dtype=np.uint8)
    shade = 0.6 + 0.4 * rng.random((size, size)).astype(np.float32)
    img[..., 0] = 60 * shade
    img[..., 1] = 170 * shade
    img[..., 2] = 50 * shade
    img[..., 3] = np.clip(a * 255.0, 0, 255)
    return img


def bistro_like(r, hm, mk, n_objects=3000, target_tris=2_800_000, n_materials=130, seed=0xB157, shadow_res=2048,
                n_lights=4, textured=False, unique=True, tex_size=1024, v2=False, v2_tex_size=2048):
    """BASELINE.json configs[2] stand-in (SURVEY.md section 8d cfg 3): street canyon with real occlusion, ~3 000 objects,
    ~2.8 M triangles (log-normal per object), 130 PBR materials (roughness U[0.2,0.9], metallic in {0,1} p=0.2),
    4 directional lights with 2048^2 shadow views, distance 100, ambient 0.1 (applied by the caller),
    Bistro test camera of examples/src/scene_viewer/mod.rs:727-751.  Right-handed like scene_viewer (:435).
    unique (default): EVERY object owns its geometry, like the real asset (~2.8 M unique triangles): its mesh is one of
    the 11 shapes below with per-quad vertices for the boxes (two vertices per triangle) and a smooth per-object
    deformation (a sine field of the vertex position, so coincident vertices move together and the surface stays
    watertight); the mesh buffer then holds ~150 MB (textured: position + normal + tangent + uv per vertex) instead of the
    1.5 MB of the instanced variant, and the cull / raster / vertex-stage gathers go to HBM instead of hitting a
    cache-resident mesh library.  unique=False: the 11 meshes are instanced (round-1 workload, labelled as such).
    textured: every material gets a base colour, a normal and a packed AO / roughness / metallic map (tex_size^2 /
    (tex_size/2)^2 RGBA8 with full mip chains, trilinear) from a pool of 32 procedural textures, like the scanned material
    set of the real asset; meshes then carry box-projected texture coordinates and tangents.
    v2 (closer to the Lumberyard asset than the default): the maps are 2048^2 (AO / roughness / metallic 1024^2)
    and arrive BLOCK-COMPRESSED -- BC7 with stored mip chains, decoded at upload like any `Bc7RgbaUnorm[Srgb]` texture of
    rend3-gltf's loader -- in 24 texture sets (a texel pool of ~1.2 GB instead of 145 MB); a fifth of the triangles are
    alpha-tested FOLIAGE: clumps of crossed double-sided cards on the CUTOUT key whose alpha comes from a leaf atlas
    (opaque.wgsl:231-235, depth.wgsl:100-127); and a third of the props are INSTANCES of shared meshes.
    Returns dict(objects, triangles, camera=(view, projection), mesh_bytes, unique_triangles)."""
    rng = Pcg32(seed)
    if v2:
        textured, tex_size = True, v2_tex_size  # (tests pass a smaller size)
    rh = r.handedness == RIGHT
    fix = (lambda i: _flip(i)) if rh else (lambda i: i)  # generators emit LH-front-facing winding

    # ---- mesh library: triangle counts from 12 to 20 480
    lib = []
    templates = []  # unique: (positions, indices, normals, uv, tangents) of every shape; instanced: the mesh handle
    deform_rng = Pcg32(seed ^ 0xD3F0, seq=77)  # its own stream: object placement is identical with and without `unique`
    mesh_bytes = [0]
    unique_tris = [0]

    def upload(p, i, nr, uv, tan):
        mesh_bytes[0] += 4 * (p.size + nr.size + i.size + (uv.size + tan.size if textured else 0))
        unique_tris[0] += len(i) // 3
        if not textured:
            return r.add_mesh(p, i, normals=nr)
        return r.add_mesh(p, i, normals=nr, uv0=uv, tangents=tan)

    def add(kind, ntri, p, i, nr):
        i = fix(i)
        uv, tan = planar_uv_tangent(p, nr, 2.0) if textured else (None, None)
        if unique:
            templates.append((p, i, nr, uv, tan))
            lib.append((kind, ntri, len(templates) - 1))
        else:
            lib.append((kind, ntri, upload(p, i, nr, uv, tan)))

    def instance_mesh(entry):
        """The mesh handle an object of library entry `entry` draws: the shared one, or its own deformed copy."""
        if not unique:
            return entry[2]
        p, i, nr, uv, tan = templates[entry[2]]
        # smooth displacement field d(x) = amp * sin(K x + phase) per axis: a function of position only
        k = np.array([[deform_rng.uniform(1.5, 6.0) for _ in range(3)] for _ in range(3)], dtype=f32)
        ph = np.array([deform_rng.uniform(0.0, 2.0 * math.pi) for _ in range(3)], dtype=f32)
        amp = f32(deform_rng.uniform(0.004, 0.02))
        q = (p + amp * np.sin(p @ k.T + ph)).astype(f32)
        return upload(q, i, nr, uv, tan)

    for n in (1, 2, 4, 8, 16, 32):
        p, i, nr = subdivided_box(n, share=not unique)
        add("box", 12 * n * n, p, i, nr)
    for sub in (1, 2, 3, 4, 5):
        p, i, nr = icosphere(sub)
        add("sphere", 20 * 4 ** sub, p, i, nr)
    boxes = [m for m in lib if m[0] == "box"]
    spheres = [m for m in lib if m[0] == "sphere"]

    tex = None
    if textured and v2:
        from . import containers
        # six distinct images per kind, encoded once; 24 texture sets cycle over them (every set owns its texels in the pool: the
        # working set is that of 24 distinct sets)
        if tex_size not in _V2_TEXTURES:  # (the oracle's and the product's renderer build the same scene: encode once per process,
            # and keep the encoded chains in the temporary directory for the next process of the same session: ~100 s at 2048^2)
            import pickle
            import tempfile
            cache = os.path.join(tempfile.gettempdir(), f"rend3_amd_bistro_v2_textures_{tex_size}.pkl")
            try:
                with open(cache, "rb") as fh:
                    _V2_TEXTURES[tex_size] = pickle.load(fh)
            except (OSError, ValueError, EOFError, pickle.UnpicklingError):
                imgs = procedural_textures(n_albedo=6, n_normal=6, n_orm=6, size=tex_size)
                _V2_TEXTURES[tex_size] = ({k: [bc7_mip_chain(im) for im in v] for k, v in imgs.items()},
                                          [bc7_mip_chain(leaf_atlas(tex_size // 2, 0x1EAF + k)) for k in range(4)])
                try:
                    with open(cache + ".tmp", "wb") as fh:
                        pickle.dump(_V2_TEXTURES[tex_size], fh)
                    os.replace(cache + ".tmp", cache)
                except OSError:
                    pass
        enc, leaf_enc = _V2_TEXTURES[tex_size]
        n_sets = 24
        tex = {"albedo": [r.add_texture_2d_encoded(containers.BC7_SRGB, tex_size, tex_size, enc["albedo"][k % 6]) for k in range(n_sets)],
               "normal": [r.add_texture_2d_encoded(containers.BC7, tex_size, tex_size, enc["normal"][k % 6]) for k in range(n_sets)],
               "orm": [r.add_texture_2d_encoded(containers.BC7, tex_size // 2, tex_size // 2, enc["orm"][k % 6]) for k in range(n_sets)]}
        leaf = [r.add_texture_2d_encoded(containers.BC7_SRGB, tex_size // 2, tex_size // 2, leaf_enc[k]) for k in range(4)]
    elif textured:
        imgs = procedural_textures(size=tex_size)
        tex = {"albedo": [r.add_texture_2d(im, srgb=True, mip_count="maximum", mip_source="generated") for im in imgs["albedo"]],
               "normal": [r.add_texture_2d(im, srgb=False, mip_count="maximum", mip_source="generated") for im in imgs["normal"]],
               "orm": [r.add_texture_2d(im, srgb=False, mip_count="maximum", mip_source="generated") for im in imgs["orm"]]}
    mats = []
    for k in range(n_materials):
        col = (rng.uniform(0.15, 0.95), rng.uniform(0.15, 0.95), rng.uniform(0.15, 0.95), 1.0)
        rough, metal = rng.uniform(0.2, 0.9), 1.0 if rng.uniform() < 0.2 else 0.0
        if textured:
            rec = mk(albedo=col, albedo_mode="texture_value", albedo_texture=tex["albedo"][k % len(tex["albedo"])],
                     roughness=rough, metallic=metal, normal_texture=tex["normal"][k % len(tex["normal"])],
                     aomr=("combined", tex["orm"][k % len(tex["orm"])]))
        else:
            rec = mk(albedo=col, albedo_mode="value", roughness=rough, metallic=metal)
        mats.append(r.add_material(rec, OPAQUE))

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
    shared = {}  # v2: library entry -> the mesh its instanced props share
    for k, (mesh, m) in enumerate(objs):
        if v2 and k >= n_fixed and k % 3 == 0:  # a third of the props are instances of shared meshes
            if id(mesh) not in shared:
                shared[id(mesh)] = instance_mesh(mesh)
            handle = shared[id(mesh)]
        else:
            handle = instance_mesh(mesh)
        r.add_object(handle, mats[rng.randint(len(mats))], m)
        total += mesh[1]
    if v2:
        # ---- foliage: clumps of alpha-tested cards (cutout key) in planters along the pavements and on the facades, until a
        # fifth of the scene's triangles are theirs; 8 clump meshes, instanced
        frng = np.random.default_rng(seed ^ 0xF011A6E)
        leaf_mats = [r.add_material(mk(albedo=(1.0, 1.0, 1.0, 1.0), albedo_mode="texture", albedo_texture=leaf[k % 4], roughness=0.8,
                                       metallic=0.0, cutout=0.5), CUTOUT) for k in range(8)]
        clumps = []
        for k in range(8):
            p, i, nr, uv, tan = foliage_cards(350, frng)
            clumps.append((upload(p, fix(i), nr, uv, tan), len(i) // 3))
        want = int(0.25 * total)  # of the opaque triangles = a fifth of the whole
        placed = 0
        while placed < want:
            k = int(frng.integers(0, 8))
            along = float(frng.uniform(-6.0, length))
            if frng.random() < 0.7:   # trees / planters on the pavements
                side = float((1.0 if frng.random() < 0.5 else -1.0) * frng.uniform(street_half - 1.5, street_half + 2.0))
                s, up = float(frng.uniform(0.8, 2.2)), float(frng.uniform(1.0, 5.0))
            else:                     # creepers on the facades
                side = float((1.0 if frng.random() < 0.5 else -1.0) * (street_half + 5.4))
                s, up = float(frng.uniform(0.6, 1.5)), float(frng.uniform(2.0, 12.0))
            r.add_object(clumps[k][0], leaf_mats[int(frng.integers(0, 8))], xf(place(along, side, up), (s, s, s), float(frng.uniform(0.0, 2 * math.pi))))
            objs.append((None, None))
            placed += clumps[k][1]
        total += placed

    # ---- lights (scene_viewer/mod.rs:736-737 + 3 rotated copies), shadow distance 100, resolution 2048
    d0 = np.array([1.0, -5.0, -1.0])
    for k in range(n_lights):
        ang = 0.5 * math.pi * k
        d = (d0[0] * math.cos(ang) + d0[2] * math.sin(ang), d0[1], -d0[0] * math.sin(ang) + d0[2] * math.cos(ang))
        r.add_directional_light(color=(1.0, 0.96, 0.9) if k == 0 else (0.6, 0.7, 1.0), intensity=15.0 if k == 0 else 4.0,
                                direction=d, distance=100.0, resolution=shadow_res)
    projection = ("perspective", 60.0, 0.1)
    r.set_camera_data(view, projection)
    return dict(objects=len(objs), triangles=total, camera=(view, projection), mesh_bytes=mesh_bytes[0],
                unique_triangles=unique_tris[0])


def _gauss(rng):
    u1 = max(rng.uniform(), 1e-12)
    u2 = rng.uniform()
    return math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2)


def cylinder(seg, rings):
    """Closed cylinder along Y, radius 1, height 2: 2*seg*rings side triangles + 2*(seg-2) cap triangles, LH-front winding."""
    pos, nrm, idx = [], [], []
    for r_ in range(rings + 1):
        y = -1.0 + 2.0 * r_ / rings
        for s_ in range(seg):
            a = 2.0 * math.pi * s_ / seg
            pos.append((math.cos(a), y, math.sin(a)))
            nrm.append((math.cos(a), 0.0, math.sin(a)))
    for r_ in range(rings):
        for s_ in range(seg):
            a0 = r_ * seg + s_
            a1 = r_ * seg + (s_ + 1) % seg
            b0, b1 = a0 + seg, a1 + seg
            idx += [a0, b0, a1, a1, b0, b1]
    for cap, y, ny in ((0, -1.0, -1.0), (1, 1.0, 1.0)):
        base = len(pos)
        for s_ in range(seg):
            a = 2.0 * math.pi * s_ / seg
            pos.append((math.cos(a), y, math.sin(a)))
            nrm.append((0.0, ny, 0.0))
        for s_ in range(1, seg - 1):
            idx += [base, base + s_, base + s_ + 1] if cap == 0 else [base, base + s_ + 1, base + s_]
    pos = np.array(pos, dtype=f32)
    idx = np.array(idx, dtype=np.uint32)
    # orient like the reference cube: cross(e1, e2) points outward
    t = idx.reshape(-1, 3)
    c = np.cross(pos[t[:, 1]] - pos[t[:, 0]], pos[t[:, 2]] - pos[t[:, 0]])
    cen = (pos[t[:, 0]] + pos[t[:, 1]] + pos[t[:, 2]]) / 3.0
    cen[:, 1] *= 0.25  # caps: outward is +-Y, sides: radial
    flip = (c * cen).sum(axis=1) < 0
    t[flip] = t[flip][:, ::-1]
    return pos, t.reshape(-1), np.array(nrm, dtype=f32)


def mesh_library_64(r, rh):
    """64 procedural meshes, 12 ... 5 120 triangles (SURVEY.md section 8d cfg 2: icosphere / box / cylinder)."""
    fix = _flip if rh else (lambda i: i)
    lib = []
    for n in range(1, 21):
        p, i, nr = subdivided_box(n)
        lib.append((12 * n * n, r.add_mesh(p, fix(i), normals=nr)))
    for sub in range(0, 5):
        p, i, nr = icosphere(sub)
        lib.append((20 * 4 ** sub, r.add_mesh(p, fix(i), normals=nr)))
    combos = [(s_, k) for s_ in (6, 8, 12, 16, 24, 32) for k in (1, 2, 4, 8, 16, 32, 64)]
    for s_, k in combos[:39]:
        p, i, nr = cylinder(s_, k)
        lib.append((len(i) // 3, r.add_mesh(p, fix(i), normals=nr)))
    assert len(lib) == 64
    return lib


def random_transforms(rng_np, n, extent, scale_lo, scale_hi):
    """n random TRS matrices (column-major f32[16]): uniform positions in +-extent, uniform random rotations,
    log-uniform scale.  Vectorised numpy; the SAME array is handed to every renderer that must agree."""
    pos = (rng_np.random((n, 3)) * 2.0 - 1.0) * np.asarray(extent)
    q = rng_np.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    s = np.exp(rng_np.uniform(math.log(scale_lo), math.log(scale_hi), size=n))
    M = np.zeros((n, 16), dtype=f32)
    for c in range(3):
        for r_ in range(3):
            M[:, 4 * c + r_] = (R[:, r_, c] * s).astype(f32)
    M[:, 12:15] = pos.astype(f32)
    M[:, 15] = 1.0
    return M


def scifi_like(r, hm, mk, n_objects=20000, seed=0xC0FFEE, n_materials=64, with_lights=False):
    """BASELINE.json configs[1] stand-in (SURVEY.md section 8d cfg 2): 20 000 objects instancing 64 procedural meshes
    (12 ... 5 120 triangles), uniform random positions in a 200 x 40 x 200 m box, random rotations, scale log-uniform
    [0.25, 4]; right-handed, vfov 60, near 0.1, camera at the box centre looking down -Z; cull + compact only."""
    rh = r.handedness == RIGHT
    lib = mesh_library_64(r, rh)
    rng = Pcg32(seed)
    mats = [r.add_material(mk(albedo=(rng.uniform(0.2, 1), rng.uniform(0.2, 1), rng.uniform(0.2, 1), 1.0), albedo_mode="value",
                              roughness=rng.uniform(0.2, 0.9), metallic=1.0 if rng.uniform() < 0.2 else 0.0), OPAQUE)
            for _ in range(n_materials)]
    rng_np = np.random.Generator(np.random.PCG64(seed))
    xf = random_transforms(rng_np, n_objects, (100.0, 20.0, 100.0), 0.25, 4.0)
    mesh_pick = rng_np.integers(0, 64, size=n_objects)
    mat_pick = rng_np.integers(0, len(mats), size=n_objects)
    r.add_objects_bulk([lib[k][1] for k in mesh_pick], [mats[k] for k in mat_pick], xf)
    if with_lights:
        r.add_directional_light(color=(1, 1, 1), intensity=3.0, direction=(0.3, -1.0, 0.2), distance=150.0, resolution=2048)
    view = hm.identity()  # camera at the origin (box centre), looking down -Z (RH) / +Z (LH)
    projection = ("perspective", 60.0, 0.1)
    r.set_camera_data(view, projection)
    return dict(objects=n_objects, triangles=int(sum(lib[k][0] for k in mesh_pick)), camera=(view, projection))


def emerald_like(r, hm, mk, n_objects=1 << 20, seed=0xE5A0, n_materials=130, n_lights=0):
    """BASELINE.json configs[3] stand-in (SURVEY.md section 8d cfg 4): 1 048 576 small objects (mean ~60 triangles) spread
    over a 2 x 2 km district; used for the objects/s figure and for object-range sharding across GPUs."""
    rh = r.handedness == RIGHT
    fix = _flip if rh else (lambda i: i)
    lib = []
    for n in (1, 2, 3):
        p, i, nr = subdivided_box(n)
        lib.append((12 * n * n, r.add_mesh(p, fix(i), normals=nr)))
    for sub in (0, 1, 2):
        p, i, nr = icosphere(sub)
        lib.append((20 * 4 ** sub, r.add_mesh(p, fix(i), normals=nr)))
    rng = Pcg32(seed)
    mats = [r.add_material(mk(albedo=(rng.uniform(0.2, 1), rng.uniform(0.2, 1), rng.uniform(0.2, 1), 1.0), albedo_mode="value",
                              roughness=rng.uniform(0.2, 0.9), metallic=1.0 if rng.uniform() < 0.2 else 0.0), OPAQUE)
            for _ in range(n_materials)]
    rng_np = np.random.Generator(np.random.PCG64(seed))
    xf = random_transforms(rng_np, n_objects, (1000.0, 30.0, 1000.0), 0.5, 6.0)
    # triangle counts 12,48,108,20,80,320 with weights giving a mean near 60
    mesh_pick = rng_np.choice(6, size=n_objects, p=[0.35, 0.2, 0.1, 0.2, 0.1, 0.05])
    mat_pick = rng_np.integers(0, len(mats), size=n_objects)
    r.add_objects_bulk([lib[k][1] for k in mesh_pick], [mats[k] for k in mat_pick], xf)
    for k in range(n_lights):
        ang = 0.5 * math.pi * k
        r.add_directional_light(color=(1, 1, 1), intensity=4.0, direction=(math.cos(ang), -3.0, math.sin(ang)), distance=400.0,
                                resolution=2048)
    view = hm.mat4_mul(hm.rotation_x(0.25), hm.translation((0.0, -40.0, 0.0)))
    projection = ("perspective", 60.0, 0.1)
    r.set_camera_data(view, projection)
    return dict(objects=n_objects, triangles=int(sum(lib[k][0] for k in mesh_pick)), camera=(view, projection))


def skinned_cylinder(joints, seg=16, rings=9):
    """Cylinder along Y (radius 0.3, y in [0, 2]) rigged to `joints` bones spaced along the axis -- the shape of
    examples/src/skinning/RiggedSimple.glb (160 vertices, 2 joints) with a variable joint count.  Every vertex blends
    its two nearest bones; ring 0 adds a third and fourth influence, the last ring has a single weight of 1
    (zero weights exercise skinning.wgsl:69).  Returns positions, indices, normals, tangents, joint_indices, weights."""
    pos, idx, nrm = cylinder(seg, rings)
    pos = pos.copy()
    pos[:, 0] *= 0.3; pos[:, 2] *= 0.3; pos[:, 1] = pos[:, 1] + 1.0
    tang = np.stack([-nrm[:, 2], np.zeros(len(nrm), dtype=f32), nrm[:, 0]], axis=1).astype(f32)
    tang[np.abs(nrm[:, 1]) > 0.5] = (1.0, 0.0, 0.0)
    ji = np.zeros((len(pos), 4), dtype=np.uint16)
    jw = np.zeros((len(pos), 4), dtype=f32)
    for v in range(len(pos)):
        t = float(pos[v, 1]) / 2.0 * (joints - 1)
        a = min(int(math.floor(t)), joints - 1)
        b = min(a + 1, joints - 1)
        fb = f32(t - a)
        if pos[v, 1] >= 2.0 - 1e-6:
            ji[v] = (joints - 1, 0, 0, 0); jw[v] = (1.0, 0.0, 0.0, 0.0)
        elif pos[v, 1] <= 1e-6 and joints >= 4:
            ji[v] = (0, 1, 2, 3); jw[v] = (0.5, 0.25, 0.125, 0.125)
        else:
            ji[v] = (a, b, 0, 0); jw[v] = (f32(1.0) - fb, fb, 0.0, 0.0)
    return pos, idx, nrm, tang, ji, jw
