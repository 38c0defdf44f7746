"""
oracle/host.py -- numpy restatement of the reference's HOST-side math for the hot path.

TEST INFRASTRUCTURE ONLY (see oracle/r3o.c header).  The product's own host layer is
rend3_amd/csrc/host.cpp; tests/test_host_parity.py checks the two agree bit-for-bit.

Follows (reference file:line):
  glam 0.25 conventions            SURVEY.md App. E (glam is an un-vendored dependency,
                                   rend3/Cargo.toml:43 `glam = "0.25.0"`; formulas restated from
                                   its documented column-major conventions, validated against the
                                   goldens in tests/test_oracle_goldens.py)
  CameraState                      rend3/src/managers/camera.rs:23-114
  Frustum::from_matrix             rend3/src/util/frustum.rs:96-145  (done in C: r3o_frustum_from_matrix)
  BoundingSphere                   rend3/src/util/frustum.rs:15-56
  shadow_camera                    rend3/src/managers/directional/shadow_camera.rs:6-33
  allocate_shadow_atlas            rend3/src/managers/directional/shadow_alloc.rs:59-136
  DirectionalLightManager::evaluate rend3/src/managers/directional.rs:99-157
  FrameUniforms::new               rend3-routine/src/uniforms.rs:28-48
  PerCameraUniform header          rend3-routine/src/culling/culler.rs:477-502
  calculate_normals_for_buffers    rend3-types/src/lib.rs:662-704

Matrices are numpy float32 arrays of 16 elements, column-major (m[4*c + r]).
All scalar arithmetic is done on np.float32 scalars in a fixed order; sin/cos are evaluated in
float64 and rounded once to float32.
"""
import math

import numpy as np

f32 = np.float32
LEFT, RIGHT = 0, 1  # rend3_types::Handedness


def _f(x):
    return np.float32(x)


def identity():
    m = np.zeros(16, dtype=f32)
    m[0] = m[5] = m[10] = m[15] = 1.0
    return m


def mat4_mul(a, b):
    """a*b, column by column: ((a0*x + a1*y) + a2*z) + a3*w."""
    a = np.asarray(a, dtype=f32)
    b = np.asarray(b, dtype=f32)
    out = np.zeros(16, dtype=f32)
    for c in range(4):
        x, y, z, w = b[4 * c : 4 * c + 4]
        out[4 * c : 4 * c + 4] = ((a[0:4] * x + a[4:8] * y) + a[8:12] * z) + a[12:16] * w
    return out


def mat4_mul_vec4(m, v):
    m = np.asarray(m, dtype=f32)
    v = np.asarray(v, dtype=f32)
    return ((m[0:4] * v[0] + m[4:8] * v[1]) + m[8:12] * v[2]) + m[12:16] * v[3]


def transform_point3(m, p):
    """Mat4::transform_point3: m * (p, 1), no perspective divide."""
    return mat4_mul_vec4(m, np.array([p[0], p[1], p[2], 1.0], dtype=f32))[:3]


def mat4_inverse(m):
    """General 4x4 inverse by cofactors (adjugate / det), fixed f32 order."""
    m = np.asarray(m, dtype=f32)
    a = lambda c, r: m[4 * c + r]
    m00, m01, m02, m03 = a(0, 0), a(0, 1), a(0, 2), a(0, 3)
    m10, m11, m12, m13 = a(1, 0), a(1, 1), a(1, 2), a(1, 3)
    m20, m21, m22, m23 = a(2, 0), a(2, 1), a(2, 2), a(2, 3)
    m30, m31, m32, m33 = a(3, 0), a(3, 1), a(3, 2), a(3, 3)
    c00 = m22 * m33 - m32 * m23
    c02 = m12 * m33 - m32 * m13
    c03 = m12 * m23 - m22 * m13
    c04 = m21 * m33 - m31 * m23
    c06 = m11 * m33 - m31 * m13
    c07 = m11 * m23 - m21 * m13
    c08 = m21 * m32 - m31 * m22
    c10 = m11 * m32 - m31 * m12
    c11 = m11 * m22 - m21 * m12
    c12 = m20 * m33 - m30 * m23
    c14 = m10 * m33 - m30 * m13
    c15 = m10 * m23 - m20 * m13
    c16 = m20 * m32 - m30 * m22
    c18 = m10 * m32 - m30 * m12
    c19 = m10 * m22 - m20 * m12
    c20 = m20 * m31 - m30 * m21
    c22 = m10 * m31 - m30 * m11
    c23 = m10 * m21 - m20 * m11
    i00 = (m11 * c00 - m12 * c04) + m13 * c08
    i01 = -((m01 * c00 - m02 * c04) + m03 * c08)
    i02 = (m01 * c02 - m02 * c06) + m03 * c10
    i03 = -((m01 * c03 - m02 * c07) + m03 * c11)
    i10 = -((m10 * c00 - m12 * c12) + m13 * c16)
    i11 = (m00 * c00 - m02 * c12) + m03 * c16
    i12 = -((m00 * c02 - m02 * c14) + m03 * c18)
    i13 = (m00 * c03 - m02 * c15) + m03 * c19
    i20 = (m10 * c04 - m11 * c12) + m13 * c20
    i21 = -((m00 * c04 - m01 * c12) + m03 * c20)
    i22 = (m00 * c06 - m01 * c14) + m03 * c22
    i23 = -((m00 * c07 - m01 * c15) + m03 * c23)
    i30 = -((m10 * c08 - m11 * c16) + m12 * c20)
    i31 = (m00 * c08 - m01 * c16) + m02 * c20
    i32 = -((m00 * c10 - m01 * c18) + m02 * c22)
    i33 = (m00 * c11 - m01 * c19) + m02 * c23
    det = ((m00 * i00 + m01 * i10) + m02 * i20) + m03 * i30
    rdet = _f(1.0) / det
    out = np.array(
        [i00, i01, i02, i03, i10, i11, i12, i13, i20, i21, i22, i23, i30, i31, i32, i33], dtype=f32
    )
    return out * rdet


def _dot3(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f32)


def _normalize(v):
    v = np.asarray(v, dtype=f32)
    r = _f(1.0) / np.sqrt(_dot3(v, v))
    return v * r


def _look_to(eye, fwd, up, rh):
    """glam look_to_lh / look_to_rh (look_at_* = look_to_*(eye, center - eye, up))."""
    eye = np.asarray(eye, dtype=f32)
    f = _normalize(np.asarray(fwd, dtype=f32))
    up = np.asarray(up, dtype=f32)
    if rh:
        s = _normalize(_cross(f, up))
        u = _cross(s, f)
        m = np.array(
            [s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0, -_dot3(eye, s), -_dot3(eye, u), _dot3(eye, f), 1],
            dtype=f32,
        )
    else:
        s = _normalize(_cross(up, f))
        u = _cross(f, s)
        m = np.array(
            [s[0], u[0], f[0], 0, s[1], u[1], f[1], 0, s[2], u[2], f[2], 0, -_dot3(eye, s), -_dot3(eye, u), -_dot3(eye, f), 1],
            dtype=f32,
        )
    return m


def look_at_lh(eye, center, up):
    eye = np.asarray(eye, dtype=f32)
    return _look_to(eye, np.asarray(center, dtype=f32) - eye, up, False)


def look_at_rh(eye, center, up):
    eye = np.asarray(eye, dtype=f32)
    return _look_to(eye, np.asarray(center, dtype=f32) - eye, up, True)


def orthographic_lh(l, r, b, t, n, fa):
    l, r, b, t, n, fa = map(_f, (l, r, b, t, n, fa))
    rw = _f(1.0) / (r - l)
    rh = _f(1.0) / (t - b)
    rd = _f(1.0) / (fa - n)
    m = np.zeros(16, dtype=f32)
    m[0] = rw + rw
    m[5] = rh + rh
    m[10] = rd
    m[12] = -(l + r) * rw
    m[13] = -(t + b) * rh
    m[14] = -rd * n
    m[15] = 1.0
    return m


def orthographic_rh(l, r, b, t, n, fa):
    l, r, b, t, n, fa = map(_f, (l, r, b, t, n, fa))
    rw = _f(1.0) / (r - l)
    rh = _f(1.0) / (t - b)
    rd = _f(1.0) / (n - fa)
    m = np.zeros(16, dtype=f32)
    m[0] = rw + rw
    m[5] = rh + rh
    m[10] = rd
    m[12] = -(l + r) * rw
    m[13] = -(t + b) * rh
    m[14] = rd * n
    m[15] = 1.0
    return m


def _sincos32(x):
    x = float(np.float32(x))
    return np.float32(math.sin(x)), np.float32(math.cos(x))


def perspective_infinite_reverse(vfov_deg, aspect, near, rh):
    fov = _f(vfov_deg) * _f(0.017453292519943295)  # f32::to_radians
    s, c = _sincos32(_f(0.5) * fov)
    h = c / s
    w = h / _f(aspect)
    m = np.zeros(16, dtype=f32)
    m[0] = w
    m[5] = h
    m[11] = -1.0 if rh else 1.0
    m[14] = _f(near)
    return m


def rotation_x(a):
    s, c = _sincos32(a)
    m = identity()
    m[5], m[6], m[9], m[10] = c, s, -s, c
    return m


def rotation_y(a):
    s, c = _sincos32(a)
    m = identity()
    m[0], m[2], m[8], m[10] = c, -s, s, c
    return m


def rotation_z(a):
    s, c = _sincos32(a)
    m = identity()
    m[0], m[1], m[4], m[5] = c, s, -s, c
    return m


def translation(t):
    m = identity()
    m[12:15] = np.asarray(t, dtype=f32)
    return m


def scale(s):
    m = identity()
    m[0], m[5], m[10] = _f(s[0]), _f(s[1]), _f(s[2])
    return m


def from_euler_xyz(a, b, c):
    """glam Mat4::from_euler(EulerRot::XYZ, a, b, c) == Rx(a)*Ry(b)*Rz(c) (validated on the cube golden)."""
    return mat4_mul(mat4_mul(rotation_x(a), rotation_y(b)), rotation_z(c))


# ---------------------------------------------------------------------------- camera
class CameraState:
    """rend3/src/managers/camera.rs:11-114."""

    def __init__(self, view, projection, handedness, aspect_ratio=None):
        # projection: ("perspective", vfov_deg, near) | ("orthographic", (sx,sy,sz)) | ("raw", mat)
        self.handedness = handedness
        self.view = np.asarray(view, dtype=f32).copy()
        aspect = _f(1.0) if aspect_ratio is None else _f(aspect_ratio)
        kind = projection[0]
        if kind == "orthographic":
            half = np.asarray(projection[1], dtype=f32) * _f(0.5)
            fn = orthographic_lh if handedness == LEFT else orthographic_rh
            self.proj = fn(-half[0], half[0], -half[1], half[1], half[2], -half[2])
        elif kind == "perspective":
            self.proj = perspective_infinite_reverse(projection[1], aspect, projection[2], handedness == RIGHT)
        elif kind == "raw":
            self.proj = np.asarray(projection[1], dtype=f32).copy()
        else:
            raise ValueError(kind)
        self.orig_view = self.view.copy()
        self.orig_view[12:16] = [0, 0, 0, 1]
        self.inv_view = mat4_inverse(self.view)
        self.view_proj = mat4_mul(self.proj, self.view)
        self.origin_view_proj = mat4_mul(self.proj, self.orig_view)
        self.location = self.inv_view[12:15].copy()


def frustum_planes(matrix, lib):
    out = np.zeros(20, dtype=f32)
    lib.r3o_frustum_from_matrix(lib.ptr(np.asarray(matrix, dtype=f32)), lib.ptr(out))
    return out


def bounding_sphere_from_mesh(positions):
    """frustum.rs:15-56: centre = AABB midpoint, radius = max distance."""
    p = np.asarray(positions, dtype=f32).reshape(-1, 3)
    if len(p) == 0:
        return np.zeros(3, dtype=f32), _f(0)
    centre = (p.max(axis=0) + p.min(axis=0)) / _f(2.0)
    d = p - centre
    dist = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
    return centre.astype(f32), f32(dist.max())


def bounding_sphere_apply_transform(centre, radius, m):
    """frustum.rs:22-32."""
    m = np.asarray(m, dtype=f32)
    l0 = _dot3(m[0:3], m[0:3])
    l1 = _dot3(m[4:7], m[4:7])
    l2 = _dot3(m[8:11], m[8:11])
    max_scale = np.sqrt(max(l0, max(l1, l2)))
    c = mat4_mul_vec4(m, np.array([centre[0], centre[1], centre[2], 1.0], dtype=f32))[:3]
    return c.astype(f32), f32(max_scale * _f(radius))


def calculate_normals(positions, indices, left_handed=True):
    """rend3-types/src/lib.rs:662-704."""
    p = np.asarray(positions, dtype=f32).reshape(-1, 3)
    idx = np.asarray(indices, dtype=np.uint32).reshape(-1, 3)
    n = np.zeros_like(p)
    for i0, i1, i2 in idx:
        e1 = p[i1] - p[i0]
        e2 = p[i2] - p[i0]
        nn = _cross(e1, e2) if left_handed else _cross(e2, e1)
        n[i0] += nn
        n[i1] += nn
        n[i2] += nn
    for i in range(len(n)):
        l2 = _dot3(n[i], n[i])
        # glam normalize_or_zero: rcp = 1/length; if rcp is finite and > 0 -> v * rcp else ZERO
        ln = np.sqrt(l2)
        with np.errstate(divide="ignore"):
            rcp = _f(1.0) / ln
        n[i] = n[i] * rcp if (np.isfinite(rcp) and rcp > 0) else 0
    return n


# ---------------------------------------------------------------------------- shadows
def shadow_camera(direction, distance, resolution, user_camera):
    """rend3/src/managers/directional/shadow_camera.rs:6-33."""
    cam_loc = user_camera.location
    texel = _f(distance) / _f(resolution)
    rh = user_camera.handedness == RIGHT
    origin_view = _look_to(np.zeros(3, dtype=f32), np.asarray(direction, dtype=f32), np.array([0, 1, 0], dtype=f32), rh)
    cov = transform_point3(origin_view, cam_loc)
    # Rust `%` on f32 == C fmodf
    off = np.array([math.fmod(float(cov[0]), float(texel)), math.fmod(float(cov[1]), float(texel))], dtype=f32)
    shadow_loc = np.array([cov[0] - off[0], cov[1] - off[1], cov[2] - _f(0.0)], dtype=f32)
    inv_origin_view = mat4_inverse(origin_view)
    new_loc = transform_point3(inv_origin_view, shadow_loc)
    view = _look_to(new_loc, np.asarray(direction, dtype=f32), np.array([0, 1, 0], dtype=f32), rh)
    # look_at(new, new + dir): direction passed to look_to is (new + dir) - new; restate exactly
    centre = new_loc + np.asarray(direction, dtype=f32)
    view = _look_to(new_loc, centre - new_loc, np.array([0, 1, 0], dtype=f32), rh)
    d = _f(distance)
    return CameraState(view, ("orthographic", (d, d, d)), user_camera.handedness, None)


def allocate_shadow_atlas(maps, max_dimension):
    """rend3/src/managers/directional/shadow_alloc.rs:59-136.
    maps: list of (handle, resolution).  Returns (texture_dimensions(x,y), [(offset(x,y), size, handle)]) or None."""
    if not maps or max_dimension == 0:
        return None
    maps = sorted(maps, key=lambda m: -m[1])  # stable, like sort_by_key(Reverse(res))

    def lz16(v):
        return 16 - int(v).bit_length()

    root_size = maps[0][1]
    min_lz = lz16(root_size)
    VACANT, LEAF, CHILDREN = 0, 1, 2
    nodes = [[VACANT, None]]
    roots = [0]

    def try_alloc(node_idx, order, handle):
        kind, payload = nodes[node_idx]
        if kind == VACANT:
            if order == 0:
                nodes[node_idx] = [LEAF, handle]
                return True
            base = len(nodes)
            nodes[node_idx] = [CHILDREN, [base, base + 1, base + 2, base + 3]]
            nodes.extend([[VACANT, None] for _ in range(4)])
            return try_alloc(node_idx, order, handle)
        if kind == LEAF:
            return False
        if order == 0:
            return False
        return any(try_alloc(ch, order - 1, handle) for ch in payload)

    for handle, res in maps:
        order = lz16(res) - min_lz
        while True:
            if try_alloc(roots[-1], order, handle):
                break
            nodes.append([VACANT, None])
            roots.append(len(nodes) - 1)

    available_columns = max_dimension // root_size
    root_count = np.float32(len(roots))
    rows_needed = np.float32(math.ceil(float(root_count / np.float32(available_columns))))
    columns_needed = int(math.ceil(float(root_count / rows_needed)))
    dims = (columns_needed * root_size, int(rows_needed) * root_size)
    queue = [
        (1, ((i % columns_needed) * root_size, (i // columns_needed) * root_size), n) for i, n in enumerate(roots)
    ]
    out = []
    while queue:
        div, off, node_idx = queue.pop(0)
        size = root_size // div
        half = size // 2
        kind, payload = nodes[node_idx]
        if kind == LEAF:
            out.append((off, size, payload))
        elif kind == CHILDREN:
            for ci, ch in enumerate(payload):
                queue.append((div * 2, (off[0] + half * (ci % 2), off[1] + half * (ci // 2)), ch))
    return dims, out


MINIMUM_SHADOW_MAP_SIZE = 32


def evaluate_directional_lights(lights, user_camera, max_dimension=16384):
    """DirectionalLightManager::evaluate, directional.rs:99-157.
    lights: list of dict(color, intensity, direction, distance, resolution) (None = removed slot).
    Returns (atlas_size(x,y), shadows[list of dict(offset,size,handle,camera)], light_buffer bytes)."""
    maps = [(i, l["resolution"]) for i, l in enumerate(lights) if l is not None]
    atlas = allocate_shadow_atlas(maps, max_dimension)
    if atlas is None:
        size = (MINIMUM_SHADOW_MAP_SIZE, MINIMUM_SHADOW_MAP_SIZE)
        return size, [], np.zeros(16, dtype=np.uint8).tobytes()
    dims, coords = atlas
    size = (max(dims[0], MINIMUM_SHADOW_MAP_SIZE), max(dims[1], MINIMUM_SHADOW_MAP_SIZE))
    sizef = np.array(size, dtype=f32)
    shadows = []
    buf = np.zeros(16 + 128 * len(coords), dtype=np.uint8)
    buf[0:4] = np.array([len(coords)], dtype=np.uint32).view(np.uint8)
    for k, (off, sz, handle) in enumerate(coords):
        l = lights[handle]
        cam = shadow_camera(l["direction"], l["distance"], l["resolution"], user_camera)
        shadows.append(dict(offset=off, size=sz, handle=handle, camera=cam))
        rec = np.zeros(32, dtype=f32)
        rec[0:16] = cam.view_proj
        rec[16:19] = np.asarray(l["color"], dtype=f32) * _f(l["intensity"])
        rec[20:23] = np.asarray(l["direction"], dtype=f32)
        rec[24:26] = _f(1.0) / sizef
        rec[26:28] = np.array(off, dtype=f32) / sizef
        rec[28:30] = _f(sz) / sizef
        buf[16 + 128 * k : 16 + 128 * (k + 1)] = rec.view(np.uint8)
    return size, shadows, buf.tobytes()


def point_light_buffer(lights):
    """rend3/src/managers/point.rs:58-74."""
    live = [l for l in lights if l is not None]
    buf = np.zeros(16 + 32 * len(live), dtype=np.uint8)
    buf[0:4] = np.array([len(live)], dtype=np.uint32).view(np.uint8)
    for k, l in enumerate(live):
        rec = np.zeros(8, dtype=f32)
        rec[0:3] = np.asarray(l["position"], dtype=f32)
        rec[3] = 1.0
        rec[4:7] = np.asarray(l["color"], dtype=f32) * _f(l["intensity"])
        rec[7] = _f(l["radius"])
        buf[16 + 32 * k : 16 + 32 * (k + 1)] = rec.view(np.uint8)
    return buf.tobytes()


# ---------------------------------------------------------------------------- uniform blocks
def front_face_positive_area_visible(handedness, shadow):
    """culler.rs:133-141,477-480 with winding = handedness.into() (rend3-types/src/lib.rs:1190-1197):
    Left -> Cw, Right -> Ccw; culling face Front for shadow cameras, Back for the viewport."""
    ccw = handedness == RIGHT
    if ccw:
        return not shadow  # (Ccw,Back)->Positive ; (Ccw,Front)->Negative
    return shadow  # (Cw,Back)->Negative ; (Cw,Front)->Positive


def camera_header(cam, shadow_index, resolution, samples, object_count, lib):
    """PerCameraUniform header, culler.rs:485-502 (240 bytes)."""
    h = np.zeros(60, dtype=f32)
    h[0:16] = cam.view
    h[16:32] = cam.view_proj
    hu = h.view(np.uint32)
    hu[32] = 0xFFFFFFFF if shadow_index is None else shadow_index
    h[36:56] = frustum_planes(cam.view_proj, lib)  # world frustum = from_matrix(proj * view), camera.rs:30
    h[56] = _f(resolution[0])
    h[57] = _f(resolution[1])
    flags = 0
    if front_face_positive_area_visible(cam.handedness, shadow_index is not None):
        flags |= 1
    if samples != 1:
        flags |= 2
    hu[58] = flags
    hu[59] = object_count
    return h


def frame_uniforms(cam, ambient, resolution, lib):
    """uniforms.rs:28-48 (496 bytes)."""
    u = np.zeros(124, dtype=f32)
    u[0:16] = cam.view
    u[16:32] = cam.view_proj
    u[32:48] = cam.origin_view_proj
    u[48:64] = mat4_inverse(cam.view)
    u[64:80] = mat4_inverse(cam.view_proj)
    u[80:96] = mat4_inverse(cam.origin_view_proj)
    u[96:116] = frustum_planes(cam.proj, lib)
    u[116:120] = np.asarray(ambient, dtype=f32)
    uu = u.view(np.uint32)
    uu[120] = resolution[0]
    uu[121] = resolution[1]
    return u


def mip_chain_texels(w, h, mips):
    return sum(max(1, w >> k) * max(1, h >> k) for k in range(mips))


def prepare_texture(lib, rgba8, srgb, mip_count, mip_source):
    """Texture upload path of rend3/src/managers/texture.rs: validates the mip count (MipmapCount::Maximum =
    floor(log2(max(w, h))) + 1, Extent3d::max_mips), lays the mips out contiguously and, for
    MipmapSource::Generated, builds them (util/mipmap.rs, restated in C: r3o_generate_mips).
    Returns (u32 texels of the whole chain, width, height, mips)."""
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if mip_source == "generated" or mip_count == 1:
        assert a.ndim == 3 and a.shape[2] == 4
        h, w = a.shape[:2]
    else:
        raise ValueError("uploaded mip chains: pass (w, h) explicitly via prepare_texture_chain")
    max_mips = int(max(w, h)).bit_length()
    mips = max_mips if mip_count == "maximum" else int(mip_count)
    assert 1 <= mips <= max_mips
    out = np.zeros(mip_chain_texels(w, h, mips), dtype=np.uint32)
    out[: w * h] = a.reshape(-1, 4).view(np.uint32).reshape(-1)
    if mips > 1:
        lib.r3o_generate_mips(1 if srgb else 0, w, h, mips, lib.ptr(out))
    return out, w, h, mips


def blend_draw_order(camera_location, object_indices, locations):
    """Back-to-front order of the blend-key objects (batching.rs:146-176 with Sorting::BLENDING: key = -distance^2 of
    camera location to object location, ascending; ties -- which the reference's unstable sort leaves open -- by
    object index).  distance_squared as glam: (dx*dx + dy*dy) + dz*dz in f32.  Returns object indices."""
    cam = np.asarray(camera_location, dtype=f32)
    keyed = []
    for idx, loc in zip(object_indices, locations):
        d = cam - np.asarray(loc, dtype=f32)
        dist = f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2])
        keyed.append((-float(f32(dist)), int(idx)))
    keyed.sort()
    return [i for _k, i in keyed]
