"""Host-side mirror of the reference's CPU math on the hot path, delegating to the C++ implementation
(rend3_amd/csrc/host.cpp, the r3n_host_* half of include/r3n.h).  Function names follow glam / rend3:

  CameraState                     rend3/src/managers/camera.rs:11-114
  frustum_planes                  rend3/src/util/frustum.rs:96-145
  shadow_camera / atlas           rend3/src/managers/directional/{shadow_camera.rs:6-33, shadow_alloc.rs:59-136}
  evaluate_directional_lights     rend3/src/managers/directional.rs:99-157
  camera_header / frame_uniforms  rend3-routine/src/culling/culler.rs:485-502, uniforms.rs:28-48

Matrices are numpy float32[16], column-major.  This module never imports the oracle.
"""
import math

import numpy as np

from . import _ffi

f32 = np.float32
LEFT, RIGHT = 0, 1
MINIMUM_SHADOW_MAP_SIZE = 32


def _m(a):
    return np.ascontiguousarray(a, dtype=f32).reshape(16)


def _v3(a):
    return np.ascontiguousarray(a, dtype=f32).reshape(3)


def identity():
    m = np.zeros(16, dtype=f32)
    m[0] = m[5] = m[10] = m[15] = 1.0
    return m


def mat4_mul(a, b):
    a, b, out = _m(a), _m(b), np.zeros(16, dtype=f32)
    _ffi.lib().r3n_host_mat4_mul(_ffi.ptr(a), _ffi.ptr(b), _ffi.ptr(out))
    return out


def mat4_inverse(m):
    m, out = _m(m), np.zeros(16, dtype=f32)
    _ffi.lib().r3n_host_mat4_inverse(_ffi.ptr(m), _ffi.ptr(out))
    return out


def look_at_lh(eye, center, up):
    out = np.zeros(16, dtype=f32)
    _ffi.lib().r3n_host_look_at(_ffi.ptr(_v3(eye)), _ffi.ptr(_v3(center)), _ffi.ptr(_v3(up)), 0, _ffi.ptr(out))
    return out


def look_at_rh(eye, center, up):
    out = np.zeros(16, dtype=f32)
    _ffi.lib().r3n_host_look_at(_ffi.ptr(_v3(eye)), _ffi.ptr(_v3(center)), _ffi.ptr(_v3(up)), 1, _ffi.ptr(out))
    return out


def orthographic_lh(l, r, b, t, n, fa):
    """glam Mat4::orthographic_lh -- scene-construction helper (the reference's tests build Raw projections with it)."""
    l, r, b, t, n, fa = (f32(x) for x in (l, r, b, t, n, fa))
    rw, rh, rd = f32(1.0) / (r - l), f32(1.0) / (t - b), f32(1.0) / (fa - n)
    m = np.zeros(16, dtype=f32)
    m[0], m[5], m[10] = rw + rw, rh + rh, rd
    m[12], m[13], m[14], m[15] = -(l + r) * rw, -(t + b) * rh, -rd * n, 1.0
    return m


def _sincos32(x):
    x = float(np.float32(x))
    return np.float32(math.sin(x)), np.float32(math.cos(x))


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
    m[0], m[5], m[10] = f32(s[0]), f32(s[1]), f32(s[2])
    return m


def from_euler_xyz(a, b, c):
    return mat4_mul(mat4_mul(rotation_x(a), rotation_y(b)), rotation_z(c))


def frustum_planes(matrix):
    out = np.zeros(20, dtype=f32)
    _ffi.lib().r3n_host_frustum_from_matrix(_ffi.ptr(_m(matrix)), _ffi.ptr(out))
    return out


def bounding_sphere_from_mesh(positions):
    p = np.ascontiguousarray(positions, dtype=f32).reshape(-1, 3)
    c, r = np.zeros(3, dtype=f32), np.zeros(1, dtype=f32)
    _ffi.lib().r3n_host_bounding_sphere_from_mesh(_ffi.ptr(p), len(p), _ffi.ptr(c), _ffi.ptr(r))
    return c, r[0]


def bounding_sphere_apply_transform(centre, radius, m):
    c, r = np.zeros(3, dtype=f32), np.zeros(1, dtype=f32)
    _ffi.lib().r3n_host_bounding_sphere_apply_transform(_ffi.ptr(_v3(centre)), float(radius), _ffi.ptr(_m(m)),
                                                        _ffi.ptr(c), _ffi.ptr(r))
    return c, r[0]


def calculate_normals(positions, indices, left_handed=True):
    p = np.ascontiguousarray(positions, dtype=f32).reshape(-1, 3)
    i = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
    n = np.zeros_like(p)
    _ffi.lib().r3n_host_calculate_normals(_ffi.ptr(p), len(p), _ffi.ptr(i), len(i), 1 if left_handed else 0, _ffi.ptr(n))
    return n


class CameraState:
    """rend3/src/managers/camera.rs:11-114."""

    def __init__(self, view, projection, handedness, aspect_ratio=None):
        self.handedness = handedness
        self.view = _m(view).copy()
        aspect = 1.0 if aspect_ratio is None else float(np.float32(aspect_ratio))
        kind = projection[0]
        rh = 1 if handedness == RIGHT else 0
        self.proj = np.zeros(16, dtype=f32)
        if kind == "orthographic":
            params = np.ascontiguousarray(projection[1], dtype=f32)
            _ffi.lib().r3n_host_projection(0, _ffi.ptr(params), rh, aspect, _ffi.ptr(self.proj))
        elif kind == "perspective":
            params = np.array([projection[1], projection[2]], dtype=f32)
            _ffi.lib().r3n_host_projection(1, _ffi.ptr(params), rh, aspect, _ffi.ptr(self.proj))
        elif kind == "raw":
            self.proj = _m(projection[1]).copy()
        else:
            raise ValueError(kind)
        self.orig_view = self.view.copy()
        self.orig_view[12:16] = [0, 0, 0, 1]
        self.inv_view = mat4_inverse(self.view)
        self.view_proj = mat4_mul(self.proj, self.view)
        self.origin_view_proj = mat4_mul(self.proj, self.orig_view)
        self.location = self.inv_view[12:15].copy()


def shadow_camera(direction, distance, resolution, user_camera):
    view, proj = np.zeros(16, dtype=f32), np.zeros(16, dtype=f32)
    _ffi.lib().r3n_host_shadow_camera(_ffi.ptr(_v3(direction)), float(np.float32(distance)), int(resolution),
                                      _ffi.ptr(_v3(user_camera.location)), 1 if user_camera.handedness == RIGHT else 0,
                                      _ffi.ptr(view), _ffi.ptr(proj))
    return CameraState(view, ("raw", proj), user_camera.handedness, None)


def allocate_shadow_atlas(maps, max_dimension):
    if not maps or max_dimension == 0:
        return None
    handles = np.array([m[0] for m in maps], dtype=np.uint32)
    res = np.array([m[1] for m in maps], dtype=np.uint16)
    dims = np.zeros(2, dtype=np.uint32)
    out = np.zeros(4 * len(maps), dtype=np.uint32)
    n = _ffi.lib().r3n_host_allocate_shadow_atlas(_ffi.ptr(handles), _ffi.ptr(res), len(maps), max_dimension,
                                                  _ffi.ptr(dims), _ffi.ptr(out))
    if n == 0:
        return None
    return (int(dims[0]), int(dims[1])), [((int(out[4 * i]), int(out[4 * i + 1])), int(out[4 * i + 2]), int(out[4 * i + 3]))
                                          for i in range(n)]


def evaluate_directional_lights(lights, user_camera, max_dimension=16384):
    """DirectionalLightManager::evaluate (directional.rs:99-157) -> (atlas size, shadow descs, light buffer bytes)."""
    maps = [(i, l["resolution"]) for i, l in enumerate(lights) if l is not None]
    atlas = allocate_shadow_atlas(maps, max_dimension)
    if atlas is None:
        return (MINIMUM_SHADOW_MAP_SIZE, MINIMUM_SHADOW_MAP_SIZE), [], np.zeros(16, dtype=np.uint8)
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
        rec[16:19] = np.asarray(l["color"], dtype=f32) * f32(l["intensity"])
        rec[20:23] = np.asarray(l["direction"], dtype=f32)
        rec[24:26] = f32(1.0) / sizef
        rec[26:28] = np.array(off, dtype=f32) / sizef
        rec[28:30] = f32(sz) / sizef
        buf[16 + 128 * k: 16 + 128 * (k + 1)] = rec.view(np.uint8)
    return size, shadows, buf


def point_light_buffer(lights):
    """PointLightManager::evaluate, rend3/src/managers/point.rs:58-74."""
    live = [l for l in lights if l is not None]
    buf = np.zeros(16 + 32 * len(live), dtype=np.uint8)
    buf[0:4] = np.array([len(live)], dtype=np.uint32).view(np.uint8)
    for k, l in enumerate(live):
        rec = np.zeros(8, dtype=f32)
        rec[0:3] = np.asarray(l["position"], dtype=f32)
        rec[3] = 1.0
        rec[4:7] = np.asarray(l["color"], dtype=f32) * f32(l["intensity"])
        rec[7] = f32(l["radius"])
        buf[16 + 32 * k: 16 + 32 * (k + 1)] = rec.view(np.uint8)
    return buf


def positive_area_visible(handedness, shadow):
    """culler.rs:133-141,477-480; winding = handedness.into() (rend3-types/src/lib.rs:1190-1197)."""
    return (not shadow) if handedness == RIGHT else shadow


def camera_header(cam, shadow_index, resolution, samples, object_count):
    """PerCameraUniform header (culler.rs:485-502), 240 bytes."""
    h = np.zeros(60, dtype=f32)
    hu = h.view(np.uint32)
    h[0:16] = cam.view
    h[16:32] = cam.view_proj
    hu[32] = 0xFFFFFFFF if shadow_index is None else shadow_index
    h[36:56] = frustum_planes(cam.view_proj)
    h[56], h[57] = f32(resolution[0]), f32(resolution[1])
    flags = (1 if positive_area_visible(cam.handedness, shadow_index is not None) else 0) | (2 if samples != 1 else 0)
    hu[58] = flags
    hu[59] = object_count
    return h


def frame_uniforms(cam, ambient, resolution):
    """FrameUniforms::new (uniforms.rs:28-48), 496 bytes."""
    u = np.zeros(124, dtype=f32)
    uu = u.view(np.uint32)
    u[0:16] = cam.view
    u[16:32] = cam.view_proj
    u[32:48] = cam.origin_view_proj
    u[48:64] = mat4_inverse(cam.view)
    u[64:80] = mat4_inverse(cam.view_proj)
    u[80:96] = mat4_inverse(cam.origin_view_proj)
    u[96:116] = frustum_planes(cam.proj)
    u[116:120] = np.asarray(ambient, dtype=f32)
    uu[120], uu[121] = resolution[0], resolution[1]
    return u


def mip_chain_texels(w, h, mips):
    return sum(max(1, w >> k) * max(1, h >> k) for k in range(mips))


def prepare_texture(rgba8, srgb, mip_count, mip_source):
    """Texture upload path of rend3/src/managers/texture.rs: mip count check (MipmapCount::Maximum =
    Extent3d::max_mips).  Returns (RGBA8 bytes of the stored levels, width, height, mips, stored levels); for
    MipmapSource::Generated only level 0 is stored and the library builds the chain on the GPU (util/mipmap.rs)."""
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    if a.ndim != 3 or a.shape[2] != 4:
        raise ValueError("texture data must be (H, W, 4) u8")
    if mip_source != "generated" and mip_count != 1:
        raise ValueError("uploaded mip chains go through add_texture_2d_encoded")
    h, w = a.shape[:2]
    max_mips = int(max(w, h)).bit_length()
    mips = max_mips if mip_count == "maximum" else int(mip_count)
    if not 1 <= mips <= max_mips:
        raise ValueError("mip_count out of range")
    return a.reshape(-1), w, h, mips, 1 if mips > 1 else 0


def blend_draw_order(camera_location, object_indices, locations):
    """Back-to-front order of the blend-key objects (rend3-routine/src/culling/batching.rs:146-176 with
    Sorting::BLENDING: key = -distance^2 from the camera location to the object location, ascending; ties -- which the
    reference's unstable sort leaves open -- by object index).  distance_squared as glam: (dx*dx + dy*dy) + dz*dz in f32."""
    cam = np.asarray(camera_location, dtype=f32)
    keyed = []
    for idx, loc in zip(object_indices, locations):
        d = cam - np.asarray(loc, dtype=f32)
        dist = f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2])
        keyed.append((-float(f32(dist)), int(idx)))
    keyed.sort()
    return [i for _k, i in keyed]
