// host.cpp -- host-side mirror of the reference's CPU math on the hot path (the r3n_host_* half of
// include/r3n.h).  In a real rend3 integration this stays in Rust (rend3 core + glam); it exists so the
// standalone harness and bench can build the C-ABI's inputs.  f32 throughout, fixed operation order,
// built with -ffp-contract=off (DESIGN.md "Arithmetic contract").
//
// Follows (reference file:line):
//   CameraState / compute_projection_matrix   rend3/src/managers/camera.rs:23-114
//   Frustum::from_matrix, contains_sphere     rend3/src/util/frustum.rs:96-161
//   BoundingSphere                            rend3/src/util/frustum.rs:15-56
//   shadow_camera                             rend3/src/managers/directional/shadow_camera.rs:6-33
//   allocate_shadow_atlas                     rend3/src/managers/directional/shadow_alloc.rs:59-136
//   calculate_normals_for_buffers             rend3-types/src/lib.rs:662-704
//   glam 0.25 (un-vendored, rend3/Cargo.toml:43): look_at_*, orthographic_*,
//   perspective_infinite_reverse_*, Mat4 mul/inverse -- restated from its documented conventions
//   (SURVEY.md App. E).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <vector>

#include "../../include/r3n.h"

namespace {

inline void mul_vec4(const float *m, float x, float y, float z, float w, float *o) {
    for (int r = 0; r < 4; ++r) o[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}
inline float dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline void cross3(const float *a, const float *b, float *o) {
    float x = a[1] * b[2] - a[2] * b[1];
    float y = a[2] * b[0] - a[0] * b[2];
    float z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
inline void normalize3(float *v) {
    float r = 1.0f / std::sqrt(dot3(v, v));
    v[0] *= r; v[1] *= r; v[2] *= r;
}

// glam look_to_{lh,rh}
void look_to(const float *eye, const float *fwd, const float *up, int rh, float *m) {
    float f[3] = {fwd[0], fwd[1], fwd[2]};
    normalize3(f);
    float s[3], u[3];
    if (rh) {
        cross3(f, up, s);
        normalize3(s);
        cross3(s, f, u);
        float cols[16] = {s[0], u[0], -f[0], 0, s[1], u[1], -f[1], 0, s[2], u[2], -f[2], 0,
                          -dot3(eye, s), -dot3(eye, u), dot3(eye, f), 1};
        std::memcpy(m, cols, sizeof cols);
    } else {
        cross3(up, f, s);
        normalize3(s);
        cross3(f, s, u);
        float cols[16] = {s[0], u[0], f[0], 0, s[1], u[1], f[1], 0, s[2], u[2], f[2], 0,
                          -dot3(eye, s), -dot3(eye, u), -dot3(eye, f), 1};
        std::memcpy(m, cols, sizeof cols);
    }
}

void orthographic(float l, float r, float b, float t, float n, float f, int rh, float *m) {
    float rw = 1.0f / (r - l);
    float rhh = 1.0f / (t - b);
    float rd = rh ? 1.0f / (n - f) : 1.0f / (f - n);
    std::memset(m, 0, 16 * sizeof(float));
    m[0] = rw + rw;
    m[5] = rhh + rhh;
    m[10] = rd;
    m[12] = -(l + r) * rw;
    m[13] = -(t + b) * rhh;
    m[14] = rh ? rd * n : -rd * n;
    m[15] = 1.0f;
}

}  // namespace

extern "C" {

void r3n_host_mat4_mul(const float *a, const float *b, float *out) {
    float tmp[16];
    for (int c = 0; c < 4; ++c) mul_vec4(a, b[4 * c], b[4 * c + 1], b[4 * c + 2], b[4 * c + 3], tmp + 4 * c);
    std::memcpy(out, tmp, sizeof tmp);
}

void r3n_host_mat4_inverse(const float *m, float *out) {
    // cofactor expansion; m[4*c + r]
    const float m00 = m[0], m01 = m[1], m02 = m[2], m03 = m[3];
    const float m10 = m[4], m11 = m[5], m12 = m[6], m13 = m[7];
    const float m20 = m[8], m21 = m[9], m22 = m[10], m23 = m[11];
    const float m30 = m[12], m31 = m[13], m32 = m[14], m33 = m[15];
    const float c00 = m22 * m33 - m32 * m23;
    const float c02 = m12 * m33 - m32 * m13;
    const float c03 = m12 * m23 - m22 * m13;
    const float c04 = m21 * m33 - m31 * m23;
    const float c06 = m11 * m33 - m31 * m13;
    const float c07 = m11 * m23 - m21 * m13;
    const float c08 = m21 * m32 - m31 * m22;
    const float c10 = m11 * m32 - m31 * m12;
    const float c11 = m11 * m22 - m21 * m12;
    const float c12 = m20 * m33 - m30 * m23;
    const float c14 = m10 * m33 - m30 * m13;
    const float c15 = m10 * m23 - m20 * m13;
    const float c16 = m20 * m32 - m30 * m22;
    const float c18 = m10 * m32 - m30 * m12;
    const float c19 = m10 * m22 - m20 * m12;
    const float c20 = m20 * m31 - m30 * m21;
    const float c22 = m10 * m31 - m30 * m11;
    const float c23 = m10 * m21 - m20 * m11;
    const float i00 = (m11 * c00 - m12 * c04) + m13 * c08;
    const float i01 = -((m01 * c00 - m02 * c04) + m03 * c08);
    const float i02 = (m01 * c02 - m02 * c06) + m03 * c10;
    const float i03 = -((m01 * c03 - m02 * c07) + m03 * c11);
    const float i10 = -((m10 * c00 - m12 * c12) + m13 * c16);
    const float i11 = (m00 * c00 - m02 * c12) + m03 * c16;
    const float i12 = -((m00 * c02 - m02 * c14) + m03 * c18);
    const float i13 = (m00 * c03 - m02 * c15) + m03 * c19;
    const float i20 = (m10 * c04 - m11 * c12) + m13 * c20;
    const float i21 = -((m00 * c04 - m01 * c12) + m03 * c20);
    const float i22 = (m00 * c06 - m01 * c14) + m03 * c22;
    const float i23 = -((m00 * c07 - m01 * c15) + m03 * c23);
    const float i30 = -((m10 * c08 - m11 * c16) + m12 * c20);
    const float i31 = (m00 * c08 - m01 * c16) + m02 * c20;
    const float i32 = -((m00 * c10 - m01 * c18) + m02 * c22);
    const float i33 = (m00 * c11 - m01 * c19) + m02 * c23;
    const float det = ((m00 * i00 + m01 * i10) + m02 * i20) + m03 * i30;
    const float rdet = 1.0f / det;
    const float inv[16] = {i00, i01, i02, i03, i10, i11, i12, i13, i20, i21, i22, i23, i30, i31, i32, i33};
    for (int k = 0; k < 16; ++k) out[k] = inv[k] * rdet;
}

void r3n_host_look_at(const float eye[3], const float center[3], const float up[3], int rh, float *out) {
    float dir[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
    look_to(eye, dir, up, rh, out);
}

void r3n_host_projection(int kind, const float *params, int rh, float aspect_ratio, float *out) {
    if (kind == 0) {
        // camera.rs:90-97: near = +half.z, far = -half.z  => reverse-Z
        float hx = params[0] * 0.5f, hy = params[1] * 0.5f, hz = params[2] * 0.5f;
        orthographic(-hx, hx, -hy, hy, hz, -hz, rh, out);
    } else {
        // glam perspective_infinite_reverse_{lh,rh}; sin/cos evaluated in double, rounded once
        float fov = params[0] * 0.017453292519943295f;
        float half = 0.5f * fov;
        float s = (float)std::sin((double)half), c = (float)std::cos((double)half);
        float h = c / s;
        float w = h / aspect_ratio;
        std::memset(out, 0, 16 * sizeof(float));
        out[0] = w;
        out[5] = h;
        out[11] = rh ? -1.0f : 1.0f;
        out[14] = params[1];
    }
}

void r3n_host_frustum_from_matrix(const float *m, float *planes) {
    const float sgn[5] = {1.0f, -1.0f, -1.0f, 1.0f, -1.0f};
    const int row[5] = {0, 0, 1, 1, 2};
    for (int p = 0; p < 5; ++p) {
        float a = m[3] + sgn[p] * m[row[p]];
        float b = m[7] + sgn[p] * m[4 + row[p]];
        float c = m[11] + sgn[p] * m[8 + row[p]];
        float d = m[15] + sgn[p] * m[12 + row[p]];
        float mag = std::sqrt((a * a + b * b) + c * c);
        planes[4 * p + 0] = a / mag;
        planes[4 * p + 1] = b / mag;
        planes[4 * p + 2] = c / mag;
        planes[4 * p + 3] = d / mag;
    }
}

int r3n_host_frustum_contains_sphere(const float *planes, const float center[3], float radius) {
    float neg_radius = -radius;
    for (int p = 0; p < 5; ++p) {
        float dist = dot3(planes + 4 * p, center) + planes[4 * p + 3];
        if (!(dist >= neg_radius)) return 0;
    }
    return 1;
}

void r3n_host_bounding_sphere_from_mesh(const float *positions, uint64_t n, float out_center[3], float *out_radius) {
    if (n == 0) {
        out_center[0] = out_center[1] = out_center[2] = 0.0f;
        *out_radius = 0.0f;
        return;
    }
    float mx[3] = {positions[0], positions[1], positions[2]}, mn[3] = {positions[0], positions[1], positions[2]};
    for (uint64_t i = 1; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            mx[c] = std::max(mx[c], positions[3 * i + c]);
            mn[c] = std::min(mn[c], positions[3 * i + c]);
        }
    for (int c = 0; c < 3; ++c) out_center[c] = (mx[c] + mn[c]) / 2.0f;
    float r = 0.0f;
    for (uint64_t i = 0; i < n; ++i) {
        float d[3] = {positions[3 * i] - out_center[0], positions[3 * i + 1] - out_center[1],
                      positions[3 * i + 2] - out_center[2]};
        r = std::max(r, std::sqrt(dot3(d, d)));
    }
    *out_radius = r;
}

void r3n_host_bounding_sphere_apply_transform(const float center[3], float radius, const float *m, float out_center[3],
                                              float *out_radius) {
    float l0 = dot3(m, m), l1 = dot3(m + 4, m + 4), l2 = dot3(m + 8, m + 8);
    float max_scale = std::sqrt(std::max(l0, std::max(l1, l2)));
    float c[4];
    mul_vec4(m, center[0], center[1], center[2], 1.0f, c);
    out_center[0] = c[0]; out_center[1] = c[1]; out_center[2] = c[2];
    *out_radius = max_scale * radius;
}

void r3n_host_build_object_records(uint32_t n, const float *transforms, const float *mesh_desc, const uint32_t *mesh_u32,
                                   const uint32_t *material_index, r3n_object128 *out) {
    for (uint32_t i = 0; i < n; ++i) {
        r3n_object128 &o = out[i];
        std::memset(&o, 0, sizeof o);
        std::memcpy(o.transform, transforms + 16 * (size_t)i, 64);
        r3n_host_bounding_sphere_apply_transform(mesh_desc + 4 * (size_t)i, mesh_desc[4 * (size_t)i + 3], o.transform,
                                                 o.bounding_sphere_center, &o.bounding_sphere_radius);
        o.first_index = mesh_u32[8 * (size_t)i];
        o.index_count = mesh_u32[8 * (size_t)i + 1];
        o.material_index = material_index[i];
        for (int k = 0; k < 6; ++k) o.vertex_attribute_start_offsets[k] = mesh_u32[8 * (size_t)i + 2 + k];
        o.enabled = 1;
    }
}

void r3n_host_calculate_normals(const float *positions, uint64_t vertex_count, const uint32_t *indices,
                                uint64_t index_count, int left_handed, float *normals) {
    std::memset(normals, 0, sizeof(float) * 3 * vertex_count);
    for (uint64_t t = 0; t + 2 < index_count; t += 3) {
        const float *p1 = positions + 3 * (uint64_t)indices[t];
        const float *p2 = positions + 3 * (uint64_t)indices[t + 1];
        const float *p3 = positions + 3 * (uint64_t)indices[t + 2];
        float e1[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
        float e2[3] = {p3[0] - p1[0], p3[1] - p1[1], p3[2] - p1[2]};
        float n[3];
        if (left_handed) cross3(e1, e2, n); else cross3(e2, e1, n);
        for (int k = 0; k < 3; ++k)
            for (int c = 0; c < 3; ++c) normals[3 * (uint64_t)indices[t + k] + c] += n[c];
    }
    for (uint64_t i = 0; i < vertex_count; ++i) {
        float *n = normals + 3 * i;
        float rcp = 1.0f / std::sqrt(dot3(n, n));  // glam normalize_or_zero
        if (std::isfinite(rcp) && rcp > 0.0f) { n[0] *= rcp; n[1] *= rcp; n[2] *= rcp; }
        else { n[0] = n[1] = n[2] = 0.0f; }
    }
}

void r3n_host_shadow_camera(const float direction[3], float distance, uint32_t resolution,
                            const float camera_location[3], int rh, float *out_view, float *out_proj) {
    const float zero[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    float texel = distance / (float)resolution;
    float origin_view[16];
    look_to(zero, direction, up, rh, origin_view);
    float cov[4];
    mul_vec4(origin_view, camera_location[0], camera_location[1], camera_location[2], 1.0f, cov);
    float off[2] = {std::fmod(cov[0], texel), std::fmod(cov[1], texel)};  // Rust f32 `%`
    float shadow_loc[3] = {cov[0] - off[0], cov[1] - off[1], cov[2] - 0.0f};
    float inv_origin_view[16];
    r3n_host_mat4_inverse(origin_view, inv_origin_view);
    float nl[4];
    mul_vec4(inv_origin_view, shadow_loc[0], shadow_loc[1], shadow_loc[2], 1.0f, nl);
    float centre[3] = {nl[0] + direction[0], nl[1] + direction[1], nl[2] + direction[2]};
    r3n_host_look_at(nl, centre, up, rh, out_view);
    float size[3] = {distance, distance, distance};
    r3n_host_projection(0, size, rh, 1.0f, out_proj);
}

uint32_t r3n_host_allocate_shadow_atlas(const uint32_t *handles, const uint16_t *resolutions, uint32_t n,
                                        uint32_t max_dimension, uint32_t out_dimensions[2], uint32_t *out_maps) {
    if (n == 0 || max_dimension == 0) return 0;
    struct Map { uint32_t handle; uint16_t res; };
    std::vector<Map> maps(n);
    for (uint32_t i = 0; i < n; ++i) maps[i] = {handles[i], resolutions[i]};
    std::stable_sort(maps.begin(), maps.end(), [](const Map &a, const Map &b) { return a.res > b.res; });
    auto lz16 = [](uint16_t v) { uint32_t k = 0; for (int b = 15; b >= 0 && !((v >> b) & 1); --b) ++k; return k; };

    enum Kind { Vacant, Leaf, Children };
    struct Node { Kind kind; uint32_t handle; uint32_t child[4]; };
    std::vector<Node> nodes;
    std::vector<uint32_t> roots;
    nodes.push_back({Vacant, 0, {0, 0, 0, 0}});
    roots.push_back(0);
    struct Alloc {
        std::vector<Node> &nodes;
        bool run(uint32_t idx, uint32_t order, uint32_t handle) {
            Kind k = nodes[idx].kind;
            if (k == Vacant) {
                if (order == 0) { nodes[idx].kind = Leaf; nodes[idx].handle = handle; return true; }
                uint32_t base = (uint32_t)nodes.size();
                nodes[idx].kind = Children;
                for (uint32_t c = 0; c < 4; ++c) nodes[idx].child[c] = base + c;
                for (uint32_t c = 0; c < 4; ++c) nodes.push_back({Vacant, 0, {0, 0, 0, 0}});
                return run(idx, order, handle);
            }
            if (k == Leaf) return false;
            if (order == 0) return false;
            for (uint32_t c = 0; c < 4; ++c) {
                uint32_t ch = nodes[idx].child[c];
                if (run(ch, order - 1, handle)) return true;
            }
            return false;
        }
    } alloc{nodes};

    uint32_t root_size = maps[0].res;
    uint32_t min_lz = lz16((uint16_t)root_size);
    for (const Map &m : maps) {
        uint32_t order = lz16(m.res) - min_lz;
        for (;;) {
            if (alloc.run(roots.back(), order, m.handle)) break;
            nodes.push_back({Vacant, 0, {0, 0, 0, 0}});
            roots.push_back((uint32_t)nodes.size() - 1);
        }
    }
    uint32_t available_columns = max_dimension / root_size;
    float root_count = (float)roots.size();
    float rows_needed = std::ceil(root_count / (float)available_columns);
    uint32_t columns_needed = (uint32_t)std::ceil(root_count / rows_needed);
    out_dimensions[0] = columns_needed * root_size;
    out_dimensions[1] = (uint32_t)rows_needed * root_size;

    struct Visit { uint32_t div, ox, oy, node; };
    std::deque<Visit> queue;
    for (uint32_t i = 0; i < roots.size(); ++i)
        queue.push_back({1, (i % columns_needed) * root_size, (i / columns_needed) * root_size, roots[i]});
    uint32_t written = 0;
    while (!queue.empty()) {
        Visit v = queue.front();
        queue.pop_front();
        uint32_t size = root_size / v.div, half = size / 2;
        const Node &nd = nodes[v.node];
        if (nd.kind == Leaf) {
            out_maps[4 * written + 0] = v.ox;
            out_maps[4 * written + 1] = v.oy;
            out_maps[4 * written + 2] = size;
            out_maps[4 * written + 3] = nd.handle;
            ++written;
        } else if (nd.kind == Children) {
            for (uint32_t c = 0; c < 4; ++c)
                queue.push_back({v.div * 2, v.ox + half * (c % 2), v.oy + half * (c / 2), nd.child[c]});
        }
    }
    return written;
}

// CameraState::new (camera.rs:23-85): proj, view_proj = proj * view, origin_view_proj (view without its translation), location
static void camera_state(const float *view, const float *proj, float *view_proj, float *origin_view_proj, float *location) {
    float orig[16];
    std::memcpy(orig, view, sizeof orig);
    orig[12] = 0.0f; orig[13] = 0.0f; orig[14] = 0.0f; orig[15] = 1.0f;
    r3n_host_mat4_mul(proj, view, view_proj);
    if (origin_view_proj) r3n_host_mat4_mul(proj, orig, origin_view_proj);
    if (location) {
        float inv[16];
        r3n_host_mat4_inverse(view, inv);
        location[0] = inv[12]; location[1] = inv[13]; location[2] = inv[14];
    }
}

// PerCameraUniform header (culler.rs:477-502)
static void camera_header(const float *view, const float *view_proj, uint32_t shadow_index, float res_x, float res_y, int rh, uint32_t samples,
                          uint32_t object_count, r3n_camera_header240 *h) {
    std::memset(h, 0, sizeof *h);
    std::memcpy(h->view, view, 64);
    std::memcpy(h->view_proj, view_proj, 64);
    h->shadow_index = shadow_index;
    r3n_host_frustum_from_matrix(view_proj, h->frustum);
    h->resolution[0] = res_x; h->resolution[1] = res_y;
    const bool shadow = shadow_index != 0xFFFFFFFFu;
    // culler.rs:133-141,477-480 with winding = handedness.into() (rend3-types/src/lib.rs:1190-1197)
    const bool positive_area_visible = rh ? !shadow : shadow;
    h->flags = (positive_area_visible ? 1u : 0u) | (samples != 1u ? 2u : 0u);
    h->object_count = object_count;
}

int r3n_host_evaluate_frame(const r3n_host_camera144 *cam, const r3n_host_directional_light48 *lights, uint32_t n_lights,
                            uint32_t max_atlas_dimension, const float ambient[4], uint32_t width, uint32_t height, uint32_t samples,
                            uint32_t object_capacity, r3n_host_frame *out) {
    const int rh = cam->handedness != 0u;
    float proj[16];
    if (cam->projection_kind == 2u) std::memcpy(proj, cam->projection_params, 64);
    else r3n_host_projection((int)cam->projection_kind, cam->projection_params, rh, cam->aspect_ratio > 0.0f ? cam->aspect_ratio : 1.0f, proj);
    float origin_view_proj[16];
    camera_state(cam->view, proj, out->view_proj, origin_view_proj, out->camera_location);
    // FrameUniforms::new (uniforms.rs:28-48)
    r3n_frame_uniforms496 &u = out->uniforms;
    std::memset(&u, 0, sizeof u);
    std::memcpy(u.view, cam->view, 64);
    std::memcpy(u.view_proj, out->view_proj, 64);
    std::memcpy(u.origin_view_proj, origin_view_proj, 64);
    r3n_host_mat4_inverse(cam->view, u.inv_view);
    r3n_host_mat4_inverse(out->view_proj, u.inv_view_proj);
    r3n_host_mat4_inverse(origin_view_proj, u.inv_origin_view_proj);
    r3n_host_frustum_from_matrix(proj, u.frustum);
    std::memcpy(u.ambient, ambient, 16);
    u.resolution[0] = width; u.resolution[1] = height;
    camera_header(cam->view, out->view_proj, 0xFFFFFFFFu, (float)width, (float)height, rh, samples, object_capacity, &out->viewport_header);
    // DirectionalLightManager::evaluate (directional.rs:99-157)
    uint32_t handles[R3N_MAX_SHADOW_VIEWS];
    uint16_t res[R3N_MAX_SHADOW_VIEWS];
    uint32_t n = 0;
    for (uint32_t i = 0; i < n_lights; ++i) {
        if (lights[i].resolution == 0u) continue;
        if (n == R3N_MAX_SHADOW_VIEWS) return -1;
        handles[n] = i;
        res[n] = (uint16_t)lights[i].resolution;
        ++n;
    }
    uint32_t dims[2] = {0, 0}, maps[4 * R3N_MAX_SHADOW_VIEWS];
    const uint32_t placed = r3n_host_allocate_shadow_atlas(handles, res, n, max_atlas_dimension, dims, maps);
    const uint32_t kMin = 32u;  // MINIMUM_SHADOW_MAP_SIZE (directional.rs:24)
    out->shadow_atlas_width = std::max(dims[0], kMin);
    out->shadow_atlas_height = std::max(dims[1], kMin);
    out->n_shadow_views = placed;
    std::memset(out->directional_buffer, 0, 16);
    std::memcpy(out->directional_buffer, &placed, 4);
    out->directional_bytes = 16u + 128u * (uint64_t)placed;
    const float sizef[2] = {(float)out->shadow_atlas_width, (float)out->shadow_atlas_height};
    for (uint32_t k = 0; k < placed; ++k) {
        const uint32_t ox = maps[4 * k], oy = maps[4 * k + 1], sz = maps[4 * k + 2], handle = maps[4 * k + 3];
        const r3n_host_directional_light48 &l = lights[handle];
        float sview[16], sproj[16], svp[16];
        r3n_host_shadow_camera(l.direction, l.distance, l.resolution, out->camera_location, rh, sview, sproj);
        camera_state(sview, sproj, svp, nullptr, nullptr);
        r3n_shadow_view272 &v = out->shadow_views[k];
        std::memset(&v, 0, sizeof v);
        camera_header(sview, svp, k, (float)sz, (float)sz, rh, 1u, object_capacity, &v.header);
        v.x = ox; v.y = oy; v.size = sz;
        out->shadow_handles[k] = handle;
        float rec[32];
        std::memset(rec, 0, sizeof rec);
        std::memcpy(rec, svp, 64);
        for (int c = 0; c < 3; ++c) rec[16 + c] = l.color[c] * l.intensity;
        for (int c = 0; c < 3; ++c) rec[20 + c] = l.direction[c];
        rec[24] = 1.0f / sizef[0]; rec[25] = 1.0f / sizef[1];
        rec[26] = (float)ox / sizef[0]; rec[27] = (float)oy / sizef[1];
        rec[28] = (float)sz / sizef[0]; rec[29] = (float)sz / sizef[1];
        std::memcpy(out->directional_buffer + 16 + 128 * (size_t)k, rec, 128);
    }
    return 0;
}

}  // extern "C"
