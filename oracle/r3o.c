/*
 * r3o.c -- CPU ORACLE for the rend3 GPU-driven object pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (rend3_amd/, include/)
 * may include, link or call this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker / reported baseline.
 *
 * It is a plain-C restatement (f32, fixed operation order, built with
 * -ffp-contract=off) of the reference's WGSL shaders and of the fixed-function
 * raster state they run under:
 *
 *   uniform bake          rend3-routine/shaders/src/uniform_prep.wgsl:9-27
 *   frustum / sphere      rend3/src/util/frustum.rs:96-161
 *   per-triangle cull     rend3-routine/shaders/src/cull.wgsl:243-324,326-390
 *   Hi-Z pyramid          rend3-routine/shaders/src/hi_z.wgsl:19-32
 *   depth-only raster     rend3-routine/shaders/src/depth.wgsl:51-127
 *                         + pipeline state rend3-routine/src/forward.rs:318-371
 *   opaque VS + FS        rend3-routine/shaders/src/opaque.wgsl:91-135,203-551
 *                         math/brdf.wgsl, math/color.wgsl, math/matrix.wgsl, shadow/pcf.wgsl
 *   tonemap blit          rend3-routine/shaders/src/blit.wgsl:22-31, tonemapping.rs:44
 *
 * PARITY PINNING: the reference cannot be executed in this environment (no Rust, no
 * Vulkan ICD).  The oracle is pinned against the reference's own golden images
 * (rend3-test/tests/results/ PNGs, examples/src/cube/screenshot.png; committed as
 * tests/golden/) by tests/test_oracle_goldens.py.  Per-triangle cull decisions and
 * HDR float values are parity-UNPINNED by the reference itself (no reference test
 * reads them back); for those the oracle is the definition.
 *
 * Choices where WGSL / the GPU leave the result implementation-defined (all
 * documented in DESIGN.md "Arithmetic contract"):
 *   - mat4*vec4 = ((c0*x + c1*y) + c2*z) + c3*w, no FMA contraction.
 *   - rasteriser: 2D homogeneous edge functions (no clipping, no snapping),
 *     top-left rule, pixel centres at +0.5, depth = sum(E_i*z_i)/det, depth clip
 *     0<=z<=1, reverse-Z GreaterEqual; exact depth ties resolved toward the larger
 *     canonical triangle slot (the reference's atomics make ties unordered).
 *   - out-of-range / NaN Hi-Z coordinates clamp into the mip, mip index clamps to
 *     the last level, OOB Hi-Z source loads read 0.0.
 *   - pow(x,5) = ((x*x)*(x*x))*x ; normalize(v) = v * (1/sqrt(dot(v,v))).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* Scopes of the op tally (oracle/tally.h, the counting build): which STAGE of the reference's pipeline the arithmetic that
 * follows belongs to -- 0 fixed-function work (triangle setup, coverage, attribute interpolation), 1 vs_main, 2 fs_main.
 * Nothing in the oracle proper: the macro is empty. */
#ifndef R3O_SCOPE
#define R3O_SCOPE(n) ((void)0)
#endif

#define R3O_INVALID 0xFFFFFFFFu

/* ------------------------------------------------------------------ layouts */
/* Object record, 128 B (rend3/src/managers/object.rs:23-36). */
typedef struct {
    float transform[16];
    float centre[3];
    float radius;
    uint32_t first_index;
    uint32_t index_count;
    uint32_t material_index;
    uint32_t attr_off[6]; /* pos, normal, tangent, uv0, uv1, color0 (pbr/material.rs:486-494) */
    uint32_t enabled;
    uint32_t _pad[2];
} r3o_object;

/* Per-camera uniform header, 240 B (rend3-routine/src/culling/culler.rs:158-175). */
typedef struct {
    float view[16];
    float view_proj[16];
    uint32_t shadow_index;
    uint32_t _pad0[3];
    float frustum[20];
    float resolution[2];
    uint32_t flags;
    uint32_t object_count;
} r3o_camera_header;

/* Per-object baked matrices, 128 B (culler.rs:177-183). */
typedef struct {
    float model_view[16];
    float model_view_proj[16];
} r3o_baked;

/* Material record, 208 B (rend3/src/managers/material.rs:25-29 + pbr/material.rs:526-543). */
typedef struct {
    uint32_t tex[10];
    uint32_t _pad[2];
    float uv_transform0[12];
    float uv_transform1[12];
    float albedo[4];
    float emissive[3];
    float roughness;
    float metallic;
    float reflectance;
    float clear_coat;
    float clear_coat_roughness;
    float anisotropy;
    float ambient_occlusion;
    float alpha_cutout;
    uint32_t flags;
} r3o_material;

/* Directional light, 128 B stride (rend3/src/managers/directional.rs:38-53). */
typedef struct {
    float view_proj[16];
    float color[3];
    float _p0;
    float direction[3];
    float _p1;
    float inv_resolution[2];
    float atlas_offset[2];
    float atlas_size[2];
    float _p2[2];
} r3o_dir_light;

/* Point light, 32 B (rend3/src/managers/point.rs:21-26). */
typedef struct {
    float position[4];
    float color[3];
    float radius;
} r3o_point_light;

/* FrameUniforms, 496 B (rend3-routine/src/uniforms.rs:17-27). */
typedef struct {
    float view[16];
    float view_proj[16];
    float origin_view_proj[16];
    float inv_view[16];
    float inv_view_proj[16];
    float inv_origin_view_proj[16];
    float frustum[20];
    float ambient[4];
    uint32_t resolution[2];
    uint32_t _pad[2];
} r3o_frame_uniforms;

#define FLAGS_ALBEDO_ACTIVE 0x0001u
#define FLAGS_ALBEDO_BLEND 0x0002u
#define FLAGS_ALBEDO_VERTEX_SRGB 0x0004u
#define FLAGS_BICOMPONENT_NORMAL 0x0008u
#define FLAGS_SWIZZLED_NORMAL 0x0010u
#define FLAGS_YDOWN_NORMAL 0x0020u
#define FLAGS_AOMR_COMBINED 0x0040u
#define FLAGS_AOMR_SWIZZLED_SPLIT 0x0080u
#define FLAGS_AOMR_SPLIT 0x0100u
#define FLAGS_AOMR_BW_SPLIT 0x0200u
#define FLAGS_CC_GLTF_COMBINED 0x0400u
#define FLAGS_CC_GLTF_SPLIT 0x0800u
#define FLAGS_CC_BW_SPLIT 0x1000u
#define FLAGS_UNLIT 0x2000u
#define FLAGS_NEAREST 0x4000u

#define PCU_POSITIVE_AREA_VISIBLE 0x1u
#define PCU_MULTISAMPLED 0x2u

/* ------------------------------------------------------------------ math */
static inline void mat4_mul_vec4(const float *m, float x, float y, float z, float w, float *o) {
    for (int r = 0; r < 4; ++r) o[r] = ((m[0 + r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

void r3o_mat4_mul(const float *a, const float *b, float *out) {
    float tmp[16];
    for (int c = 0; c < 4; ++c) mat4_mul_vec4(a, b[4 * c + 0], b[4 * c + 1], b[4 * c + 2], b[4 * c + 3], tmp + 4 * c);
    memcpy(out, tmp, sizeof tmp);
}

static inline float dot3(const float *a, const float *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static inline float sat(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
static inline void normalize3(float *v) {
    float r = 1.0f / sqrtf(dot3(v, v));
    v[0] *= r; v[1] *= r; v[2] *= r;
}
/* mat3 (columns c0,c1,c2 given as pointers to 3 floats) * vec3 */
static inline void mat3_mul_vec3(const float *c0, const float *c1, const float *c2, const float *v, float *o) {
    for (int r = 0; r < 3; ++r) o[r] = (c0[r] * v[0] + c1[r] * v[1]) + c2[r] * v[2];
}

/* float -> IEEE half, round to nearest even; half -> float exact. */
static uint16_t f32_to_f16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ex = (x >> 23) & 0xFFu;
    uint32_t man = x & 0x7FFFFFu;
    if (ex == 0xFF) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0));
    int32_t e = (int32_t)ex - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        man |= 0x800000u;
        uint32_t shift = (uint32_t)(14 - e);
        uint32_t h = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)e << 10) | (man >> 13);
    uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}
static float f16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t ex = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t x;
    if (ex == 0) {
        if (man == 0) x = sign;
        else {
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            x = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (ex == 31) x = sign | 0x7F800000u | (man << 13);
    else x = sign | ((ex + 127 - 15) << 23) | (man << 13);
    float f; memcpy(&f, &x, 4);
    return f;
}
uint16_t r3o_f32_to_f16(float f) { return f32_to_f16(f); }
float r3o_f16_to_f32(uint16_t h) { return f16_to_f32(h); }

/* ------------------------------------------------------------------ K1: uniform bake */
/* uniform_prep.wgsl:9-27: runs over buffer capacity, leaves disabled slots untouched. */
void r3o_uniform_bake(const r3o_camera_header *hdr, const r3o_object *objects, r3o_baked *out) {
    for (uint32_t i = 0; i < hdr->object_count; ++i) {
        if (objects[i].enabled == 0u) continue;
        r3o_mat4_mul(hdr->view, objects[i].transform, out[i].model_view);
        r3o_mat4_mul(hdr->view_proj, objects[i].transform, out[i].model_view_proj);
    }
}

/* ------------------------------------------------------------------ A2: frustum */
/* frustum.rs:96-145 -- planes left,right,top,bottom,near from a column-major matrix, normalised. */
void r3o_frustum_from_matrix(const float *m, float *planes) {
    /* mat_arr[c][r] = m[4*c+r] */
    const float sgn[5] = {1.0f, -1.0f, -1.0f, 1.0f, -1.0f};
    const int row[5] = {0, 0, 1, 1, 2};
    for (int p = 0; p < 5; ++p) {
        float a = m[0 * 4 + 3] + sgn[p] * m[0 * 4 + row[p]];
        float b = m[1 * 4 + 3] + sgn[p] * m[1 * 4 + row[p]];
        float c = m[2 * 4 + 3] + sgn[p] * m[2 * 4 + row[p]];
        float d = m[3 * 4 + 3] + sgn[p] * m[3 * 4 + row[p]];
        float mag = sqrtf((a * a + b * b) + c * c);
        planes[4 * p + 0] = a / mag;
        planes[4 * p + 1] = b / mag;
        planes[4 * p + 2] = c / mag;
        planes[4 * p + 3] = d / mag;
    }
}

/* frustum.rs:148-161 */
int r3o_frustum_contains_sphere(const float *planes, const float *centre, float radius) {
    float neg_radius = -radius;
    for (int p = 0; p < 5; ++p) {
        float dist = dot3(planes + 4 * p, centre) + planes[4 * p + 3];
        if (!(dist >= neg_radius)) return 0;
    }
    return 1;
}

/* L1 visible set: batching.rs:144-148 over live objects.  Disabled slots are not visible. */
void r3o_frustum_cull(const r3o_camera_header *hdr, const r3o_object *objects, uint8_t *visible) {
    for (uint32_t i = 0; i < hdr->object_count; ++i) {
        visible[i] = (objects[i].enabled != 0u && objects[i].index_count >= 3u &&
                      r3o_frustum_contains_sphere(hdr->frustum, objects[i].centre, objects[i].radius))
                         ? 1 : 0;
    }
}

/* ------------------------------------------------------------------ Hi-Z */
typedef struct {
    const float *data;  /* mips stored consecutively, mip0 first */
    uint32_t width, height, mips;
} r3o_hiz;

static inline uint32_t mip_dim(uint32_t d, uint32_t k) { uint32_t v = d >> k; return v ? v : 1u; }
uint32_t r3o_hiz_mip_count(uint32_t w, uint32_t h) {
    uint32_t m = w > h ? w : h, n = 0;
    while (m) { n++; m >>= 1; }
    return n;
}
uint64_t r3o_hiz_mip_offset(uint32_t w, uint32_t h, uint32_t mip) {
    uint64_t off = 0;
    for (uint32_t k = 0; k < mip; ++k) off += (uint64_t)mip_dim(w, k) * mip_dim(h, k);
    return off;
}
/* hi_z.wgsl:19-32; mip0 (depth) must already be at pyr[0..w*h). */
void r3o_hiz_build(float *pyr, uint32_t w, uint32_t h) {
    uint32_t mips = r3o_hiz_mip_count(w, h);
    for (uint32_t k = 1; k < mips; ++k) {
        const float *src = pyr + r3o_hiz_mip_offset(w, h, k - 1);
        float *dst = pyr + r3o_hiz_mip_offset(w, h, k);
        uint32_t sw = mip_dim(w, k - 1), sh = mip_dim(h, k - 1), dw = mip_dim(w, k), dh = mip_dim(h, k);
        uint32_t ox = sw & 1u, oy = sh & 1u;
        for (uint32_t y = 0; y < dh; ++y)
            for (uint32_t x = 0; x < dw; ++x) {
                float nearest = 1.0f;
                for (uint32_t ix = 0; ix < 2u + ox; ++ix)
                    for (uint32_t iy = 0; iy < 2u + oy; ++iy) {
                        uint32_t sx = 2u * x + ix, sy = 2u * y + iy;
                        float v = (sx < sw && sy < sh) ? src[(uint64_t)sy * sw + sx] : 0.0f;
                        nearest = fminf(nearest, v);
                    }
                dst[(uint64_t)y * dw + x] = nearest;
            }
    }
}

/* float -> texel coordinate with NaN/out-of-range made deterministic. */
static inline uint32_t clamp_texel(float v, uint32_t dim) {
    if (!(v >= 0.0f)) return 0u; /* negative or NaN */
    float top = (float)(dim - 1u);
    if (v >= top) return dim - 1u;
    return (uint32_t)v;
}

/* cull.wgsl:243-262 */
static float hiz_sample_min(const r3o_hiz *hz, float u, float v, uint32_t mip) {
    uint32_t mw = mip_dim(hz->width, mip), mh = mip_dim(hz->height, mip);
    const float *tex = hz->data + r3o_hiz_mip_offset(hz->width, hz->height, mip);
    float px = u * (float)mw - 0.5f;
    float py = v * (float)mh - 0.5f;
    uint32_t lx = clamp_texel(fmaxf(floorf(px), 0.0f), mw);
    uint32_t ly = clamp_texel(fmaxf(floorf(py), 0.0f), mh);
    uint32_t hx = clamp_texel(fminf(ceilf(px), (float)mw - 1.0f), mw);
    uint32_t hy = clamp_texel(fminf(ceilf(py), (float)mh - 1.0f), mh);
    float m = tex[(uint64_t)ly * mw + lx];
    m = fminf(m, tex[(uint64_t)ly * mw + hx]);
    m = fminf(m, tex[(uint64_t)hy * mw + lx]);
    m = fminf(m, tex[(uint64_t)hy * mw + hx]);
    return m;
}

/* ceil(log2(max(x,1))) computed exactly from the float's exponent. */
static uint32_t ceil_log2_ge1(float x) {
    x = fmaxf(x, 1.0f);  /* NaN -> 1 */
    if (isinf(x)) return 1000u;
    int e;
    float m = frexpf(x, &e); /* x = m*2^e, m in [0.5,1) */
    return (uint32_t)((m == 0.5f) ? e - 1 : e);
}

/* ------------------------------------------------------------------ vertex fetch */
static inline void fetch_vec3(const uint32_t *mesh, uint32_t byte_off, uint32_t vtx, float *o) {
    uint32_t w = byte_off / 4u + vtx * 3u; /* vertex_attributes.wgsl:51-58 */
    memcpy(o, mesh + w, 12);
}

static inline float det3_xyw(const float *p0, const float *p1, const float *p2) {
    /* determinant(mat3x3(p0.xyw, p1.xyw, p2.xyw)), columns a,b,c */
    float ax = p0[0], ay = p0[1], az = p0[3];
    float bx = p1[0], by = p1[1], bz = p1[3];
    float cx = p2[0], cy = p2[1], cz = p2[3];
    return (ax * (by * cz - cy * bz) - bx * (ay * cz - cy * az)) + cx * (ay * bz - by * az);
}

/* cull.wgsl:264-324 */
static int execute_culling(const r3o_camera_header *hdr, const float *mvp, const float v[3][3], const r3o_hiz *hz) {
    float p[3][4];
    for (int k = 0; k < 3; ++k) mat4_mul_vec4(mvp, v[k][0], v[k][1], v[k][2], 1.0f, p[k]);
    float det = det3_xyw(p[0], p[1], p[2]);
    if ((hdr->flags & PCU_POSITIVE_AREA_VISIBLE) && det <= 0.0f) return 0;
    if (!(hdr->flags & PCU_POSITIVE_AREA_VISIBLE) && det >= 0.0f) return 0;

    float ndc[3][3];
    for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 3; ++c) ndc[k][c] = p[k][c] / p[k][3];
    float mn[2], mx[2];
    for (int c = 0; c < 2; ++c) {
        mn[c] = fminf(ndc[0][c], fminf(ndc[1][c], ndc[2][c]));
        mx[c] = fmaxf(ndc[0][c], fmaxf(ndc[1][c], ndc[2][c]));
    }
    float half_res[2] = {hdr->resolution[0] / 2.0f, hdr->resolution[1] / 2.0f};
    float smin[2], smax[2];
    for (int c = 0; c < 2; ++c) {
        smin[c] = (mn[c] + 1.0f) * half_res[c];
        smax[c] = (mx[c] + 1.0f) * half_res[c];
    }
    if (!(hdr->flags & PCU_MULTISAMPLED)) {
        /* WGSL round = ties to even = rintf under the default rounding mode */
        if (rintf(smin[0]) == rintf(smax[0]) || rintf(smin[1]) == rintf(smax[1])) return 0;
    }
    if (hdr->shadow_index != R3O_INVALID) return 1;

    float mintc[2] = {(mn[0] + 1.0f) / 2.0f, (mn[1] + 1.0f) / 2.0f};
    float maxtc[2] = {(mx[0] + 1.0f) / 2.0f, (mx[1] + 1.0f) / 2.0f};
    mintc[1] = 1.0f - mintc[1];
    maxtc[1] = 1.0f - maxtc[1];
    float uv[2] = {(maxtc[0] + mintc[0]) / 2.0f, (maxtc[1] + mintc[1]) / 2.0f};
    float edges[2] = {smax[0] - smin[0], smax[1] - smin[1]};
    float longest = fmaxf(edges[0], edges[1]);
    uint32_t mip = ceil_log2_ge1(longest);
    if (mip > hz->mips - 1u) mip = hz->mips - 1u;
    float depth = fmaxf(fmaxf(ndc[0][2], ndc[1][2]), ndc[2][2]);
    float occ = hiz_sample_min(hz, uv[0], uv[1], mip);
    if (depth < occ) return 0;
    return 1;
}

/*
 * K2 for one camera.  Canonical triangle slot of (object o, triangle t) = tri_base[o] + t where
 * tri_base is the exclusive scan of index_count/3 over object slots (built by the caller).
 * pass[slot]     = 1 iff object o is in the L1 visible set and execute_culling passes (L2 set).
 * residual[slot] = 1 iff pass && viewport camera && the triangle did not pass last frame
 *                  (cull.wgsl:362-372, get_previous_culling_result :152-160).
 * prev_tri_base[o] = last frame's base, or INVALID when o was not in last frame's batch
 * (batching.rs:226).  prev_pass may be NULL on the first frame.
 */
void r3o_cull_triangles(const r3o_camera_header *hdr, const r3o_object *objects, const uint32_t *mesh,
                        const r3o_baked *baked, const uint8_t *visible, const uint32_t *tri_base,
                        const float *hiz_data, uint32_t hiz_w, uint32_t hiz_h,
                        const uint32_t *prev_tri_base, const uint8_t *prev_pass,
                        uint8_t *pass, uint8_t *residual) {
    r3o_hiz hz = {hiz_data, hiz_w, hiz_h, hiz_data ? r3o_hiz_mip_count(hiz_w, hiz_h) : 0};
    int shadow = hdr->shadow_index != R3O_INVALID;
#pragma omp parallel for schedule(dynamic, 16)
    for (uint32_t o = 0; o < hdr->object_count; ++o) {
        if (!visible[o]) continue;
        const r3o_object *ob = &objects[o];
        uint32_t ntri = ob->index_count / 3u;
        for (uint32_t t = 0; t < ntri; ++t) {
            uint32_t idx[3];
            float v[3][3];
            for (int k = 0; k < 3; ++k) {
                idx[k] = mesh[ob->first_index + t * 3u + (uint32_t)k];
                fetch_vec3(mesh, ob->attr_off[0], idx[k], v[k]);
            }
            int ok = execute_culling(hdr, baked[o].model_view_proj, v, &hz);
            uint32_t slot = tri_base[o] + t;
            pass[slot] = (uint8_t)ok;
            if (residual) {
                int prev = 0;
                if (prev_pass && prev_tri_base && prev_tri_base[o] != R3O_INVALID) prev = prev_pass[prev_tri_base[o] + t];
                residual[slot] = (uint8_t)(ok && !shadow && !prev);
            }
        }
    }
}

/* ------------------------------------------------------------------ rasteriser */
typedef struct {
    float e[3][3];   /* oriented edge functions (A,B,C): inside >= 0 */
    float z[3];      /* clip-space z per vertex */
    float det;       /* oriented determinant (> 0) */
    int valid;
    /* EXPERIMENT ONLY (r3o_set_depth_mode != 0): window-space positions and z / w of the vertices, the depth plane through them */
    float sx[3], sy[3], zn[3], dzdx, dzdy, zc;
} tri_setup;

/*
 * Homogeneous 2D triangle setup in viewport-local pixel space.
 * Xh = (x + w) * W/2, Yh = (w - y) * H/2 (y down), third coordinate w.
 * e0 = v1 x v2, e1 = v2 x v0, e2 = v0 x v1; det = v0 . e0.
 * A triangle with positive area in NDC (cull.wgsl's det > 0) has det < 0 here because of the
 * y flip; forward.rs:338-342 front-face/cull-mode folded into `positive_visible` exactly as
 * culler.rs:133-141 does.
 */
/* EXPERIMENT ONLY (tests/test_oracle_goldens.py::test_subpixel_snapping_experiment): hardware rasterisers
 * snap window coordinates to a sub-pixel grid (8 fractional bits on the GPUs wgpu runs on) before edge setup; the contract
 * here does not.  r3o_set_snap_bits(n > 0) snaps x/w and y/w to 2^-n pixels (and rebuilds the homogeneous coordinates) so the
 * effect on the reference's self-shadowed goldens can be measured.  0 = the contract (default). */
static int g_snap_bits = 0;
void r3o_set_snap_bits(int bits) { g_snap_bits = bits; }
/* EXPERIMENT ONLY (tests/test_oracle_goldens.py::test_depth_interpolation_experiment): how a fixed-function rasteriser might
 * interpolate depth, against the contract's sum(E_i * z_i) / det.  0 = the contract (default).  1 = plane equation: z / w per vertex,
 * f32 gradients from the window-space triangle, z = zn0 + dzdx * (x - x0) + dzdy * (y - y0).  2 = barycentric weights first:
 * l_i = E_i / det rounded one by one, z = (l0 * z0 + l1 * z1) + l2 * z2.  3 = mode 1 on vertices snapped to 8 sub-pixel bits. */
/* Round 4: the plane through the window-space vertices (mode 1's idea) IS the contract now -- see setup_triangle -- because it
 * reproduces the reference's self-shadowed screenshots (animation: mean 1.62 -> 0.02 LSB) where the homogeneous quotient
 * sum(E_i * z_i) / det did not; 7 = that former contract, kept for the table in tests/test_oracle_goldens.py. */
static int g_depth_mode = 0;
/* 4 = mode 1 folded: z = (dzdx * x + dzdy * y) + c, c = zn0 - dzdx * x0 - dzdy * y0.  5 = the same plane from the homogeneous edge
 * coefficients, no division by a vertex's w (valid for triangles that cross w = 0): dzdx = sum(A_i z_i) / det, dzdy = sum(B_i z_i) / det,
 * c = sum(C_i z_i) / det.  6 = mode 5's gradients around the first vertex when all w > 0 (else mode 5). */
void r3o_set_depth_mode(int mode) { g_depth_mode = mode; g_snap_bits = mode == 3 ? 8 : 0; }
static void setup_triangle(const float *mvp, const float v[3][3], float half_w, float half_h, int positive_visible,
                           tri_setup *ts) {
    float h[3][3];
    for (int k = 0; k < 3; ++k) {
        float p[4];
        mat4_mul_vec4(mvp, v[k][0], v[k][1], v[k][2], 1.0f, p);
        h[k][0] = (p[0] + p[3]) * half_w;
        h[k][1] = (p[3] - p[1]) * half_h;
        h[k][2] = p[3];
        ts->z[k] = p[2];
        if (g_snap_bits > 0 && p[3] > 0.0f) {
            const float g = (float)(1 << g_snap_bits);
            h[k][0] = (rintf(h[k][0] / p[3] * g) / g) * p[3];
            h[k][1] = (rintf(h[k][1] / p[3] * g) / g) * p[3];
        }
        if (g_depth_mode != 0) {
            ts->sx[k] = h[k][0] / p[3]; ts->sy[k] = h[k][1] / p[3]; ts->zn[k] = p[2] / p[3];
        }
    }
    if (g_depth_mode == 1 || g_depth_mode == 3 || g_depth_mode == 4) {
        const float ax = ts->sx[1] - ts->sx[0], ay = ts->sy[1] - ts->sy[0], bx = ts->sx[2] - ts->sx[0], by = ts->sy[2] - ts->sy[0];
        const float az = ts->zn[1] - ts->zn[0], bz = ts->zn[2] - ts->zn[0];
        const float area = ax * by - bx * ay;
        ts->dzdx = (az * by - bz * ay) / area;
        ts->dzdy = (bz * ax - az * bx) / area;
        ts->zc = (ts->zn[0] - ts->dzdx * ts->sx[0]) - ts->dzdy * ts->sy[0];
    }
    for (int i = 0; i < 3; ++i) {
        const float *a = h[(i + 1) % 3], *b = h[(i + 2) % 3];
        ts->e[i][0] = a[1] * b[2] - a[2] * b[1];
        ts->e[i][1] = a[2] * b[0] - a[0] * b[2];
        ts->e[i][2] = a[0] * b[1] - a[1] * b[0];
    }
    float det = (h[0][0] * ts->e[0][0] + h[0][1] * ts->e[0][1]) + h[0][2] * ts->e[0][2];
    ts->valid = positive_visible ? (det < 0.0f) : (det > 0.0f);
    if (!ts->valid) return;
    if (det < 0.0f) {
        det = -det;
        for (int i = 0; i < 3; ++i)
            for (int c = 0; c < 3; ++c) ts->e[i][c] = -ts->e[i][c];
    }
    ts->det = det;
    /* Depth plane (the contract): depth is affine in window space, z(x, y) = (dzdx * x + dzdy * y) + zc.
     * With every vertex in front of the eye plane the plane is the one through the three window-space vertices
     * (x / w, y / w, z / w), anchored at vertex 0 -- what a fixed-function rasteriser's setup computes, and what makes a surface's
     * rasterised depth agree with the depth the fragment stage derives for the same surface when it looks itself up in a
     * shadow map (no depth bias in the reference: that comparison is decided by the last bits).  A triangle with a vertex
     * at w <= 0 (or a degenerate window-space area) takes the same plane from the homogeneous edge coefficients instead, which
     * needs no division by w. */
    if (g_depth_mode == 0) {
        int planar = 0;
        if (h[0][2] > 0.0f && h[1][2] > 0.0f && h[2][2] > 0.0f) {
            float sxk[3], syk[3], znk[3];
            for (int k = 0; k < 3; ++k) {
                const float rw = 1.0f / h[k][2];
                sxk[k] = h[k][0] * rw; syk[k] = h[k][1] * rw; znk[k] = ts->z[k] * rw;
            }
            const float ax = sxk[1] - sxk[0], ay = syk[1] - syk[0], bx = sxk[2] - sxk[0], by = syk[2] - syk[0];
            const float az = znk[1] - znk[0], bz = znk[2] - znk[0];
            const float ia = 1.0f / (ax * by - bx * ay);
            const float gx = (az * by - bz * ay) * ia, gy = (bz * ax - az * bx) * ia;
            const float c = (znk[0] - gx * sxk[0]) - gy * syk[0];
            if (gx - gx == 0.0f && gy - gy == 0.0f && c - c == 0.0f) {
                ts->dzdx = gx; ts->dzdy = gy; ts->zc = c;
                planar = 1;
            }
        }
        if (!planar) {
            ts->dzdx = ((ts->e[0][0] * ts->z[0] + ts->e[1][0] * ts->z[1]) + ts->e[2][0] * ts->z[2]) / det;
            ts->dzdy = ((ts->e[0][1] * ts->z[0] + ts->e[1][1] * ts->z[1]) + ts->e[2][1] * ts->z[2]) / det;
            ts->zc = ((ts->e[0][2] * ts->z[0] + ts->e[1][2] * ts->z[1]) + ts->e[2][2] * ts->z[2]) / det;
        }
    }
    if (g_depth_mode == 5 || g_depth_mode == 6) {
        ts->dzdx = ((ts->e[0][0] * ts->z[0] + ts->e[1][0] * ts->z[1]) + ts->e[2][0] * ts->z[2]) / det;
        ts->dzdy = ((ts->e[0][1] * ts->z[0] + ts->e[1][1] * ts->z[1]) + ts->e[2][1] * ts->z[2]) / det;
        ts->zc = ((ts->e[0][2] * ts->z[0] + ts->e[1][2] * ts->z[1]) + ts->e[2][2] * ts->z[2]) / det;
    }
    /* conservative pixel bounds are computed by the caller */
    (void)h;
}

/* Evaluate the three edge functions at a point (all of them: the shading of a multisampled pixel needs E at the pixel
 * centre even when the centre itself is not covered); returns 1 if covered (top-left rule). */
static inline int edge_eval(const tri_setup *ts, float px, float py, float E[3]) {
    int in = 1;
    for (int i = 0; i < 3; ++i) {
        float A = ts->e[i][0], B = ts->e[i][1];
        float v = (A * px + B * py) + ts->e[i][2];
        E[i] = v;
        if (v > 0.0f) continue;
        if (v == 0.0f && (A > 0.0f || (A == 0.0f && B > 0.0f))) continue;
        in = 0; /* negative, NaN, or on a non-owned edge */
    }
    return in;
}

/* Conservative integer pixel bounds of a triangle in a (vw x vh) viewport. */
static void tri_bounds(const float *mvp, const float v[3][3], float half_w, float half_h, int vw, int vh, int *x0,
                       int *y0, int *x1, int *y1) {
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    int all_front = 1;
    for (int k = 0; k < 3; ++k) {
        float p[4];
        mat4_mul_vec4(mvp, v[k][0], v[k][1], v[k][2], 1.0f, p);
        if (!(p[3] > 0.0f)) { all_front = 0; break; }
        float sx = (p[0] / p[3] + 1.0f) * half_w;
        float sy = (1.0f - p[1] / p[3]) * half_h;
        mnx = fminf(mnx, sx); mxx = fmaxf(mxx, sx);
        mny = fminf(mny, sy); mxy = fmaxf(mxy, sy);
    }
    if (!all_front || !(mnx == mnx) || !(mny == mny) || !(mxx == mxx) || !(mxy == mxy)) {
        *x0 = 0; *y0 = 0; *x1 = vw - 1; *y1 = vh - 1;
        return;
    }
    /* one pixel of slack each side: bounds only limit the scan, coverage decides */
    float fx0 = floorf(mnx) - 1.0f, fy0 = floorf(mny) - 1.0f, fx1 = ceilf(mxx) + 1.0f, fy1 = ceilf(mxy) + 1.0f;
    *x0 = fx0 < 0.0f ? 0 : (fx0 > (float)(vw - 1) ? vw : (int)fx0);
    *y0 = fy0 < 0.0f ? 0 : (fy0 > (float)(vh - 1) ? vh : (int)fy0);
    *x1 = fx1 < 0.0f ? -1 : (fx1 > (float)(vw - 1) ? vw - 1 : (int)fx1);
    *y1 = fy1 < 0.0f ? -1 : (fy1 > (float)(vh - 1) ? vh - 1 : (int)fy1);
}

static inline float frag_depth_at(const tri_setup *ts, const float E[3], float px, float py) {
    if (g_depth_mode == 0) return (ts->dzdx * px + ts->dzdy * py) + ts->zc;
    if (g_depth_mode == 1 || g_depth_mode == 3) return (ts->zn[0] + ts->dzdx * (px - ts->sx[0])) + ts->dzdy * (py - ts->sy[0]);
    if (g_depth_mode == 4 || g_depth_mode == 5) return (ts->dzdx * px + ts->dzdy * py) + ts->zc;
    if (g_depth_mode == 6) {
        if (ts->sx[0] == ts->sx[0] && ts->sx[0] - ts->sx[0] == 0.0f) return (ts->zn[0] + ts->dzdx * (px - ts->sx[0])) + ts->dzdy * (py - ts->sy[0]);
        return (ts->dzdx * px + ts->dzdy * py) + ts->zc;
    }
    if (g_depth_mode == 2) return ((E[0] / ts->det) * ts->z[0] + (E[1] / ts->det) * ts->z[1]) + (E[2] / ts->det) * ts->z[2];
    return ((E[0] * ts->z[0] + E[1] * ts->z[1]) + E[2] * ts->z[2]) / ts->det;
}

static void fetch_triangle(const r3o_object *ob, const uint32_t *mesh, uint32_t t, uint32_t idx[3], float v[3][3]) {
    for (int k = 0; k < 3; ++k) {
        idx[k] = mesh[ob->first_index + t * 3u + (uint32_t)k];
        fetch_vec3(mesh, ob->attr_off[0], idx[k], v[k]);
    }
}

/* ------------------------------------------------------------------ textures (row N2, first slice: albedo)
 * The reference binds every texture of the world as one bindless array (rend3/src/managers/texture.rs) and samples
 * them with textureSampleGrad through one of two samplers (rend3-routine/src/common/samplers.rs:22-56): `linear`
 * (mag = min = mipmap = Linear) or `nearest` (all Nearest), both AddressMode::Repeat, anisotropy_clamp 1, LOD clamp
 * [0, 100].  Restated here for RGBA8 textures (Rgba8Unorm / Rgba8UnormSrgb):
 *   - texel -> float: c / 255, sRGB-decoded per texel before filtering (exact formula) for the sRGB format;
 *   - bilinear footprint and weights exactly like the comparison sampler above (u * w - 0.5, floor, Repeat wrap);
 *   - level of detail from the gradients: rho = max(|ddx * size|, |ddy * size|) (Euclidean lengths, f32).  With
 *     rho = m * 2^e (1 <= m < 2) the LOD is e + (m - 1): the exponent is exact and the fraction is the mantissa
 *     (a monotone piecewise-linear stand-in for log2, max error 0.086 levels; hardware LOD units are
 *     fixed-point approximations of the same kind).  No libm call, so CPU and GPU agree bit for bit.
 *     rho <= 1 (magnification), NaN gradients: level 0.  Linear: mix of the two adjacent levels with the fraction;
 *     nearest: the level nearest to the LOD (ties up).
 */
typedef struct {
    uint32_t offset;  /* first word of mip 0 in the pool (u32 words: one per RGBA8 texel, four per float texel; mips contiguous) */
    uint32_t width, height, mips;
    uint32_t format;  /* 0 = Rgba8Unorm, 1 = Rgba8UnormSrgb, 2 = four f32 per texel (the float-decoded formats, bcn.c) */
    uint32_t _pad[3];
} r3o_texture_desc;

typedef struct {
    const r3o_texture_desc *descs;
    uint32_t n;
    const uint32_t *texels;
} r3o_textures;

static float g_srgb8_to_linear[256];
static int g_srgb8_ready = 0;
static void init_srgb8(void) {
    if (g_srgb8_ready) return;
    for (int i = 0; i < 256; ++i) {
        float e = (float)i / 255.0f;
        g_srgb8_to_linear[i] = e > 0.04045f ? powf((e + 0.055f) / 1.055f, 2.4f) : e / 12.92f;
    }
    g_srgb8_ready = 1;
}
void r3o_srgb8_table(float *out256) {
    init_srgb8();
    memcpy(out256, g_srgb8_to_linear, sizeof g_srgb8_to_linear);
}

static inline uint32_t tex_mip_dim(uint32_t d, uint32_t k) { uint32_t v = d >> k; return v ? v : 1u; }
static inline uint32_t wrap_texel_i(float f, uint32_t n) { /* f = floor(coordinate); Repeat; NaN / huge -> 0 */
    long long i = (f == f && fabsf(f) < 1e9f) ? (long long)f : 0ll;
    long long w = (long long)n;
    return (uint32_t)(((i % w) + w) % w);
}
static void tex_fetch(const r3o_textures *tt, const r3o_texture_desc *d, uint32_t mip, uint32_t x, uint32_t y, float o[4]) {
    uint64_t off = 0;
    for (uint32_t k = 0; k < mip; ++k) off += (uint64_t)tex_mip_dim(d->width, k) * tex_mip_dim(d->height, k);
    off += (uint64_t)y * tex_mip_dim(d->width, mip) + x;
    if (d->format == 2u) { /* float-decoded formats (bcn.c r3o_texture_decode_level_f32): four f32 per texel, used as they are */
        memcpy(o, tt->texels + d->offset + 4u * off, 16);
        return;
    }
    uint32_t t = tt->texels[d->offset + off];
    for (int c = 0; c < 4; ++c) {
        uint32_t b = (t >> (8 * c)) & 0xFFu;
        o[c] = (d->format == 1u && c < 3) ? g_srgb8_to_linear[b] : (float)b / 255.0f;
    }
}
static void tex_bilinear(const r3o_textures *tt, const r3o_texture_desc *d, uint32_t mip, float u, float v, float o[4]) {
    uint32_t w = tex_mip_dim(d->width, mip), h = tex_mip_dim(d->height, mip);
    float tx = u * (float)w - 0.5f, ty = v * (float)h - 0.5f;
    float fx0 = floorf(tx), fy0 = floorf(ty);
    float fx = tx - fx0, fy = ty - fy0;
    if (!(fx == fx)) fx = 0.0f;
    if (!(fy == fy)) fy = 0.0f;
    uint32_t x0 = wrap_texel_i(fx0, w), x1 = wrap_texel_i(fx0 + 1.0f, w);
    uint32_t y0 = wrap_texel_i(fy0, h), y1 = wrap_texel_i(fy0 + 1.0f, h);
    float c00[4], c10[4], c01[4], c11[4];
    tex_fetch(tt, d, mip, x0, y0, c00); tex_fetch(tt, d, mip, x1, y0, c10);
    tex_fetch(tt, d, mip, x0, y1, c01); tex_fetch(tt, d, mip, x1, y1, c11);
    for (int c = 0; c < 4; ++c) {
        float top = c00[c] * (1.0f - fx) + c10[c] * fx;
        float bot = c01[c] * (1.0f - fx) + c11[c] * fx;
        o[c] = top * (1.0f - fy) + bot * fy;
    }
}
static void tex_nearest(const r3o_textures *tt, const r3o_texture_desc *d, uint32_t mip, float u, float v, float o[4]) {
    uint32_t w = tex_mip_dim(d->width, mip), h = tex_mip_dim(d->height, mip);
    tex_fetch(tt, d, mip, wrap_texel_i(floorf(u * (float)w), w), wrap_texel_i(floorf(v * (float)h), h), o);
}
/* textureSampleGrad(textures[id - 1], nearest ? nearest_sampler : primary_sampler, (u, v), ddx, ddy) */
static void tex_sample_grad(const r3o_textures *tt, uint32_t id, int nearest, float u, float v, const float ddx[2],
                            const float ddy[2], float o[4]) {
    if (id == 0u || id > tt->n) { o[0] = o[1] = o[2] = o[3] = 0.0f; return; }
    const r3o_texture_desc *d = &tt->descs[id - 1u];
    float W = (float)d->width, H = (float)d->height;
    float ax = ddx[0] * W, ay = ddx[1] * H, bx = ddy[0] * W, by = ddy[1] * H;
    float rho = fmaxf(sqrtf(ax * ax + ay * ay), sqrtf(bx * bx + by * by));
    uint32_t level = 0;
    float frac = 0.0f;
    if (rho > 1.0f && rho < INFINITY) {
        uint32_t bits; memcpy(&bits, &rho, 4);
        level = (bits >> 23) - 127u;
        frac = (float)(bits & 0x7FFFFFu) / 8388608.0f;
    } else if (rho == INFINITY) {
        level = d->mips;  /* clamped below */
    }
    if (level >= d->mips - 1u) { level = d->mips - 1u; frac = 0.0f; }
    if (nearest) {
        if (frac >= 0.5f) level += 1u;  /* level + 1 <= mips - 1 here */
        tex_nearest(tt, d, level, u, v, o);
        return;
    }
    tex_bilinear(tt, d, level, u, v, o);
    if (frac > 0.0f) {
        float hi[4];
        tex_bilinear(tt, d, level + 1u, u, v, hi);
        for (int c = 0; c < 4; ++c) o[c] = o[c] * (1.0f - frac) + hi[c] * frac;
    }
}
/*
 * MipmapSource::Generated (rend3/src/util/mipmap.rs:139-236 + rend3/shaders/mipmap.wgsl): every level is a blit of the
 * previous one through a Linear / ClampToEdge sampler at the destination texel centres, rendered into the
 * texture's own format (so an sRGB texture is decoded, filtered and re-encoded per level).  `texels` holds mip 0 on
 * entry and the whole chain on return.  Float -> unorm8: x * 255 + 0.5, truncated (sRGB: exact OETF first).
 */
static float srgb_oetf_tex(float x) {
    if (!(x > 0.0f)) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    if (x <= 0.0031308f) return x * 12.92f;
    return 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
}
void r3o_generate_mips(uint32_t format, uint32_t width, uint32_t height, uint32_t mips, uint32_t *texels) {
    init_srgb8();
    uint64_t src_off = 0;
    for (uint32_t l = 1; l < mips; ++l) {
        uint32_t sw = tex_mip_dim(width, l - 1u), sh = tex_mip_dim(height, l - 1u);
        uint32_t dw = tex_mip_dim(width, l), dh = tex_mip_dim(height, l);
        uint64_t dst_off = src_off + (uint64_t)sw * sh;
        for (uint32_t y = 0; y < dh; ++y)
            for (uint32_t x = 0; x < dw; ++x) {
                float u = ((float)x + 0.5f) / (float)dw, v = ((float)y + 0.5f) / (float)dh;
                float tx = u * (float)sw - 0.5f, ty = v * (float)sh - 0.5f;
                float fx0 = floorf(tx), fy0 = floorf(ty);
                float fx = tx - fx0, fy = ty - fy0;
                int ix = (int)fx0, iy = (int)fy0;
                uint32_t x0 = (uint32_t)(ix < 0 ? 0 : (ix > (int)sw - 1 ? (int)sw - 1 : ix));
                uint32_t x1 = (uint32_t)(ix + 1 < 0 ? 0 : (ix + 1 > (int)sw - 1 ? (int)sw - 1 : ix + 1));
                uint32_t y0 = (uint32_t)(iy < 0 ? 0 : (iy > (int)sh - 1 ? (int)sh - 1 : iy));
                uint32_t y1 = (uint32_t)(iy + 1 < 0 ? 0 : (iy + 1 > (int)sh - 1 ? (int)sh - 1 : iy + 1));
                uint32_t t[4] = {texels[src_off + (uint64_t)y0 * sw + x0], texels[src_off + (uint64_t)y0 * sw + x1],
                                 texels[src_off + (uint64_t)y1 * sw + x0], texels[src_off + (uint64_t)y1 * sw + x1]};
                uint32_t out = 0;
                for (int c = 0; c < 4; ++c) {
                    float q[4];
                    for (int k = 0; k < 4; ++k) {
                        uint32_t b = (t[k] >> (8 * c)) & 0xFFu;
                        q[k] = (format == 1u && c < 3) ? g_srgb8_to_linear[b] : (float)b / 255.0f;
                    }
                    float top = q[0] * (1.0f - fx) + q[1] * fx;
                    float bot = q[2] * (1.0f - fx) + q[3] * fx;
                    float r = top * (1.0f - fy) + bot * fy;
                    float e = (format == 1u && c < 3) ? srgb_oetf_tex(r) : fminf(fmaxf(r, 0.0f), 1.0f);
                    out |= (uint32_t)(e * 255.0f + 0.5f) << (8 * c);
                }
                texels[dst_off + (uint64_t)y * dw + x] = out;
            }
        src_off = dst_off;
    }
}

/* vertex_attributes.wgsl: vec2<f32> texture coordinates (attribute 3); missing attribute reads (0, 0) */
static inline void fetch_uv0(const r3o_object *ob, const uint32_t *mesh, uint32_t vtx, float o[2]) {
    if (ob->attr_off[3] == R3O_INVALID) { o[0] = o[1] = 0.0f; return; }
    uint32_t w = ob->attr_off[3] / 4u + vtx * 2u;
    memcpy(&o[0], &mesh[w], 4);
    memcpy(&o[1], &mesh[w + 1u], 4);
}

/* alpha for the cutout test (opaque.wgsl:213-235 / depth.wgsl:112-125); tex_alpha = albedo texture alpha or 1 */
static float material_alpha(const r3o_material *m, float tex_alpha, float vertex_alpha) {
    float alpha = 1.0f;
    if (m->flags & FLAGS_ALBEDO_ACTIVE) {
        alpha = tex_alpha;
        if (m->flags & FLAGS_ALBEDO_BLEND) alpha *= vertex_alpha;
    }
    alpha *= m->albedo[3];
    return alpha;
}

/* perspective-correct interpolation of a vec2 attribute at pixel centre (px + 0.5, py + 0.5), covered or not */
static void interp_vec2(const tri_setup *ts, const float a[3][2], int px, int py, float o[2]) {
    float E[3];
    (void)edge_eval(ts, (float)px + 0.5f, (float)py + 0.5f, E);
    float rs = 1.0f / ((E[0] + E[1]) + E[2]);
    float l0 = E[0] * rs, l1 = E[1] * rs, l2 = E[2] * rs;
    for (int c = 0; c < 2; ++c) o[c] = (l0 * a[0][c] + l1 * a[1][c]) + l2 * a[2][c];
}
/* (uv_transform * vec3(uv, 1)).xy, mat3x3 stored as three padded vec4 columns */
static void uv_transform(const float *m, const float uv[2], float o[2]) {
    for (int c = 0; c < 2; ++c) o[c] = (m[c] * uv[0] + m[4 + c] * uv[1]) + m[8 + c] * 1.0f;
}
/*
 * Fragment-stage texture coordinates of pixel (x, y) and their screen-space derivatives.  dpdx / dpdy are the
 * differences inside the pixel's 2x2 quad ("fine" derivatives: same row for dpdx, same column for dpdy), each
 * operand evaluated as its own fragment invocation would (helper invocations extrapolate the plane of the
 * triangle exactly like interp_vec2).  `m` = uv_transform0 or NULL (depth.wgsl uses the raw coordinates).
 */
static void frag_coords(const tri_setup *ts, const float uv[3][2], const float *m, int x, int y, float coords[2],
                        float ddx[2], float ddy[2]) {
    int xq = x & ~1, yq = y & ~1;
    float c[4][2]; /* (xq,y) (xq+1,y) (x,yq) (x,yq+1) */
    const int pts[4][2] = {{xq, y}, {xq + 1, y}, {x, yq}, {x, yq + 1}};
    for (int k = 0; k < 4; ++k) {
        float raw[2];
        interp_vec2(ts, uv, pts[k][0], pts[k][1], raw);
        if (m) uv_transform(m, raw, c[k]); else { c[k][0] = raw[0]; c[k][1] = raw[1]; }
    }
    const float *self = (x & 1) ? c[1] : c[0];
    coords[0] = self[0]; coords[1] = self[1];
    for (int k = 0; k < 2; ++k) { ddx[k] = c[1][k] - c[0][k]; ddy[k] = c[3][k] - c[2][k]; }
}

static float fetch_color_alpha(const r3o_object *ob, const uint32_t *mesh, uint32_t vtx) {
    if (ob->attr_off[5] == R3O_INVALID) return 1.0f;
    uint32_t w = mesh[ob->attr_off[5] / 4u + vtx];
    return (float)((w >> 24) & 0xFFu) / 255.0f;
}

/*
 * Visibility rasterisation of a list of (object, triangle) pairs into a 64-bit buffer:
 * key = depth_bits << 32 | (canonical slot + 1); larger key wins (reverse-Z GreaterEqual,
 * forward.rs:347-351).  material_keys[material] : 0 opaque, 1 cutout, 2 blend (pbr/material.rs:497-499);
 * blend objects are skipped here (drawn by the transparent pass, base.rs:451-465).
 *
 * entry_keys (NULL or one byte per list entry): the key of the DRAW RANGE the entry sits in.  The pipeline -- with or without
 * the cutout discard, pbr/routine.rs:61-83 -- belongs to the range (forward.rs:286-313), and last frame's predicted triangles
 * sit in the ranges batch_objects made LAST frame (forward.rs:224-232: the cached DrawCallSet), from Material::key() as it was then
 * (batching.rs:153).  After Renderer::update_material changed a material's transparency (renderer/mod.rs:256-266,
 * managers/material.rs:163-188: only the TYPE is fixed) the two differ for one frame: the record read here is the current one
 * (forward.rs:257: the archetype buffer), the discard is the old key's.  NULL: every entry is drawn under its material's current key.
 */
/*
 * Multisampling (row N4; forward.rs:358 MultisampleState{count}, base.rs:236-258): `samples` = 1 or 4.  The
 * buffer holds `samples` keys per pixel (pixel-major).  Coverage and depth are evaluated at the standard sample
 * positions (the D3D / Vulkan "standard sample locations" every wgpu backend uses for 4x); the fragment --
 * here only its cutout alpha -- is evaluated once per pixel at the pixel centre, whether or not the centre is
 * covered (no centroid / per-sample qualifier in opaque.wgsl).
 */
static const float SAMPLE_POS_1[1][2] = {{0.5f, 0.5f}};
static const float SAMPLE_POS_4[4][2] = {{0.375f, 0.125f}, {0.875f, 0.375f}, {0.125f, 0.625f}, {0.625f, 0.875f}};

/* The two rasterisers run their triangle loops under OpenMP.  Depth test GreaterEqual + write is a max over
 * (depth, slot) keys, which is order-independent, so the threads merge through compare-and-swap max and the result
 * equals the serial loop's (depth bits of non-negative floats order like the floats; -0 is canonicalised to +0). */
static inline void atomic_max_u64(uint64_t *dst, uint64_t v) {
    uint64_t cur = __atomic_load_n(dst, __ATOMIC_RELAXED);
    while (v > cur && !__atomic_compare_exchange_n(dst, &cur, v, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
static inline void atomic_max_u32(uint32_t *dst, uint32_t v) {
    uint32_t cur = __atomic_load_n(dst, __ATOMIC_RELAXED);
    while (v > cur && !__atomic_compare_exchange_n(dst, &cur, v, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}

void r3o_raster_visibility(const r3o_camera_header *hdr, const r3o_object *objects, const uint32_t *mesh,
                           const r3o_baked *baked, const r3o_material *materials, const uint8_t *material_keys,
                           const uint32_t *tri_base, const uint32_t *list_obj, const uint32_t *list_tri,
                           uint64_t n, uint32_t w, uint32_t h, uint32_t samples, const r3o_texture_desc *tdescs,
                           uint32_t ntex, const uint32_t *texels, uint64_t *vis, const uint8_t *entry_keys) {
    const float(*spos)[2] = samples == 4u ? SAMPLE_POS_4 : SAMPLE_POS_1;
    r3o_textures tt = {tdescs, ntex, texels};
    init_srgb8();
    float half_w = (float)w / 2.0f, half_h = (float)h / 2.0f;
    int positive_visible = (hdr->flags & PCU_POSITIVE_AREA_VISIBLE) != 0;
#pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t o = list_obj[i], t = list_tri[i];
        const r3o_object *ob = &objects[o];
        if (ob->enabled == 0u) continue; /* opaque.wgsl:104-112 */
        const r3o_material *mat = &materials[ob->material_index];
        uint8_t key = entry_keys ? entry_keys[i] : material_keys[ob->material_index];
        if (key > 1) continue;
        uint32_t idx[3];
        float v[3][3];
        fetch_triangle(ob, mesh, t, idx, v);
        tri_setup ts;
        setup_triangle(baked[o].model_view_proj, v, half_w, half_h, positive_visible, &ts);
        if (!ts.valid) continue;
        int x0, y0, x1, y1;
        tri_bounds(baked[o].model_view_proj, v, half_w, half_h, (int)w, (int)h, &x0, &y0, &x1, &y1);
        float va[3] = {1.0f, 1.0f, 1.0f}, uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
        int need_alpha = key == 1;
        int alpha_tex = need_alpha && (mat->flags & FLAGS_ALBEDO_ACTIVE) && mat->tex[0] != 0u;
        if (need_alpha)
            for (int k = 0; k < 3; ++k) {
                va[k] = fetch_color_alpha(ob, mesh, idx[k]);
                if (alpha_tex) fetch_uv0(ob, mesh, idx[k], uv[k]);
            }
        uint32_t slot = tri_base[o] + t;
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                uint32_t mask = 0;
                float zs[4];
                for (uint32_t sm = 0; sm < samples; ++sm) {
                    float E[3];
                    if (!edge_eval(&ts, (float)x + spos[sm][0], (float)y + spos[sm][1], E)) continue;
                    float z = frag_depth_at(&ts, E, (float)x + spos[sm][0], (float)y + spos[sm][1]);
                    if (!(z >= 0.0f && z <= 1.0f)) continue;
                    if (z == 0.0f) z = 0.0f; /* -0 -> +0 */
                    zs[sm] = z;
                    mask |= 1u << sm;
                }
                if (!mask) continue;
                if (need_alpha) {
                    float E[3];
                    (void)edge_eval(&ts, (float)x + 0.5f, (float)y + 0.5f, E);
                    float rs = 1.0f / ((E[0] + E[1]) + E[2]);
                    float a = ((E[0] * rs) * va[0] + (E[1] * rs) * va[1]) + (E[2] * rs) * va[2];
                    float ta = 1.0f;
                    if (alpha_tex) { /* opaque.wgsl:207-215: transformed coords, sampler by FLAGS_NEAREST */
                        float coords[2], ddx[2], ddy[2], texel[4];
                        frag_coords(&ts, uv, mat->uv_transform0, x, y, coords, ddx, ddy);
                        tex_sample_grad(&tt, mat->tex[0], (mat->flags & FLAGS_NEAREST) != 0, coords[0], coords[1], ddx, ddy, texel);
                        ta = texel[3];
                    }
                    if (material_alpha(mat, ta, a) < mat->alpha_cutout) continue;
                }
                for (uint32_t sm = 0; sm < samples; ++sm) {
                    if (!(mask & (1u << sm))) continue;
                    uint32_t zb; memcpy(&zb, &zs[sm], 4);
                    uint64_t k64 = ((uint64_t)zb << 32) | (uint64_t)(slot + 1u);
                    atomic_max_u64(&vis[((uint64_t)y * w + (uint64_t)x) * samples + sm], k64);
                }
            }
    }
}

/*
 * Depth-only rasterisation for a shadow view into an atlas viewport (depth.wgsl, base.rs:366-396).
 * Depth compare GreaterEqual + write == max.
 */
void r3o_raster_depth(const r3o_camera_header *hdr, const r3o_object *objects, const uint32_t *mesh,
                      const r3o_baked *baked, const r3o_material *materials, const uint8_t *material_keys,
                      const uint32_t *list_obj, const uint32_t *list_tri, uint64_t n, float *atlas,
                      uint32_t atlas_w, uint32_t vp_x, uint32_t vp_y, uint32_t vp_size, const r3o_texture_desc *tdescs,
                      uint32_t ntex, const uint32_t *texels) {
    r3o_textures tt = {tdescs, ntex, texels};
    init_srgb8();
    float half = (float)vp_size / 2.0f;
    int positive_visible = (hdr->flags & PCU_POSITIVE_AREA_VISIBLE) != 0;
#pragma omp parallel for schedule(dynamic, 64)
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t o = list_obj[i], t = list_tri[i];
        const r3o_object *ob = &objects[o];
        if (ob->enabled == 0u) continue;
        const r3o_material *mat = &materials[ob->material_index];
        uint8_t key = material_keys[ob->material_index];
        if (key > 1) continue;
        uint32_t idx[3];
        float v[3][3];
        fetch_triangle(ob, mesh, t, idx, v);
        tri_setup ts;
        setup_triangle(baked[o].model_view_proj, v, half, half, positive_visible, &ts);
        if (!ts.valid) continue;
        int x0, y0, x1, y1;
        tri_bounds(baked[o].model_view_proj, v, half, half, (int)vp_size, (int)vp_size, &x0, &y0, &x1, &y1);
        float va[3] = {1.0f, 1.0f, 1.0f}, uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
        int need_alpha = key == 1;
        int alpha_tex = need_alpha && (mat->flags & FLAGS_ALBEDO_ACTIVE) && mat->tex[0] != 0u;
        if (need_alpha)
            for (int k = 0; k < 3; ++k) {
                va[k] = fetch_color_alpha(ob, mesh, idx[k]);
                if (alpha_tex) fetch_uv0(ob, mesh, idx[k], uv[k]);
            }
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) {
                float E[3];
                if (!edge_eval(&ts, (float)x + 0.5f, (float)y + 0.5f, E)) continue;
                float z = frag_depth_at(&ts, E, (float)x + 0.5f, (float)y + 0.5f);
                if (!(z >= 0.0f && z <= 1.0f)) continue;
                if (need_alpha) {
                    float rs = 1.0f / ((E[0] + E[1]) + E[2]);
                    float a = ((E[0] * rs) * va[0] + (E[1] * rs) * va[1]) + (E[2] * rs) * va[2];
                    float ta = 1.0f;
                    if (alpha_tex) { /* depth.wgsl:108-118 quirks reproduced: raw coords0 (no uv_transform), always the
                                      * primary sampler, and uvdy = dpdx(coords) */
                        float coords[2], ddx[2], ddy[2], texel[4];
                        frag_coords(&ts, uv, NULL, x, y, coords, ddx, ddy);
                        tex_sample_grad(&tt, mat->tex[0], 0, coords[0], coords[1], ddx, ddx, texel);
                        ta = texel[3];
                    }
                    if (material_alpha(mat, ta, a) < mat->alpha_cutout) continue;
                }
                if (z == 0.0f) z = 0.0f; /* -0 -> +0 */
                uint32_t zb; memcpy(&zb, &z, 4);
                atomic_max_u32((uint32_t *)&atlas[(uint64_t)(vp_y + (uint32_t)y) * atlas_w + vp_x + (uint32_t)x], zb);
            }
    }
}

/* depth plane of a visibility buffer (high 32 bits), background = 0.0.  With multisampling the single-sample
 * depth the Hi-Z pyramid starts from is resolve_depth_min.wgsl:19-27: nearest = 1.0, min over the samples. */
void r3o_vis_to_depth(const uint64_t *vis, uint64_t n, uint32_t samples, float *depth) {
    for (uint64_t i = 0; i < n; ++i) {
        if (samples == 1u) {
            uint32_t zb = (uint32_t)(vis[i] >> 32);
            memcpy(&depth[i], &zb, 4);
        } else {
            float nearest = 1.0f;
            for (uint32_t sm = 0; sm < samples; ++sm) {
                uint32_t zb = (uint32_t)(vis[i * samples + sm] >> 32);
                float z; memcpy(&z, &zb, 4);
                nearest = fminf(nearest, z);
            }
            depth[i] = nearest;
        }
    }
}

/* ------------------------------------------------------------------ shading */
/* pcf.wgsl + comparison sampler common/samplers.rs:24,42-57 (linear, GreaterEqual, Repeat). */
static float sample_compare(const float *atlas, uint32_t aw, uint32_t ah, float u, float v, float ref, int ox, int oy) {
    float tx = (u * (float)aw - 0.5f) + (float)ox;
    float ty = (v * (float)ah - 0.5f) + (float)oy;
    float fx0 = floorf(tx), fy0 = floorf(ty);
    float fx = tx - fx0, fy = ty - fy0;
    /* Repeat addressing; NaN / huge coordinates -> texel 0 deterministically */
    int64_t ix = (fx0 == fx0 && fabsf(fx0) < 1e9f) ? (int64_t)fx0 : 0;
    int64_t iy = (fy0 == fy0 && fabsf(fy0) < 1e9f) ? (int64_t)fy0 : 0;
    if (!(fx == fx)) fx = 0.0f;
    if (!(fy == fy)) fy = 0.0f;
    uint32_t x0 = (uint32_t)(((ix % (int64_t)aw) + aw) % aw), x1 = (uint32_t)((((ix + 1) % (int64_t)aw) + aw) % aw);
    uint32_t y0 = (uint32_t)(((iy % (int64_t)ah) + ah) % ah), y1 = (uint32_t)((((iy + 1) % (int64_t)ah) + ah) % ah);
    float c00 = ref >= atlas[(uint64_t)y0 * aw + x0] ? 1.0f : 0.0f;
    float c10 = ref >= atlas[(uint64_t)y0 * aw + x1] ? 1.0f : 0.0f;
    float c01 = ref >= atlas[(uint64_t)y1 * aw + x0] ? 1.0f : 0.0f;
    float c11 = ref >= atlas[(uint64_t)y1 * aw + x1] ? 1.0f : 0.0f;
    float top = c00 * (1.0f - fx) + c10 * fx;
    float bot = c01 * (1.0f - fx) + c11 * fx;
    return top * (1.0f - fy) + bot * fy;
}

static float shadow_pcf5(const float *atlas, uint32_t aw, uint32_t ah, float u, float v, float ref) {
    float r = 0.0f;
    r = r + sample_compare(atlas, aw, ah, u, v, ref, 0, 0);
    r = r + sample_compare(atlas, aw, ah, u, v, ref, 0, 1);
    r = r + sample_compare(atlas, aw, ah, u, v, ref, 0, -1);
    r = r + sample_compare(atlas, aw, ah, u, v, ref, 1, 0);
    r = r + sample_compare(atlas, aw, ah, u, v, ref, -1, 0);
    return r * 0.2f;
}

typedef struct {
    float albedo[4];
    float diffuse[3];
    float roughness;
    float normal[3];
    float f0[3];
    float emissive[3];
    float ao;
} pixel_data;

#define R3O_PI 3.14159265359f

/* opaque.wgsl:440-468 */
static void surface_shading(const float *l, const float *intensity, const pixel_data *px, const float *v,
                            float occlusion, float *out) {
    const float *n = px->normal;
    float h[3] = {v[0] + l[0], v[1] + l[1], v[2] + l[2]};
    normalize3(h);
    float nov = fabsf(dot3(n, v)) + 0.00001f;
    float nol = sat(dot3(n, l));
    float noh = sat(dot3(n, h));
    float loh = sat(dot3(l, h));
    float c165[3] = {16.5f, 16.5f, 16.5f};
    float f90 = sat(dot3(px->f0, c165));
    float a = px->roughness;
    float a2 = a * a;
    float f = (noh * a2 - noh) * noh + 1.0f;
    float d = a2 / ((R3O_PI * f) * f);
    float x = 1.0f - loh, x2 = x * x, x5 = (x2 * x2) * x;
    float ggxl = nov * sqrtf((-nol * a2 + nol) * nol + a2);
    float ggxv = nol * sqrtf((-nov * a2 + nov) * nov + a2);
    float vis = 0.5f / (ggxl + ggxv);
    float k = nol * occlusion;
    for (int c = 0; c < 3; ++c) {
        float fres = px->f0[c] + (f90 - px->f0[c]) * x5;
        float fr = (d * vis) * fres;
        float fd = px->diffuse[c] * (1.0f / R3O_PI);
        float color = fd + fr;
        out[c] = (color * intensity[c]) * k;
    }
}

static float srgb_to_linear(float e) {
    return e > 0.04045f ? powf((e + 0.055f) / 1.055f, 2.4f) : e / 12.92f;
}

/*
 * Deferred evaluation of opaque.wgsl's VS+FS for the nearest fragment of every pixel.
 * light_mats[i] = dir_lights[i].view_proj * uniforms.inv_view (opaque.wgsl:491), built by the caller
 * with r3o_mat4_mul; point_view_pos[i] = (uniforms.view * light.position).xyz (opaque.wgsl:528).
 * Output: Rgba16Float bits (base.rs:236-244).
 */
typedef struct {
    uint32_t w, h;
    const r3o_frame_uniforms *fu;
    const r3o_camera_header *hdr;
    const r3o_object *objects;
    const uint32_t *mesh;
    const r3o_baked *baked;
    const r3o_material *materials;
    const uint32_t *tri_base;
    uint32_t n_dir;
    const r3o_dir_light *dir;
    uint32_t n_point;
    const r3o_point_light *point;
    const float *atlas;
    uint32_t atlas_w, atlas_h;
    const float *light_mats, *light_l, *pview;
    r3o_textures tt;
} shade_ctx;

/* opaque.wgsl VS (:91-135) + FS (:203-551) for triangle slot `id - 1` at the centre of pixel (x, y) */
static void shade_fragment(const shade_ctx *sc, uint32_t id, uint32_t x, uint32_t y, float out[4]) {
    const uint32_t w = sc->w, h = sc->h;
    const r3o_frame_uniforms *fu = sc->fu;
    const r3o_camera_header *hdr = sc->hdr;
    const r3o_object *objects = sc->objects;
    const uint32_t *mesh = sc->mesh;
    const r3o_baked *baked = sc->baked;
    const r3o_material *materials = sc->materials;
    const uint32_t *tri_base = sc->tri_base;
    const uint32_t n_dir = sc->n_dir, n_point = sc->n_point;
    const r3o_dir_light *dir = sc->dir;
    const r3o_point_light *point = sc->point;
    const float *atlas = sc->atlas;
    const uint32_t atlas_w = sc->atlas_w, atlas_h = sc->atlas_h;
    const float *light_mats = sc->light_mats, *light_l = sc->light_l, *pview = sc->pview;
    float half_w = (float)w / 2.0f, half_h = (float)h / 2.0f;
    int positive_visible = (hdr->flags & PCU_POSITIVE_AREA_VISIBLE) != 0;
    uint32_t nobj = hdr->object_count;
    uint32_t slot = id - 1u;
    R3O_SCOPE(0);
    /* object = last slot o with tri_base[o] <= slot and a non-empty range */
    uint32_t lo = 0, hi = nobj;
    while (hi - lo > 1u) {
        uint32_t mid = lo + (hi - lo) / 2u;
        if (tri_base[mid] <= slot) lo = mid; else hi = mid;
    }
    uint32_t o = lo, t = slot - tri_base[o];
    const r3o_object *ob = &objects[o];
    const r3o_material *mat = &materials[ob->material_index];
    uint32_t idx[3];
    float v[3][3];
    fetch_triangle(ob, mesh, t, idx, v);
    tri_setup ts;
    setup_triangle(baked[o].model_view_proj, v, half_w, half_h, positive_visible, &ts);
    float E[3];
    (void)edge_eval(&ts, (float)x + 0.5f, (float)y + 0.5f, E);
    float rs = 1.0f / ((E[0] + E[1]) + E[2]);
    float lam[3] = {E[0] * rs, E[1] * rs, E[2] * rs};

    /* vertex stage, opaque.wgsl:114-134 */
    R3O_SCOPE(1);
    const float *mv = baked[o].model_view;
    float inv_s2[3] = {1.0f / dot3(mv + 0, mv + 0), 1.0f / dot3(mv + 4, mv + 4), 1.0f / dot3(mv + 8, mv + 8)};
    float vpos[4] = {0, 0, 0, 0}, nrm[3] = {0, 0, 0}, col[4] = {0, 0, 0, 0};
    float vp[3][4], vn[3][3], vc[3][4], vt[3][3], tng[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k) {
        mat4_mul_vec4(mv, v[k][0], v[k][1], v[k][2], 1.0f, vp[k]);
        float nm[3] = {0, 0, 0};
        if (ob->attr_off[1] != R3O_INVALID) fetch_vec3(mesh, ob->attr_off[1], idx[k], nm);
        float sn[3] = {inv_s2[0] * nm[0], inv_s2[1] * nm[1], inv_s2[2] * nm[2]};
        mat3_mul_vec3(mv + 0, mv + 4, mv + 8, sn, vn[k]);
        normalize3(vn[k]);
        { /* vs_out.tangent = normalize(mv_mat3 * (inv_scale_sq * vs_in.tangent)), opaque.wgsl:129 */
            float tg[3] = {0, 0, 0};
            if (ob->attr_off[2] != R3O_INVALID) fetch_vec3(mesh, ob->attr_off[2], idx[k], tg);
            float st[3] = {inv_s2[0] * tg[0], inv_s2[1] * tg[1], inv_s2[2] * tg[2]};
            mat3_mul_vec3(mv + 0, mv + 4, mv + 8, st, vt[k]);
            normalize3(vt[k]);
        }
        if (ob->attr_off[5] != R3O_INVALID) {
            uint32_t cw = mesh[ob->attr_off[5] / 4u + idx[k]];
            for (int c = 0; c < 4; ++c) vc[k][c] = (float)((cw >> (8 * c)) & 0xFFu) / 255.0f;
        } else
            for (int c = 0; c < 4; ++c) vc[k][c] = 1.0f;
    }
    R3O_SCOPE(0); /* attribute interpolation: the rasteriser's */
    for (int c = 0; c < 4; ++c) vpos[c] = (lam[0] * vp[0][c] + lam[1] * vp[1][c]) + lam[2] * vp[2][c];
    for (int c = 0; c < 3; ++c) nrm[c] = (lam[0] * vn[0][c] + lam[1] * vn[1][c]) + lam[2] * vn[2][c];
    for (int c = 0; c < 3; ++c) tng[c] = (lam[0] * vt[0][c] + lam[1] * vt[1][c]) + lam[2] * vt[2][c];
    for (int c = 0; c < 4; ++c) col[c] = (lam[0] * vc[0][c] + lam[1] * vc[1][c]) + lam[2] * vc[2][c];

    /* fragment stage, opaque.wgsl:203-424.  Texture slots (managers/material.rs:25-29 order): 0 albedo, 1 normal,
     * 2 roughness, 3 metallic, 4 reflectance, 5 clear coat, 6 clear coat roughness, 7 emissive, 8 anisotropy, 9 AO */
    R3O_SCOPE(2);
    pixel_data px;
    int any_tex = 0;
    for (int k = 0; k < 10; ++k) any_tex |= mat->tex[k] != 0u;
    float coords[2] = {0, 0}, ddx[2] = {0, 0}, ddy[2] = {0, 0};
    const int nearest = (mat->flags & FLAGS_NEAREST) != 0;
    if (any_tex) { /* opaque.wgsl:207-209 */
        float uv[3][2];
        for (int k = 0; k < 3; ++k) fetch_uv0(ob, mesh, idx[k], uv[k]);
        frag_coords(&ts, uv, mat->uv_transform0, (int)x, (int)y, coords, ddx, ddy);
    }
#define TEX(slot, dst) tex_sample_grad(&sc->tt, mat->tex[slot], nearest, coords[0], coords[1], ddx, ddy, dst)
    if (mat->flags & FLAGS_ALBEDO_ACTIVE) {
        for (int c = 0; c < 4; ++c) px.albedo[c] = 1.0f;
        if (mat->tex[0] != 0u) TEX(0, px.albedo);
        if (mat->flags & FLAGS_ALBEDO_BLEND) {
            if (mat->flags & FLAGS_ALBEDO_VERTEX_SRGB) {
                for (int c = 0; c < 3; ++c) px.albedo[c] *= srgb_to_linear(col[c]);
                px.albedo[3] *= col[3];
            } else
                for (int c = 0; c < 4; ++c) px.albedo[c] *= col[c];
        }
    } else {
        px.albedo[0] = px.albedo[1] = px.albedo[2] = 0.0f;
        px.albedo[3] = 1.0f;
    }
    for (int c = 0; c < 4; ++c) px.albedo[c] *= mat->albedo[c];

    if (mat->flags & FLAGS_UNLIT) {
        for (int c = 0; c < 4; ++c) out[c] = px.albedo[c];
    } else {
        /* --- normal (opaque.wgsl:246-273) */
        if (mat->tex[1] != 0u) {
            float t[4], n[3];
            TEX(1, t);
            if (mat->flags & FLAGS_BICOMPONENT_NORMAL) {
                const int sw = (mat->flags & FLAGS_SWIZZLED_NORMAL) != 0;
                float b0 = sw ? t[3] : t[0], b1 = t[1]; /* texture_read.ag : texture_read.rg */
                b0 = b0 * 2.0f - 1.0f;
                b1 = b1 * 2.0f - 1.0f;
                n[0] = b0; n[1] = b1;
                n[2] = sqrtf((1.0f - b0 * b0) - b1 * b1);
            } else {
                for (int c = 0; c < 3; ++c) n[c] = t[c] * 2.0f - 1.0f;
                normalize3(n);
            }
            if (mat->flags & FLAGS_YDOWN_NORMAL) n[1] = -n[1];
            float nn[3] = {nrm[0], nrm[1], nrm[2]}, tn[3] = {tng[0], tng[1], tng[2]};
            normalize3(nn);
            normalize3(tn);
            float bt[3] = {nn[1] * tn[2] - tn[1] * nn[2], nn[2] * tn[0] - tn[2] * nn[0], nn[0] * tn[1] - tn[0] * nn[1]};
            mat3_mul_vec3(tn, bt, nn, n, px.normal); /* tbn * normal */
        } else {
            for (int c = 0; c < 3; ++c) px.normal[c] = nrm[c];
        }
        normalize3(px.normal);
        /* --- AO, metallic, roughness (opaque.wgsl:277-351) */
        float ao = mat->ambient_occlusion, pr = mat->roughness, metallic = mat->metallic;
        if (mat->flags & FLAGS_AOMR_COMBINED) {
            if (mat->tex[2] != 0u) {
                float t[4];
                TEX(2, t);
                ao = mat->ambient_occlusion * t[0];
                pr = mat->roughness * t[1];
                metallic = mat->metallic * t[2];
            }
        } else if (mat->flags & FLAGS_AOMR_BW_SPLIT) {
            float t[4];
            if (mat->tex[2] != 0u) { TEX(2, t); pr = mat->roughness * t[0]; }
            if (mat->tex[3] != 0u) { TEX(3, t); metallic = mat->metallic * t[0]; }
            if (mat->tex[9] != 0u) { TEX(9, t); ao = mat->ambient_occlusion * t[0]; }
        } else {
            float t[4];
            if (mat->tex[2] != 0u) {
                TEX(2, t);
                int sw = (mat->flags & FLAGS_AOMR_SWIZZLED_SPLIT) != 0;
                pr = mat->roughness * (sw ? t[1] : t[0]);
                metallic = mat->metallic * (sw ? t[2] : t[1]);
            }
            if (mat->tex[9] != 0u) { TEX(9, t); ao = mat->ambient_occlusion * t[0]; }
        }
        /* --- reflectance (opaque.wgsl:355-359) */
        float reflectance = mat->reflectance;
        if (mat->tex[4] != 0u) { float t[4]; TEX(4, t); reflectance = mat->reflectance * t[0]; }
        /* --- clear coat (opaque.wgsl:363-391) */
        float cc = mat->clear_coat, ccpr = mat->clear_coat_roughness;
        if (mat->flags & FLAGS_CC_GLTF_COMBINED) {
            if (mat->tex[5] != 0u) {
                float t[4];
                TEX(5, t);
                cc = mat->clear_coat * t[0];
                ccpr = mat->clear_coat_roughness * t[1];
            }
        } else {
            float t[4];
            if (mat->tex[5] != 0u) { TEX(5, t); cc = mat->clear_coat * t[0]; }
            if (mat->tex[6] != 0u) {
                TEX(6, t);
                ccpr = mat->clear_coat_roughness * ((mat->flags & FLAGS_CC_GLTF_SPLIT) ? t[1] : t[0]);
            }
        }
        /* --- emissive (opaque.wgsl:395-399); the anisotropy texture (:403-407) feeds nothing downstream */
        for (int c = 0; c < 3; ++c) px.emissive[c] = mat->emissive[c];
        if (mat->tex[7] != 0u) {
            float t[4];
            TEX(7, t);
            for (int c = 0; c < 3; ++c) px.emissive[c] = mat->emissive[c] * t[c];
        }
#undef TEX
        for (int c = 0; c < 3; ++c) px.diffuse[c] = px.albedo[c] * (1.0f - metallic);
        float refl = (0.16f * reflectance) * reflectance;
        for (int c = 0; c < 3; ++c) px.f0[c] = px.albedo[c] * metallic + (refl * (1.0f - metallic));
        if (cc != 0.0f) {
            float base_pr = fmaxf(pr, ccpr);
            pr = pr * (1.0f - cc) + base_pr * cc;
        }
        px.roughness = pr * pr;
        px.ao = ao;

        float vv[3] = {vpos[0], vpos[1], vpos[2]};
        normalize3(vv);
        for (int c = 0; c < 3; ++c) vv[c] = -vv[c];
        float color[3] = {px.emissive[0], px.emissive[1], px.emissive[2]};
        for (uint32_t i = 0; i < n_dir; ++i) {
            float sn[4];
            mat4_mul_vec4(light_mats + 16 * i, vpos[0], vpos[1], vpos[2], vpos[3], sn);
            float fl[2] = {sn[0] * 0.5f + 0.5f, sn[1] * 0.5f + 0.5f};
            float local[2] = {fl[0], 1.0f - fl[1]};
            float tl[2] = {dir[i].atlas_offset[0], dir[i].atlas_offset[1]};
            float tr[2] = {tl[0] + dir[i].atlas_size[0], tl[1] + dir[i].atlas_size[1]};
            float coords[2] = {tl[0] * (1.0f - local[0]) + tr[0] * local[0],
                               tl[1] * (1.0f - local[1]) + tr[1] * local[1]};
            float border[2] = {dir[i].inv_resolution[0] * 1.5f, dir[i].inv_resolution[1] * 1.5f};
            tl[0] += border[0]; tl[1] += border[1];
            tr[0] -= border[0]; tr[1] -= border[1];
            float shadow = 1.0f;
            if ((fl[0] >= tl[0] || fl[1] >= tl[1]) && (fl[0] <= tr[0] || fl[1] <= tr[1]) && sn[2] >= 0.0f &&
                sn[2] <= 1.0f)
                shadow = shadow_pcf5(atlas, atlas_w, atlas_h, coords[0], coords[1], sn[2]);
            float res[3];
            surface_shading(light_l + 3 * i, dir[i].color, &px, vv, shadow * px.ao, res);
            for (int c = 0; c < 3; ++c) color[c] += res[c];
        }
        for (uint32_t i = 0; i < n_point; ++i) {
            float delta[3] = {pview[4 * i + 0] - vpos[0], pview[4 * i + 1] - vpos[1], pview[4 * i + 2] - vpos[2]};
            float d = sqrtf(dot3(delta, delta));
            float s = sat(d / point[i].radius);
            float s2 = s * s, is2 = 1.0f - s2;
            float att = is2 * is2 / (1.0f + s2);
            float inten[3] = {point[i].color[0] * att, point[i].color[1] * att, point[i].color[2] * att};
            float l[3] = {delta[0] / d, delta[1] / d, delta[2] / d};
            float res[3];
            surface_shading(l, inten, &px, vv, px.ao, res);
            for (int c = 0; c < 3; ++c) color[c] += (res[c] > 0.0f ? res[c] : 0.0f);
        }
        for (int c = 0; c < 3; ++c) out[c] = fmaxf(fu->ambient[c] * px.albedo[c], color[c]);
        out[3] = fmaxf(fu->ambient[3] * px.albedo[3], px.albedo[3]);
    }
}

/*
 * `samples` keys per pixel (pixel-major).  Every sample of the multisampled Rgba16Float target holds the
 * half-rounded colour of its nearest fragment (or the clear colour); the render pass resolve
 * (base.rs:245-258) is their box average, evaluated here as ((s0 + s1) + (s2 + s3)) * 0.25 in f32.
 *
 * Transparent pass (row N3; base.rs:451-465, pbr/routine.rs:113-118): the triangles of the blend-key objects in
 * DRAW ORDER (blend_obj / blend_tri: objects back to front, batching.rs:146-176, triangles in index order --
 * the non-atomic residual list of cull.wgsl:372-378) are rasterised after the opaque passes with depth test
 * GreaterEqual against the final opaque depth, no depth write, and BlendState::ALPHA_BLENDING
 * (rgb = src * a + dst * (1 - a), alpha = src.a + dst.a * (1 - a)) into the Rgba16Float samples: f32
 * arithmetic on the half-rounded destination, result rounded to half.  The fragment is evaluated once per
 * pixel and triangle (pixel centre) and blended into every covered sample that passes the depth test.
 */
void r3o_shade(const uint64_t *vis, uint32_t w, uint32_t h, uint32_t samples, const r3o_frame_uniforms *fu,
               const r3o_camera_header *hdr, const r3o_object *objects, const uint32_t *mesh,
               const r3o_baked *baked, const r3o_material *materials, const uint32_t *tri_base,
               uint32_t n_dir, const r3o_dir_light *dir, uint32_t n_point, const r3o_point_light *point,
               const float *atlas, uint32_t atlas_w, uint32_t atlas_h, const float *clear_color,
               const r3o_texture_desc *tdescs, uint32_t ntex, const uint32_t *texels,
               const uint32_t *blend_obj, const uint32_t *blend_tri, uint64_t n_blend, uint16_t *hdr_out) {
    init_srgb8();
    float *light_mats = (float *)malloc(sizeof(float) * 16 * (n_dir ? n_dir : 1));
    float *light_l = (float *)malloc(sizeof(float) * 3 * (n_dir ? n_dir : 1));
    float *pview = (float *)malloc(sizeof(float) * 4 * (n_point ? n_point : 1));
    for (uint32_t i = 0; i < n_dir; ++i) {
        r3o_mat4_mul(dir[i].view_proj, fu->inv_view, light_mats + 16 * i);
        float nd[3] = {-dir[i].direction[0], -dir[i].direction[1], -dir[i].direction[2]};
        mat3_mul_vec3(fu->view + 0, fu->view + 4, fu->view + 8, nd, light_l + 3 * i);
        normalize3(light_l + 3 * i);
    }
    for (uint32_t i = 0; i < n_point; ++i)
        mat4_mul_vec4(fu->view, point[i].position[0], point[i].position[1], point[i].position[2], point[i].position[3],
                      pview + 4 * i);
    shade_ctx sc = {w, h, fu, hdr, objects, mesh, baked, materials, tri_base, n_dir, dir, n_point, point,
                    atlas, atlas_w, atlas_h, light_mats, light_l, pview, {tdescs, ntex, texels}};
    /* the Rgba16Float sample values, as floats */
    float *smp = (float *)malloc(sizeof(float) * 4 * (size_t)w * h * samples);

#pragma omp parallel for schedule(dynamic, 4)
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const uint64_t pix = (uint64_t)y * w + x;
            float *half_s = smp + 4 * pix * samples;
            uint32_t last_id = 0xFFFFFFFFu;
            uint32_t last_s = 0;
            for (uint32_t sm = 0; sm < samples; ++sm) {
                uint32_t id = (uint32_t)(vis[pix * samples + sm] & 0xFFFFFFFFu);
                if (id == last_id) {  /* same triangle, same pixel centre: same value */
                    for (int c = 0; c < 4; ++c) half_s[4 * sm + c] = half_s[4 * last_s + c];
                    continue;
                }
                float out[4];
                if (id == 0u)
                    for (int c = 0; c < 4; ++c) out[c] = clear_color[c];
                else
                    shade_fragment(&sc, id, x, y, out);
                for (int c = 0; c < 4; ++c) half_s[4 * sm + c] = f16_to_f32(f32_to_f16(out[c]));
                last_id = id;
                last_s = sm;
            }
        }

    /* transparent pass, strictly in draw order */
    {
        const float(*spos)[2] = samples == 4u ? SAMPLE_POS_4 : SAMPLE_POS_1;
        float half_w = (float)w / 2.0f, half_h = (float)h / 2.0f;
        int positive_visible = (hdr->flags & PCU_POSITIVE_AREA_VISIBLE) != 0;
        for (uint64_t i = 0; i < n_blend; ++i) {
            uint32_t o = blend_obj[i], t = blend_tri[i];
            const r3o_object *ob = &objects[o];
            if (ob->enabled == 0u) continue;
            uint32_t idx[3];
            float v[3][3];
            fetch_triangle(ob, mesh, t, idx, v);
            tri_setup ts;
            setup_triangle(baked[o].model_view_proj, v, half_w, half_h, positive_visible, &ts);
            if (!ts.valid) continue;
            int x0, y0, x1, y1;
            tri_bounds(baked[o].model_view_proj, v, half_w, half_h, (int)w, (int)h, &x0, &y0, &x1, &y1);
            uint32_t id = tri_base[o] + t + 1u;
            /* one parallel region PER TRIANGLE (draw order is serial): a team as large as the box has hardware threads costs tens
             * of milliseconds to fork and join, a few thousand times per frame (tools/fuzz_parity.py: 67 s for a 226 x 220 frame on
             * the GPU box's 256 threads, 1 s on 8) -- the team is sized by the triangle's rows, small triangles run serially.
             * Scheduling only: every pixel is written by one iteration. */
            const int blend_rows = y1 - y0 + 1;
            const int blend_team = blend_rows / 16 < 1 ? 1 : (blend_rows / 16 > 16 ? 16 : blend_rows / 16);
#pragma omp parallel for schedule(dynamic, 4) num_threads(blend_team) if (blend_rows >= 32)
            for (int y = y0; y <= y1; ++y)
                for (int x = x0; x <= x1; ++x) {
                    const uint64_t pix = (uint64_t)y * w + (uint64_t)x;
                    uint32_t mask = 0;
                    for (uint32_t sm = 0; sm < samples; ++sm) {
                        float E[3];
                        if (!edge_eval(&ts, (float)x + spos[sm][0], (float)y + spos[sm][1], E)) continue;
                        float z = frag_depth_at(&ts, E, (float)x + spos[sm][0], (float)y + spos[sm][1]);
                        if (!(z >= 0.0f && z <= 1.0f)) continue;
                        uint32_t db = (uint32_t)(vis[pix * samples + sm] >> 32);
                        float dz; memcpy(&dz, &db, 4);
                        if (!(z >= dz)) continue; /* CompareFunction::GreaterEqual, depth write off */
                        mask |= 1u << sm;
                    }
                    if (!mask) continue;
                    float src[4];
                    shade_fragment(&sc, id, (uint32_t)x, (uint32_t)y, src);
                    float a = src[3];
                    for (uint32_t sm = 0; sm < samples; ++sm) {
                        if (!(mask & (1u << sm))) continue;
                        float *d = smp + 4 * (pix * samples + sm);
                        float r[4];
                        for (int c = 0; c < 3; ++c) r[c] = src[c] * a + d[c] * (1.0f - a);
                        r[3] = src[3] * 1.0f + d[3] * (1.0f - a);
                        for (int c = 0; c < 4; ++c) d[c] = f16_to_f32(f32_to_f16(r[c]));
                    }
                }
        }
    }

#pragma omp parallel for schedule(static)
    for (uint64_t pix = 0; pix < (uint64_t)w * h; ++pix) {
        const float *half_s = smp + 4 * pix * samples;
        uint16_t *o16 = hdr_out + 4 * pix;
        if (samples == 1u) {
            for (int c = 0; c < 4; ++c) o16[c] = f32_to_f16(half_s[c]);
        } else {
            for (int c = 0; c < 4; ++c)
                o16[c] = f32_to_f16(((half_s[c] + half_s[4 + c]) + (half_s[8 + c] + half_s[12 + c])) * 0.25f);
        }
    }
    free(smp);
    free(light_mats);
    free(light_l);
    free(pview);
}

/* ------------------------------------------------------------------ tonemap */
/* blit.wgsl fs_main_scene into an Rgba8UnormSrgb target (tonemapping.rs:44): exact sRGB OETF. */
static float srgb_oetf(float x) {
    if (!(x > 0.0f)) return 0.0f; /* also NaN */
    if (x >= 1.0f) return 1.0f;
    if (x <= 0.0031308f) return x * 12.92f;
    return 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
}
/* blit.wgsl fs_main_monitor (tonemapping.rs:44: targets whose format is not *Srgb): the shader applies
 * math/color.wgsl:13-19 srgb_scene_to_display -- exponent 0.4166, not 1 / 2.4 -- and the unorm store clamps. */
static float srgb_scene_to_display(float x) {
    const float e = x > 0.0031308f ? 1.055f * powf(x, 0.4166f) - 0.055f : x * 12.92f;
    if (!(e > 0.0f)) return 0.0f; /* unorm conversion: NaN and negatives -> 0 */
    return e >= 1.0f ? 1.0f : e;
}
/* output_format: 0 Rgba8UnormSrgb, 1 Bgra8UnormSrgb, 2 Rgba8Unorm, 3 Bgra8Unorm (bit 0: B and R swapped in memory,
 * bit 1: manual transfer function).  out_f32 stays in r, g, b, a order. */
void r3o_tonemap_format(const uint16_t *hdr_in, uint64_t npix, float *out_f32, uint8_t *out_u8, uint32_t output_format) {
    const int bgr = (output_format & 1u) != 0u, manual = (output_format & 2u) != 0u;
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < npix; ++i) {
        for (int c = 0; c < 4; ++c) {
            float v = f16_to_f32(hdr_in[4 * i + c]);
            float e = c < 3 ? (manual ? srgb_scene_to_display(v) : srgb_oetf(v)) : ((!(v > 0.0f)) ? 0.0f : (v >= 1.0f ? 1.0f : v));
            if (out_f32) out_f32[4 * i + c] = e;
            if (out_u8) out_u8[4 * i + ((bgr && c < 3) ? 2 - c : c)] = (uint8_t)(e * 255.0f + 0.5f);
        }
    }
}
void r3o_tonemap(const uint16_t *hdr_in, uint64_t npix, float *out_f32, uint8_t *out_u8) {
    r3o_tonemap_format(hdr_in, npix, out_f32, out_u8, 0u);
}

/* ------------------------------------------------------------------ K8: GPU skinning (row S1) */
/* GpuSkinningInput, 40 B (rend3-routine/src/skinning.rs:23-46 / skinning.wgsl:3-25). */
typedef struct {
    uint32_t base_position_offset, base_normal_offset, base_tangent_offset;
    uint32_t joint_indices_offset, joint_weight_offset;
    uint32_t updated_position_offset, updated_normal_offset, updated_tangent_offset;
    uint32_t joint_matrix_base_offset, vertex_count;
} r3o_skinning_input;

/* skinning.wgsl:37-94, one dispatch per skeleton in the reference (skinning.rs:181-198); sequential here.
 * Writes the skinned position / normal / tangent runs into the mesh buffer. */
void r3o_skinning(uint32_t *mesh, const r3o_skinning_input *inputs, uint32_t n_skeletons, const float *joint_matrices) {
    for (uint32_t s = 0; s < n_skeletons; ++s) {
        const r3o_skinning_input *in = &inputs[s];
        for (uint32_t idx = 0; idx < in->vertex_count; ++idx) {
            /* extract_attribute_vec4_u16 / vec4_f32 (vertex_attributes.wgsl:60-80) */
            uint32_t j0 = mesh[in->joint_indices_offset / 4u + idx * 2u], j1 = mesh[in->joint_indices_offset / 4u + idx * 2u + 1u];
            uint32_t ji[4] = {j0 & 0xFFFFu, (j0 >> 16) & 0xFFFFu, j1 & 0xFFFFu, (j1 >> 16) & 0xFFFFu};
            float jw[4];
            memcpy(jw, mesh + in->joint_weight_offset / 4u + idx * 4u, 16);
            float pos[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, tan[3] = {0, 0, 0};
            if (in->base_position_offset != R3O_INVALID) fetch_vec3(mesh, in->base_position_offset, idx, pos);
            if (in->base_normal_offset != R3O_INVALID) fetch_vec3(mesh, in->base_normal_offset, idx, nrm);
            if (in->base_tangent_offset != R3O_INVALID) fetch_vec3(mesh, in->base_tangent_offset, idx, tan);
            float pa[3] = {0, 0, 0}, na[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
            for (int i = 0; i < 4; ++i) {
                float w = jw[i];
                if (w > 0.0f) {
                    const float *jm = joint_matrices + 16 * (size_t)(in->joint_matrix_base_offset + ji[i]);
                    float p4[4];
                    mat4_mul_vec4(jm, pos[0], pos[1], pos[2], 1.0f, p4);
                    for (int c = 0; c < 3; ++c) pa[c] += p4[c] * w;
                    float inv_s2[3] = {1.0f / dot3(jm + 0, jm + 0), 1.0f / dot3(jm + 4, jm + 4), 1.0f / dot3(jm + 8, jm + 8)};
                    float sn[3] = {inv_s2[0] * nrm[0], inv_s2[1] * nrm[1], inv_s2[2] * nrm[2]};
                    float st[3] = {inv_s2[0] * tan[0], inv_s2[1] * tan[1], inv_s2[2] * tan[2]};
                    float rn[3], rt[3];
                    mat3_mul_vec3(jm + 0, jm + 4, jm + 8, sn, rn);
                    mat3_mul_vec3(jm + 0, jm + 4, jm + 8, st, rt);
                    for (int c = 0; c < 3; ++c) { na[c] += rn[c] * w; ta[c] += rt[c] * w; }
                }
            }
            normalize3(na);
            normalize3(ta);
            if (in->updated_position_offset != R3O_INVALID) memcpy(mesh + in->updated_position_offset / 4u + idx * 3u, pa, 12);
            if (in->updated_normal_offset != R3O_INVALID) memcpy(mesh + in->updated_normal_offset / 4u + idx * 3u, na, 12);
            if (in->updated_tangent_offset != R3O_INVALID) memcpy(mesh + in->updated_tangent_offset / 4u + idx * 3u, ta, 12);
        }
    }
}

/* The same skinning in the operation order of the matrix-core variant (rend3_amd/csrc/skin_mfma.hip, R3N_SKIN_MFMA): the f32
 * MFMA is a fused-multiply-add chain over k = 0..3, the blend weights are applied per JOINT SLOT j < 4 of the rig (not per
 * influence) and the four slots are summed in slot order.  njoints[s] <= 4.  Not the arithmetic contract of r3o_skinning: a
 * second, equally explicit order that the opt-in kernel is bit-identical to. */
void r3o_skinning_mfma_order(uint32_t *mesh, const r3o_skinning_input *inputs, uint32_t n_skeletons, const float *joint_matrices,
                             const uint32_t *njoints) {
    for (uint32_t s = 0; s < n_skeletons; ++s) {
        const r3o_skinning_input *in = &inputs[s];
        for (uint32_t idx = 0; idx < in->vertex_count; ++idx) {
            uint32_t j0 = mesh[in->joint_indices_offset / 4u + idx * 2u], j1 = mesh[in->joint_indices_offset / 4u + idx * 2u + 1u];
            uint32_t ji[4] = {j0 & 0xFFFFu, (j0 >> 16) & 0xFFFFu, j1 & 0xFFFFu, (j1 >> 16) & 0xFFFFu};
            float jw[4];
            memcpy(jw, mesh + in->joint_weight_offset / 4u + idx * 4u, 16);
            float pos[3] = {0, 0, 0}, nrm[3] = {0, 0, 0}, tan[3] = {0, 0, 0};
            if (in->base_position_offset != R3O_INVALID) fetch_vec3(mesh, in->base_position_offset, idx, pos);
            if (in->base_normal_offset != R3O_INVALID) fetch_vec3(mesh, in->base_normal_offset, idx, nrm);
            if (in->base_tangent_offset != R3O_INVALID) fetch_vec3(mesh, in->base_tangent_offset, idx, tan);
            float tp[4][3], tn[4][3], tt[4][3];
            for (uint32_t j = 0; j < 4u; ++j) {
                float W = 0.0f;
                for (int i = 0; i < 4; ++i) W += (ji[i] == j && jw[i] > 0.0f) ? jw[i] : 0.0f;
                for (int r = 0; r < 3; ++r) {
                    float qp = 0.0f, qn = 0.0f, qt = 0.0f;
                    if (j < njoints[s]) {
                        const float *jm = joint_matrices + 16 * (size_t)(in->joint_matrix_base_offset + j);
                        const float bp[4] = {pos[0], pos[1], pos[2], 1.0f};
                        for (int k = 0; k < 4; ++k) qp = fmaf(jm[4 * k + r], bp[k], qp);
                        for (int k = 0; k < 3; ++k) {
                            const float a = jm[4 * k + r] * (1.0f / dot3(jm + 4 * k, jm + 4 * k));
                            qn = fmaf(a, nrm[k], qn);
                            qt = fmaf(a, tan[k], qt);
                        }
                        qn = fmaf(0.0f, 0.0f, qn);  /* k = 3 of the instruction: a = 0, b = 0 */
                        qt = fmaf(0.0f, 0.0f, qt);
                    }
                    tp[j][r] = qp * W; tn[j][r] = qn * W; tt[j][r] = qt * W;
                }
            }
            float pa[3], na[3], ta[3];
            for (int r = 0; r < 3; ++r) {
                pa[r] = ((tp[0][r] + tp[1][r]) + tp[2][r]) + tp[3][r];
                na[r] = ((tn[0][r] + tn[1][r]) + tn[2][r]) + tn[3][r];
                ta[r] = ((tt[0][r] + tt[1][r]) + tt[2][r]) + tt[3][r];
            }
            normalize3(na);
            normalize3(ta);
            if (in->updated_position_offset != R3O_INVALID) memcpy(mesh + in->updated_position_offset / 4u + idx * 3u, pa, 12);
            if (in->updated_normal_offset != R3O_INVALID) memcpy(mesh + in->updated_normal_offset / 4u + idx * 3u, na, 12);
            if (in->updated_tangent_offset != R3O_INVALID) memcpy(mesh + in->updated_tangent_offset / 4u + idx * 3u, ta, 12);
        }
    }
}
