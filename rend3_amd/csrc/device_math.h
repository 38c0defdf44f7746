// device_math.h -- f32 helpers shared by the kernels.  Every expression is written with an explicit
// operation order and the TU is built with -ffp-contract=off, so each op rounds once (no FMA) and the
// results are bit-identical to the arithmetic contract in DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>

#include "exact_math.h"
#include "layouts.h"

#define R3N_DEV __device__ __forceinline__

// A wave-uniform value of read-only memory through the SCALAR unit (constant address space + readfirstlane'd address =>
// s_load into SGPRs): one load for the wave instead of 64 lanes' worth, no vector registers, and VALU instructions take the
// result as a scalar operand.  Only for memory no kernel of the same launch writes.
typedef uint32_t r3n_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t r3n_u32x16 __attribute__((ext_vector_type(16)));
template <class V> R3N_DEV V scalar_load(const void *p) {
    const unsigned long long v = (unsigned long long)p;
    // (the builtin returns int: through uint32_t first, or the low half sign-extends into the high one)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    const unsigned long long u = ((unsigned long long)hi << 32) | (unsigned long long)lo;
    return *reinterpret_cast<__attribute__((address_space(4))) const V *>(u);
}

// m * (x,y,z,w), column-major m: ((c0*x + c1*y) + c2*z) + c3*w
R3N_DEV void mul_vec4(const float *__restrict__ m, float x, float y, float z, float w, float o[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}
// position variant: w == 1 still multiplies (keeps the op sequence identical to the contract)
R3N_DEV void mul_point(const float *__restrict__ m, const float v[3], float o[4]) { mul_vec4(m, v[0], v[1], v[2], 1.0f, o); }

// Two floats in one register pair: `*` and `+` on this type compile to v_pk_mul_f32 / v_pk_add_f32 -- one issue slot for both
// lanes' operations, each still rounded once, so packing is invisible to the arithmetic contract.
typedef float f2 __attribute__((ext_vector_type(2)));
R3N_DEV f2 splat2(float v) { return (f2){v, v}; }

// Arithmetic policy of the shading code (kernels_shade.h, texture.h).
//   MathExact  the contract of DESIGN.md section 2: a * b + c rounds twice (the TU is built with -ffp-contract=off), division and
//              square root are the correctly rounded IEEE operations.  Bit-identical to the oracle.
//   MathFast   opt-in (r3n_config.shade_mode = R3N_SHADE_FAST): fused multiply-add, v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 (1 ulp).
//              Not bit-identical; the framebuffer stays within the north-star tolerance (1e-3 after tonemap), tested.
struct MathExact {
    static constexpr bool fast = false;
    static R3N_DEV float mad(float a, float b, float c) { return a * b + c; }
    static R3N_DEV f2 mad(f2 a, f2 b, f2 c) { return a * b + c; }
    // the single-argument operations through exact_math.h: the same correctly rounded values in a third of the instructions
    // for arguments inside a guarded exponent range (exhaustively checked on the device), the compiler's expansion otherwise
    static R3N_DEV float rcp(float x) { return exact_math::rcp(x); }
    static R3N_DEV float div(float a, float b) { return a / b; }
    static R3N_DEV float half_over(float x) { return exact_math::half_rcp(x); }  // 0.5f / x
    static R3N_DEV float sqrt(float x) { return exact_math::sqrt(x); }
    static R3N_DEV float rsqrt(float x) { return exact_math::rsqrt(x); }
};
struct MathFast {
    static constexpr bool fast = true;
    static R3N_DEV float mad(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static R3N_DEV f2 mad(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
    static R3N_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static R3N_DEV float div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
    static R3N_DEV float half_over(float x) { return 0.5f * __builtin_amdgcn_rcpf(x); }
    static R3N_DEV float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    static R3N_DEV float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
};

R3N_DEV float dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
// policy forms: same value as the plain ones under MathExact (the additions only swap commutative operands)
template <class M> R3N_DEV float dot3m(const float a[3], const float b[3]) { return M::mad(a[2], b[2], M::mad(a[1], b[1], a[0] * b[0])); }
template <class M> R3N_DEV void normalize3m(float v[3]) {
    const float r = M::rsqrt(dot3m<M>(v, v));
    v[0] *= r; v[1] *= r; v[2] *= r;
}
// m * (x,y,z,w) as two row pairs: ((c0*x + c1*y) + c2*z) + c3*w per row
template <class M> R3N_DEV void mul_vec4m(const float *__restrict__ m, float x, float y, float z, float w, float o[4]) {
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        const f2 c0 = {m[r], m[r + 1]}, c1 = {m[4 + r], m[5 + r]}, c2 = {m[8 + r], m[9 + r]}, c3 = {m[12 + r], m[13 + r]};
        const f2 t = M::mad(c3, splat2(w), M::mad(c2, splat2(z), M::mad(c1, splat2(y), c0 * splat2(x))));
        o[r] = t.x; o[r + 1] = t.y;
    }
}
template <class M> R3N_DEV void mat3_mul_vec3m(const float *__restrict__ c0, const float *__restrict__ c1, const float *__restrict__ c2,
                                               const float v[3], float o[3]) {
    const f2 t = M::mad((f2){c2[0], c2[1]}, splat2(v[2]), M::mad((f2){c1[0], c1[1]}, splat2(v[1]), (f2){c0[0], c0[1]} * splat2(v[0])));
    o[0] = t.x; o[1] = t.y;
    o[2] = M::mad(c2[2], v[2], M::mad(c1[2], v[1], c0[2] * v[0]));
}
R3N_DEV float sat(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }
R3N_DEV void normalize3(float v[3]) {
    float r = 1.0f / sqrtf(dot3(v, v));
    v[0] *= r; v[1] *= r; v[2] *= r;
}
R3N_DEV void mat3_mul_vec3(const float *__restrict__ c0, const float *__restrict__ c1, const float *__restrict__ c2,
                           const float v[3], float o[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) o[r] = (c0[r] * v[0] + c1[r] * v[1]) + c2[r] * v[2];
}

// determinant(mat3x3(p0.xyw, p1.xyw, p2.xyw)) -- cull.wgsl:272
R3N_DEV float det3_xyw(const float p0[4], const float p1[4], const float p2[4]) {
    float ax = p0[0], ay = p0[1], az = p0[3];
    float bx = p1[0], by = p1[1], bz = p1[3];
    float cx = p2[0], cy = p2[1], cz = p2[3];
    return (ax * (by * cz - cy * bz) - bx * (ay * cz - cy * az)) + cx * (ay * bz - by * az);
}

// ceil(log2(max(x,1))) from the exponent bits: exact, unlike ceil(log2f(x)) (cull.wgsl:314)
R3N_DEV uint32_t ceil_log2_ge1(float x) {
    x = fmaxf(x, 1.0f);  // NaN -> 1
    uint32_t b = __float_as_uint(x);
    uint32_t e = (b >> 23) & 0xFFu;
    if (e == 0xFFu) return 1000u;  // +inf
    uint32_t l = e - 127u;
    return (b & 0x7FFFFFu) ? l + 1u : l;
}

// float -> texel index, NaN / out of range made deterministic
R3N_DEV uint32_t clamp_texel(float v, uint32_t dim) {
    if (!(v >= 0.0f)) return 0u;
    float top = (float)(dim - 1u);
    if (v >= top) return dim - 1u;
    return (uint32_t)v;
}

R3N_DEV uint32_t mip_dim(uint32_t d, uint32_t k) {
    uint32_t v = d >> k;
    return v ? v : 1u;
}

// vertex_attributes.wgsl:51-58
// Dword-aligned groups of 2 / 3 / 4 words read with ONE load instruction (global_load_dwordx2/3/4 take 4-byte alignment on
// gfx950): the mesh buffer's vec3 attributes and index triples are 12-byte records at arbitrary word offsets, and fetched
// word by word a triangle cost twelve memory instructions instead of four.
struct __attribute__((packed, aligned(4))) r3n_words2 { uint32_t x, y; };
struct __attribute__((packed, aligned(4))) r3n_words3 { uint32_t x, y, z; };
struct __attribute__((packed, aligned(4))) r3n_words4 { uint32_t x, y, z, w; };
R3N_DEV void fetch_vec3(const uint32_t *__restrict__ mesh, uint32_t byte_off, uint32_t vtx, float o[3]) {
    const uint32_t w = byte_off / 4u + vtx * 3u;
    const r3n_words3 v = *reinterpret_cast<const r3n_words3 *>(mesh + w);
    o[0] = __uint_as_float(v.x);
    o[1] = __uint_as_float(v.y);
    o[2] = __uint_as_float(v.z);
}
// the three vertex indices of a triangle (index buffer words first .. first + 2)
R3N_DEV void fetch_indices3(const uint32_t *__restrict__ mesh, uint32_t first, uint32_t idx[3]) {
    const r3n_words3 v = *reinterpret_cast<const r3n_words3 *>(mesh + first);
    idx[0] = v.x; idx[1] = v.y; idx[2] = v.z;
}

// ---- homogeneous triangle setup (DESIGN.md "Rasteriser contract") --------------------------------
struct TriSetup {
    float e[3][3];  // oriented edge functions (A,B,C), inside >= 0
    float z[3];     // the depth plane: depth(x, y) = (z[0] * x + z[1] * y) + z[2] in viewport pixels (see setup_triangle)
    float det;      // oriented determinant (> 0 when valid)
    bool valid;
};

// p[k] = clip-space position of vertex k.  Pixel-space homogeneous coords: Xh = (x + w) * W/2,
// Yh = (w - y) * H/2 (y down), third coordinate w.  e0 = v1 x v2, e1 = v2 x v0, e2 = v0 x v1.
R3N_DEV void setup_triangle(const float p[3][4], float half_w, float half_h, bool positive_visible, TriSetup &ts) {
    float h[3][3], zc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        h[k][0] = (p[k][0] + p[k][3]) * half_w;
        h[k][1] = (p[k][3] - p[k][1]) * half_h;
        h[k][2] = p[k][3];
        zc[k] = p[k][2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float *a = h[(i + 1) % 3], *b = h[(i + 2) % 3];
        ts.e[i][0] = a[1] * b[2] - a[2] * b[1];
        ts.e[i][1] = a[2] * b[0] - a[0] * b[2];
        ts.e[i][2] = a[0] * b[1] - a[1] * b[0];
    }
    float det = (h[0][0] * ts.e[0][0] + h[0][1] * ts.e[0][1]) + h[0][2] * ts.e[0][2];
    ts.valid = positive_visible ? (det < 0.0f) : (det > 0.0f);
    if (det < 0.0f) {
        det = -det;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) ts.e[i][c] = -ts.e[i][c];
    }
    ts.det = det;
    // Depth plane (the arithmetic contract since round 4, oracle/r3o.c setup_triangle): depth is affine in window space.  With
    // every vertex in front of the eye plane: the plane through the window-space vertices (x / w, y / w, z / w), anchored at
    // vertex 0 -- a rasterised depth that reproduces the vertices' own depths is what lets a lit surface pass the reference's
    // unbiased shadow comparison against itself (profiles/r04_depth_modes.md: the homogeneous quotient sum(E_i z_i) / det left a
    // 1.6 LSB speckle against the reference's screenshots, this form 0.02).  Otherwise (a vertex at w <= 0, a degenerate
    // window-space area): the same plane from the homogeneous edge coefficients, which needs no division by w.
    bool planar = false;
    if (h[0][2] > 0.0f && h[1][2] > 0.0f && h[2][2] > 0.0f) {
        float sx[3], sy[3], zn[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float rw = exact_math::rcp(h[k][2]);  // 1 / w, correctly rounded
            sx[k] = h[k][0] * rw; sy[k] = h[k][1] * rw; zn[k] = zc[k] * rw;
        }
        const float ax = sx[1] - sx[0], ay = sy[1] - sy[0], bx = sx[2] - sx[0], by = sy[2] - sy[0];
        const float az = zn[1] - zn[0], bz = zn[2] - zn[0];
        const float ia = exact_math::rcp(ax * by - bx * ay);
        const float gx = (az * by - bz * ay) * ia, gy = (bz * ax - az * bx) * ia;
        const float c = (zn[0] - gx * sx[0]) - gy * sy[0];
        if (gx - gx == 0.0f && gy - gy == 0.0f && c - c == 0.0f) {  // finite
            ts.z[0] = gx; ts.z[1] = gy; ts.z[2] = c;
            planar = true;
        }
    }
    if (!planar) {
        ts.z[0] = ((ts.e[0][0] * zc[0] + ts.e[1][0] * zc[1]) + ts.e[2][0] * zc[2]) / det;
        ts.z[1] = ((ts.e[0][1] * zc[0] + ts.e[1][1] * zc[1]) + ts.e[2][1] * zc[2]) / det;
        ts.z[2] = ((ts.e[0][2] * zc[0] + ts.e[1][2] * zc[1]) + ts.e[2][2] * zc[2]) / det;
    }
}

// Edge functions at a pixel centre; true when covered under the top-left rule.
R3N_DEV bool f32_positive(float a) { return (int)__float_as_uint(a) > 0; }             // a > 0 for non-NaN a (a positive NaN: true)
// top-left rule as a per-edge threshold: 0 for a top / left edge, the smallest subnormal otherwise (see edge_eval)
R3N_DEV float edge_threshold(float A, float B) {
    return (A > 0.0f || (A == 0.0f && B > 0.0f)) ? 0.0f : 1.401298464324817e-45f;
}
// edge_eval with the three thresholds precomputed (the rasteriser's scan loops: once per triangle)
R3N_DEV bool edge_eval_thr(const TriSetup &ts, const float thr[3], float px, float py, float E[3]) {
    bool in = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float v = (ts.e[i][0] * px + ts.e[i][1] * py) + ts.e[i][2];
        E[i] = v;
        in = in && (v >= thr[i]);
    }
    return in;
}
R3N_DEV bool edge_eval(const TriSetup &ts, float px, float py, float E[3]) {
    bool in = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float A = ts.e[i][0], B = ts.e[i][1];
        const float v = (A * px + B * py) + ts.e[i][2];
        E[i] = v;
        // top-left rule: v > 0, or v == 0 on a top / left edge.  As ONE comparison against a per-edge threshold:
        // 0 for a top / left edge (accepts +-0), the smallest subnormal otherwise (accepts exactly v > 0; f32
        // subnormals are kept, float_denorm_mode_32 = 3).  NaN fails both forms.
        const float thr = edge_threshold(A, B);
        in = in && (v >= thr);
    }
    return in;
}

// depth at window position (px, py): the triangle's plane (setup_triangle)
R3N_DEV float frag_depth(const TriSetup &ts, float px, float py) { return (ts.z[0] * px + ts.z[1] * py) + ts.z[2]; }
// Conservative integer pixel bounds inside a (vw x vh) viewport.  Returns false when no pixel can be covered.
// Bounds only limit the scan (coverage is decided per pixel by edge_eval + the depth clip), so any conservative
// box gives identical results.  Triangles that cross the depth-clip planes (0 <= z <= w, which also implies
// w >= 0) are clipped for the purpose of the box: without this a triangle with a vertex behind the camera would
// be scanned over the whole viewport.
R3N_DEV bool tri_bounds(const float p[3][4], float half_w, float half_h, int vw, int vh, int &x0, int &y0, int &x1,
                        int &y1) {
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    bool ok = true;      // false -> fall back to the whole viewport
    bool any = false;    // some point contributes
    auto add = [&](float x, float y, float w) {
        if (!(w > 1e-30f)) { ok = false; return; }
        const float sx = (x / w + 1.0f) * half_w;
        const float sy = (1.0f - y / w) * half_h;
        if (!(sx - sx == 0.0f) || !(sy - sy == 0.0f)) { ok = false; return; }  // inf / NaN
        mnx = fminf(mnx, sx); mxx = fmaxf(mxx, sx);
        mny = fminf(mny, sy); mxy = fmaxf(mxy, sy);
        any = true;
    };
    const bool all_front = p[0][3] > 0.0f && p[1][3] > 0.0f && p[2][3] > 0.0f;
    if (all_front) {
        // projecting every vertex is conservative whether or not the depth clip removes part of the triangle
#pragma unroll
        for (int k = 0; k < 3; ++k) add(p[k][0], p[k][1], p[k][3]);
    } else {
        float dA[3], dB[3];
        bool crossA = false, crossB = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            dA[k] = p[k][2];            // z >= 0
            dB[k] = p[k][3] - p[k][2];  // z <= w
            crossA = crossA || !(dA[k] >= 0.0f);
            crossB = crossB || !(dB[k] >= 0.0f);
        }
        if (crossA && crossB) {
            ok = false;  // crosses both planes: rare, scan everything
        } else if (!crossA && !crossB) {
            ok = false;  // some w <= 0 yet inside both planes: only degenerate / NaN input
        } else {
            const float *d = crossA ? dA : dB;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int n = (k + 1) % 3;
                if (d[k] >= 0.0f) add(p[k][0], p[k][1], p[k][3]);
                if ((d[k] >= 0.0f) != (d[n] >= 0.0f)) {
                    const float t = d[k] / (d[k] - d[n]);
                    add(p[k][0] + t * (p[n][0] - p[k][0]), p[k][1] + t * (p[n][1] - p[k][1]),
                        p[k][3] + t * (p[n][3] - p[k][3]));
                }
            }
            if (ok && !any) return false;  // entirely outside the depth range: nothing can pass the depth clip
        }
    }
    if (!ok) {
        x0 = 0; y0 = 0; x1 = vw - 1; y1 = vh - 1;
        return true;
    }
    const float fx0 = floorf(mnx) - 1.0f, fy0 = floorf(mny) - 1.0f, fx1 = ceilf(mxx) + 1.0f, fy1 = ceilf(mxy) + 1.0f;
    const float wm = (float)(vw - 1), hm = (float)(vh - 1);
    x0 = fx0 < 0.0f ? 0 : (fx0 > wm ? vw : (int)fx0);
    y0 = fy0 < 0.0f ? 0 : (fy0 > hm ? vh : (int)fy0);
    x1 = fx1 < 0.0f ? -1 : (fx1 > wm ? vw - 1 : (int)fx1);
    y1 = fy1 < 0.0f ? -1 : (fy1 > hm ? vh - 1 : (int)fy1);
    return x1 >= x0 && y1 >= y0;
}
