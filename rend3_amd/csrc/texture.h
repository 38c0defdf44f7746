// texture.h -- material textures (row N2), device side: the sampler over the decoded RGBA8 texel pool.
//
// Reference behaviour restated (file:line):
//   rend3/src/managers/texture.rs            one bindless array of every 2D texture of the world
//   rend3-routine/src/common/samplers.rs:22-56  `linear` (mag = min = mipmap = Linear) and `nearest` (all Nearest)
//                                            samplers, AddressMode::Repeat, anisotropy_clamp 1, LOD clamp [0, 100]
//   opaque.wgsl:151-160,207-215              textureSampleGrad(textures[id - 1], s, coords, dpdx(coords), dpdy(coords))
//   depth.wgsl:108-118                       the shadow / depth cutout test (its quirks are reproduced by the caller)
//
// Contract (identical in oracle/r3o.c): RGBA8 texels, c / 255, sRGB decoded per texel BEFORE filtering through a
// 256-entry table of the exact formula; bilinear footprint u * w - 0.5 / floor / Repeat; level of detail from
// rho = max(|ddx * size|, |ddy * size|) = m * 2^e as e + (m - 1) -- exponent exact, mantissa as the fraction, no
// transcendental call, so CPU and GPU agree bit for bit; rho <= 1 or NaN -> level 0; linear mixes two levels,
// nearest takes the level nearest to the LOD (ties up).
#pragma once
#include "device_math.h"

struct TextureArgs {
    const r3n_texture_desc32 *descs;
    uint32_t count;
    const uint32_t *texels;       // RGBA8, every texture's mips contiguous
    const float *decode;          // 512 entries: [0, 256) = c / 255 (unorm), [256, 512) = sRGB8 -> linear, built on the
                                  // host with libm like the oracle's.  Both hold exactly what the per-texel expressions
                                  // give.  The resolve stages them in LDS: a texel decode is then 4 LDS reads instead of
                                  // 4 IEEE divisions.  (Measured alternative: a pre-decoded float4 pool, 4x the memory,
                                  // same speed -- the kernel is bound by VALU work, not by the decode.)
};

R3N_DEV uint32_t tex_mip_dim(uint32_t d, uint32_t k) {
    const uint32_t v = d >> k;
    return v ? v : 1u;
}
R3N_DEV uint32_t tex_wrap(float f, uint32_t n) {  // f = floor(coordinate); Repeat; NaN / huge -> texel 0
    const int i = (f == f && fabsf(f) < 1e9f) ? (int)f : 0;
    if ((uint32_t)i < n) return (uint32_t)i;
    // tiled coordinates leave [0, n) all the time: keep this path cheap.  Power-of-two extents (the common case) wrap
    // with a mask -- two's complement makes that the floor-modulo for negative indices too; other extents (<= 65535) use
    // a 32-bit remainder.
    if ((n & (n - 1u)) == 0u) return (uint32_t)i & (n - 1u);
    const int m = i % (int)n;
    return (uint32_t)(m < 0 ? m + (int)n : m);
}
// Everything about one sample that depends only on the texture's extent / mip count, the coordinates and the gradients
// -- not on its texels: the level(s), the blend fraction and the bilinear footprints.
struct TexFootprint {
    uint32_t width, height, mips;  // key
    bool nearest;
    uint32_t level;                // first (or only) level
    float frac;                    // weight of level + 1 (0: single level)
    uint32_t level_off;            // first texel of `level`, relative to the texture's offset
    struct Lvl {
        uint32_t w, i00, i10, i01, i11;  // texel indices inside the level (row-major)
        float fx, fy;
    } l[2];
};
R3N_DEV void tex_level_footprint(uint32_t w, uint32_t h, float u, float v, TexFootprint::Lvl &l) {
    const float tx = u * (float)w - 0.5f, ty = v * (float)h - 0.5f;
    const float fx0 = floorf(tx), fy0 = floorf(ty);
    float fx = tx - fx0, fy = ty - fy0;
    if (!(fx == fx)) fx = 0.0f;
    if (!(fy == fy)) fy = 0.0f;
    const uint32_t x0 = tex_wrap(fx0, w), x1 = tex_wrap(fx0 + 1.0f, w);
    const uint32_t y0 = tex_wrap(fy0, h), y1 = tex_wrap(fy0 + 1.0f, h);
    l.w = w;
    l.i00 = y0 * w + x0; l.i10 = y0 * w + x1; l.i01 = y1 * w + x0; l.i11 = y1 * w + x1;
    l.fx = fx; l.fy = fy;
}
R3N_DEV void tex_footprint(const r3n_texture_desc32 &d, bool nearest, float u, float v, const float ddx[2], const float ddy[2],
                           TexFootprint &f) {
    f.width = d.width; f.height = d.height; f.mips = d.mips; f.nearest = nearest;
    const float W = (float)d.width, H = (float)d.height;
    const float ax = ddx[0] * W, ay = ddx[1] * H, bx = ddy[0] * W, by = ddy[1] * H;
    // max(sqrt(a), sqrt(b)) == sqrt(max(a, b)): correctly rounded sqrt is monotone
    const float rho = sqrtf(fmaxf(ax * ax + ay * ay, bx * bx + by * by));
    uint32_t level = 0;
    float frac = 0.0f;
    if (rho > 1.0f && rho < INFINITY) {
        const uint32_t bits = __float_as_uint(rho);
        level = (bits >> 23) - 127u;
        frac = (float)(bits & 0x7FFFFFu) / 8388608.0f;
    } else if (rho == INFINITY) {
        level = d.mips;  // clamped below
    }
    if (level >= d.mips - 1u) { level = d.mips - 1u; frac = 0.0f; }
    if (nearest) {
        if (frac >= 0.5f) level += 1u;  // level + 1 <= mips - 1 here
        frac = 0.0f;
    }
    uint32_t off = 0;  // the chain is contiguous (a whole chain is < 2^32 texels: extents <= 65535)
    for (uint32_t k = 0; k < level; ++k) off += tex_mip_dim(d.width, k) * tex_mip_dim(d.height, k);
    f.level = level; f.frac = frac; f.level_off = off;
    const uint32_t w = tex_mip_dim(d.width, level), h = tex_mip_dim(d.height, level);
    if (nearest) {
        f.l[0].w = w;
        f.l[0].i00 = tex_wrap(floorf(v * (float)h), h) * w + tex_wrap(floorf(u * (float)w), w);
        return;
    }
    tex_level_footprint(w, h, u, v, f.l[0]);
    if (frac > 0.0f) tex_level_footprint(tex_mip_dim(d.width, level + 1u), tex_mip_dim(d.height, level + 1u), u, v, f.l[1]);
}
R3N_DEV void tex_texel(const TextureArgs &t, bool srgb, const uint32_t *__restrict__ lvl, uint32_t i, float o[4]) {
    const uint32_t v = lvl[i];
    const float *rgb = t.decode + (srgb ? 256 : 0);
    o[0] = rgb[v & 0xFFu]; o[1] = rgb[(v >> 8) & 0xFFu]; o[2] = rgb[(v >> 16) & 0xFFu]; o[3] = t.decode[v >> 24];
}
R3N_DEV void tex_bilinear(const TextureArgs &t, bool srgb, size_t lvl_off, const TexFootprint::Lvl &l, float o[4]) {
    float c00[4], c10[4], c01[4], c11[4];
    const uint32_t *lvl = t.texels + lvl_off;
    tex_texel(t, srgb, lvl, l.i00, c00); tex_texel(t, srgb, lvl, l.i10, c10);
    tex_texel(t, srgb, lvl, l.i01, c01); tex_texel(t, srgb, lvl, l.i11, c11);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float top = c00[c] * (1.0f - l.fx) + c10[c] * l.fx;
        const float bot = c01[c] * (1.0f - l.fx) + c11[c] * l.fx;
        o[c] = top * (1.0f - l.fy) + bot * l.fy;
    }
}
R3N_DEV void tex_apply(const TextureArgs &t, const r3n_texture_desc32 &d, const TexFootprint &f, float o[4]) {
    const size_t lvl = (size_t)d.offset + f.level_off;
    const bool srgb = d.format == 1u;
    if (f.nearest) {
        tex_texel(t, srgb, t.texels + lvl, f.l[0].i00, o);
        return;
    }
    tex_bilinear(t, srgb, lvl, f.l[0], o);
    if (f.frac > 0.0f) {
        float hi[4];
        tex_bilinear(t, srgb, lvl + (size_t)f.l[0].w * tex_mip_dim(d.height, f.level), f.l[1], hi);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = o[c] * (1.0f - f.frac) + hi[c] * f.frac;
    }
}
// textureSampleGrad(textures[id - 1], nearest ? nearest_sampler : primary_sampler, (u, v), ddx, ddy).
// (Sharing one footprint between the maps of a material that have the same extent was measured: slower -- the cached
// footprint stays live across the whole fragment stage and costs an occupancy step.)
R3N_DEV void tex_sample_grad(const TextureArgs &t, uint32_t id, bool nearest, float u, float v, const float ddx[2],
                             const float ddy[2], float o[4]) {
    if (id == 0u || id > t.count) { o[0] = o[1] = o[2] = o[3] = 0.0f; return; }
    const r3n_texture_desc32 d = t.descs[id - 1u];
    TexFootprint f;
    tex_footprint(d, nearest, u, v, ddx, ddy, f);
    tex_apply(t, d, f, o);
}

// vertex_attributes.wgsl: vec2<f32> texture coordinates (attribute 3); a missing attribute reads (0, 0)
R3N_DEV void fetch_uv0(const uint32_t *__restrict__ mesh, uint32_t byte_off, uint32_t vtx, float o[2]) {
    if (byte_off == R3N_INVALID) { o[0] = o[1] = 0.0f; return; }
    const uint32_t w = byte_off / 4u + vtx * 2u;
    o[0] = __uint_as_float(mesh[w]);
    o[1] = __uint_as_float(mesh[w + 1u]);
}
// perspective-correct interpolation of a vec2 attribute at the centre of pixel (px, py), covered or not
R3N_DEV void interp_vec2(const TriSetup &ts, const float a[3][2], int px, int py, float o[2]) {
    float E[3];
    (void)edge_eval(ts, (float)px + 0.5f, (float)py + 0.5f, E);
    const float rs = 1.0f / ((E[0] + E[1]) + E[2]);
    const float l0 = E[0] * rs, l1 = E[1] * rs, l2 = E[2] * rs;
#pragma unroll
    for (int c = 0; c < 2; ++c) o[c] = (l0 * a[0][c] + l1 * a[1][c]) + l2 * a[2][c];
}
// (uv_transform * vec3(uv, 1)).xy; mat3x3 stored as three padded vec4 columns
R3N_DEV void uv_transform(const float *m, const float uv[2], float o[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) o[c] = (m[c] * uv[0] + m[4 + c] * uv[1]) + m[8 + c] * 1.0f;
}
// Fragment-stage texture coordinates of pixel (x, y) and their derivatives: differences inside the pixel's 2x2 quad
// ("fine": same row for dpdx, same column for dpdy), every operand evaluated like its own (helper) invocation.
// m = uv_transform0 or nullptr (depth.wgsl uses the raw coordinates).  Only the two quad neighbours are evaluated
// here; the pixel's own value comes from `self_raw` (its interpolated coordinates) when the caller already has them.
R3N_DEV void frag_coords(const TriSetup &ts, const float uv[3][2], const float *m, int x, int y, float coords[2],
                         float ddx[2], float ddy[2], const float *self_raw = nullptr) {
    float raw[2], self[2], nx[2], ny[2];
    if (self_raw) { raw[0] = self_raw[0]; raw[1] = self_raw[1]; } else interp_vec2(ts, uv, x, y, raw);
    if (m) uv_transform(m, raw, self); else { self[0] = raw[0]; self[1] = raw[1]; }
    interp_vec2(ts, uv, x ^ 1, y, raw);
    if (m) uv_transform(m, raw, nx); else { nx[0] = raw[0]; nx[1] = raw[1]; }
    interp_vec2(ts, uv, x, y ^ 1, raw);
    if (m) uv_transform(m, raw, ny); else { ny[0] = raw[0]; ny[1] = raw[1]; }
    coords[0] = self[0]; coords[1] = self[1];
    // value at the odd pixel of the pair minus value at the even one
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ddx[k] = (x & 1) ? self[k] - nx[k] : nx[k] - self[k];
        ddy[k] = (y & 1) ? self[k] - ny[k] : ny[k] - self[k];
    }
}
