// texture.h -- material textures (row N2, first slice: albedo), device side.
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
    const float *srgb8_to_linear; // 256 entries
};

R3N_DEV uint32_t tex_mip_dim(uint32_t d, uint32_t k) {
    const uint32_t v = d >> k;
    return v ? v : 1u;
}
R3N_DEV uint32_t tex_wrap(float f, uint32_t n) {  // f = floor(coordinate); Repeat; NaN / huge -> texel 0
    const int i = (f == f && fabsf(f) < 1e9f) ? (int)f : 0;
    if ((uint32_t)i < n) return (uint32_t)i;
    const long long w = (long long)n;
    return (uint32_t)((((long long)i % w) + w) % w);
}
R3N_DEV void tex_fetch(const TextureArgs &t, const r3n_texture_desc32 &d, uint32_t mip, uint32_t x, uint32_t y, float o[4]) {
    size_t off = d.offset;
    for (uint32_t k = 0; k < mip; ++k) off += (size_t)tex_mip_dim(d.width, k) * tex_mip_dim(d.height, k);
    const uint32_t v = t.texels[off + (size_t)y * tex_mip_dim(d.width, mip) + x];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t b = (v >> (8 * c)) & 0xFFu;
        o[c] = (d.format == 1u && c < 3) ? t.srgb8_to_linear[b] : (float)b / 255.0f;
    }
}
R3N_DEV void tex_bilinear(const TextureArgs &t, const r3n_texture_desc32 &d, uint32_t mip, float u, float v, float o[4]) {
    const uint32_t w = tex_mip_dim(d.width, mip), h = tex_mip_dim(d.height, mip);
    const float tx = u * (float)w - 0.5f, ty = v * (float)h - 0.5f;
    const float fx0 = floorf(tx), fy0 = floorf(ty);
    float fx = tx - fx0, fy = ty - fy0;
    if (!(fx == fx)) fx = 0.0f;
    if (!(fy == fy)) fy = 0.0f;
    const uint32_t x0 = tex_wrap(fx0, w), x1 = tex_wrap(fx0 + 1.0f, w);
    const uint32_t y0 = tex_wrap(fy0, h), y1 = tex_wrap(fy0 + 1.0f, h);
    float c00[4], c10[4], c01[4], c11[4];
    tex_fetch(t, d, mip, x0, y0, c00); tex_fetch(t, d, mip, x1, y0, c10);
    tex_fetch(t, d, mip, x0, y1, c01); tex_fetch(t, d, mip, x1, y1, c11);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float top = c00[c] * (1.0f - fx) + c10[c] * fx;
        const float bot = c01[c] * (1.0f - fx) + c11[c] * fx;
        o[c] = top * (1.0f - fy) + bot * fy;
    }
}
R3N_DEV void tex_nearest(const TextureArgs &t, const r3n_texture_desc32 &d, uint32_t mip, float u, float v, float o[4]) {
    const uint32_t w = tex_mip_dim(d.width, mip), h = tex_mip_dim(d.height, mip);
    tex_fetch(t, d, mip, tex_wrap(floorf(u * (float)w), w), tex_wrap(floorf(v * (float)h), h), o);
}
// textureSampleGrad(textures[id - 1], nearest ? nearest_sampler : primary_sampler, (u, v), ddx, ddy)
R3N_DEV void tex_sample_grad(const TextureArgs &t, uint32_t id, bool nearest, float u, float v, const float ddx[2],
                             const float ddy[2], float o[4]) {
    if (id == 0u || id > t.count) { o[0] = o[1] = o[2] = o[3] = 0.0f; return; }
    const r3n_texture_desc32 d = t.descs[id - 1u];
    const float W = (float)d.width, H = (float)d.height;
    const float ax = ddx[0] * W, ay = ddx[1] * H, bx = ddy[0] * W, by = ddy[1] * H;
    const float rho = fmaxf(sqrtf(ax * ax + ay * ay), sqrtf(bx * bx + by * by));
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
        tex_nearest(t, d, level, u, v, o);
        return;
    }
    tex_bilinear(t, d, level, u, v, o);
    if (frac > 0.0f) {
        float hi[4];
        tex_bilinear(t, d, level + 1u, u, v, hi);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = o[c] * (1.0f - frac) + hi[c] * frac;
    }
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
// m = uv_transform0 or nullptr (depth.wgsl uses the raw coordinates).
R3N_DEV void frag_coords(const TriSetup &ts, const float uv[3][2], const float *m, int x, int y, float coords[2],
                         float ddx[2], float ddy[2]) {
    const int xq = x & ~1, yq = y & ~1;
    float c[4][2];
    const int pts[4][2] = {{xq, y}, {xq + 1, y}, {x, yq}, {x, yq + 1}};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float raw[2];
        interp_vec2(ts, uv, pts[k][0], pts[k][1], raw);
        if (m) uv_transform(m, raw, c[k]); else { c[k][0] = raw[0]; c[k][1] = raw[1]; }
    }
    const int self = (x & 1) ? 1 : 0;
    coords[0] = c[self][0]; coords[1] = c[self][1];
#pragma unroll
    for (int k = 0; k < 2; ++k) { ddx[k] = c[1][k] - c[0][k]; ddy[k] = c[3][k] - c[2][k]; }
}
