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
// 256-entry table of the exact formula (textures of the float-decoded formats, R3N_POOL_FLOAT: four f32 per texel, used as
// they are); bilinear footprint u * w - 0.5 / floor / Repeat; level of detail from
// rho = max(|ddx * size|, |ddy * size|) = m * 2^e as e + (m - 1) -- exponent exact, mantissa as the fraction, no
// transcendental call, so CPU and GPU agree bit for bit; rho <= 1 or NaN -> level 0; linear mixes two levels,
// nearest takes the level nearest to the LOD (ties up).
#pragma once
#include "device_math.h"

// Measured and NOT kept (round 4, profiles/r04_summary.md section 1; code in the history, commit 79bedf3): short-path samplers that
// branch on the texture's encoding to fold the decode table's offset into the LDS read (resolve 439 vs 436 us without the branch);
// RGBA8 texels decoded ONCE at upload into a float4 pool (4x the memory, -10 % vector instructions, 437.8 vs 438.2 us).

struct TextureArgs {
    const r3n_texture_desc32 *descs;
    uint32_t count;
    const uint32_t *texels;       // the pool: every texture's mips contiguous; one word per RGBA8 texel, four per float texel
    const float *decode;          // 512 entries: [0, 256) = c / 255 (unorm), [256, 512) = sRGB8 -> linear, built on the
                                  // host with libm like the oracle's.  Both hold exactly what the per-texel expressions
                                  // give.  The resolve stages them in LDS: a texel decode is then 4 LDS reads instead of
                                  // 4 IEEE divisions.  (Measured alternative: a pre-decoded float4 pool, 4x the memory,
                                  // same speed -- the kernel is bound by VALU work, not by the decode.)
    const uint32_t *level_off;    // R3N_TEX_LEVELS entries per texture: pool index of the first texel of each level
    uint32_t small_pool;          // 1: the pool holds <= 2^30 texels, so a texel's BYTE offset fits 32 bits
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
// First texel of level L of a POWER-OF-TWO texture's contiguous chain, relative to the texture's first texel:
// sum over k < L of max(W >> k, 1) * max(H >> k, 1) -- what r3n.hip's level-offset table holds (minus the texture's offset), in closed
// form: the geometric part (both extents still halving) is (4 W H - 4 W H / 4^L1) / 3, an exact division done as a multiplication by
// 3's inverse modulo 2^32; a non-square texture's tail (one extent stuck at 1) and the 1x1 levels behind it are added on a rarely
// taken path.  A dozen integer instructions instead of a dependent load: on the short path of the sampler the level offsets were one
// of the round trips in a chain of them (descriptor -> level offset -> texels).  W * H * 4 / 3 < 2^31 (the short path's pool holds at
// most 2^30 texels), L <= 15.
R3N_DEV uint32_t tex_level_start_pow2(uint32_t W, uint32_t H, uint32_t L) {
    const uint32_t a = 31u - (uint32_t)__builtin_clz(W | 1u), b = 31u - (uint32_t)__builtin_clz(H | 1u);  // (an extent of 0 counts as 1, like tex_mip_dim)
    const uint32_t m = a < b ? a : b, M = a < b ? b : a;
    const uint32_t L1 = L < m + 1u ? L : m + 1u;
    const uint32_t top = a + b + 2u;
    uint32_t off = ((1u << top) - (1u << (top - 2u * L1))) * 0xAAAAAAABu;
    if (L > m + 1u) {
        const uint32_t L2 = L < M + 1u ? L : M + 1u;
        off += (1u << (M - m)) - (1u << (M + 1u - L2));
        if (L > M + 1u) off += L - (M + 1u);
    }
    return off;
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
    const float rho = exact_math::sqrt(fmaxf(ax * ax + ay * ay, bx * bx + by * by));
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
// A decoded texel / filtered sample as two channel pairs: the bilinear and trilinear blends below then run as packed f32
// operations (v_pk_mul_f32 / v_pk_add_f32: two channels per issue slot, each operation rounded once like its scalar form).
struct Texel4 {
    f2 rg, ba;
};
// NEED_A = false: the caller reads r, g, b only (AO / roughness / metallic, emissive ...): alpha is neither decoded nor filtered
template <bool NEED_A>
R3N_DEV Texel4 tex_texel(const TextureArgs &t, uint32_t fmt, const uint32_t *__restrict__ lvl, uint32_t i) {
    if (fmt == R3N_POOL_FLOAT) {  // 16-byte aligned: textures start on 4-word boundaries, levels are whole float4s
        const float4 v = reinterpret_cast<const float4 *>(lvl)[i];
        return Texel4{(f2){v.x, v.y}, (f2){v.z, NEED_A ? v.w : 0.0f}};
    }
    const uint32_t v = lvl[i];
    const float *rgb = t.decode + (fmt == 1u ? 256 : 0);
    Texel4 o;
    o.rg = (f2){rgb[v & 0xFFu], rgb[(v >> 8) & 0xFFu]};
    o.ba = (f2){rgb[(v >> 16) & 0xFFu], NEED_A ? t.decode[v >> 24] : 0.0f};
    return o;
}
template <class M> R3N_DEV f2 mix2(f2 a, f2 b, float t, float one_minus_t) {  // a * (1 - t) + b * t
    return M::mad(b, splat2(t), a * splat2(one_minus_t));
}
template <class M, bool NEED_A>
R3N_DEV Texel4 tex_bilinear(const TextureArgs &t, uint32_t fmt, size_t lvl_off, const TexFootprint::Lvl &l) {
    const uint32_t *lvl = t.texels + lvl_off;
    const Texel4 c00 = tex_texel<NEED_A>(t, fmt, lvl, l.i00), c10 = tex_texel<NEED_A>(t, fmt, lvl, l.i10);
    const Texel4 c01 = tex_texel<NEED_A>(t, fmt, lvl, l.i01), c11 = tex_texel<NEED_A>(t, fmt, lvl, l.i11);
    const float omx = 1.0f - l.fx, omy = 1.0f - l.fy;
    Texel4 o;
    o.rg = mix2<M>(mix2<M>(c00.rg, c10.rg, l.fx, omx), mix2<M>(c01.rg, c11.rg, l.fx, omx), l.fy, omy);
    if (NEED_A) {
        o.ba = mix2<M>(mix2<M>(c00.ba, c10.ba, l.fx, omx), mix2<M>(c01.ba, c11.ba, l.fx, omx), l.fy, omy);
    } else {  // blue alone
        const float top = M::mad(c10.ba.x, l.fx, c00.ba.x * omx), bot = M::mad(c11.ba.x, l.fx, c01.ba.x * omx);
        o.ba = (f2){M::mad(bot, l.fy, top * omy), 0.0f};
    }
    return o;
}
template <class M, bool NEED_A>
R3N_DEV Texel4 tex_apply(const TextureArgs &t, const r3n_texture_desc32 &d, const TexFootprint &f) {
    const uint32_t fmt = d.format, words = fmt == R3N_POOL_FLOAT ? 4u : 1u;  // pool words per texel
    const size_t lvl = (size_t)d.offset + (size_t)f.level_off * words;
    if (f.nearest) return tex_texel<NEED_A>(t, fmt, t.texels + lvl, f.l[0].i00);
    Texel4 o = tex_bilinear<M, NEED_A>(t, fmt, lvl, f.l[0]);
    if (f.frac > 0.0f) {
        const Texel4 hi = tex_bilinear<M, NEED_A>(t, fmt, lvl + (size_t)f.l[0].w * tex_mip_dim(d.height, f.level) * words, f.l[1]);
        const float omf = 1.0f - f.frac;
        o.rg = mix2<M>(o.rg, hi.rg, f.frac, omf);
        if (NEED_A) o.ba = mix2<M>(o.ba, hi.ba, f.frac, omf);
        else o.ba.x = M::mad(hi.ba.x, f.frac, o.ba.x * omf);
    }
    return o;
}
// One level of the short path: bilinear footprint of a power-of-two level whose coordinates are tame.  Returns false when
// the general code has to take over (a floor beyond 2^24 or NaN: then `floor + 1` and the integer increment part ways).
struct TexLvlFast {
    uint32_t o00, o10, o01, o11;  // BYTE offsets of the four texels in the pool
    float fx, fy;
};
// TOTAL: instead of reporting wild coordinates the function handles them itself -- the footprint by the general formulas
// (tex_level_footprint: `floor + 1` in float, NaN / huge -> texel 0), as a rarely taken block that only rewrites `l`.  The callers
// of that form need no general path behind them.
template <bool TOTAL = false>
R3N_DEV bool tex_level_fast(uint32_t w, uint32_t h, uint32_t base, float u, float v, TexLvlFast &l) {
    const float tx = u * (float)w - 0.5f, ty = v * (float)h - 0.5f;
    const float fx0 = floorf(tx), fy0 = floorf(ty);
    l.fx = tx - fx0; l.fy = ty - fy0;
    const int ix = (int)fx0, iy = (int)fy0;
    // power-of-two extent: the mask is the floor-modulo of Repeat addressing, for negative indices too
    const uint32_t x0 = (uint32_t)ix & (w - 1u), x1 = (uint32_t)(ix + 1) & (w - 1u);
    const uint32_t y0 = __umul24((uint32_t)iy & (h - 1u), w), y1 = __umul24((uint32_t)(iy + 1) & (h - 1u), w);  // extents <= 65535
    l.o00 = (base + y0 + x0) << 2; l.o10 = (base + y0 + x1) << 2;
    l.o01 = (base + y1 + x0) << 2; l.o11 = (base + y1 + x1) << 2;
    const bool tame = fabsf(fx0) < 16777216.0f && fabsf(fy0) < 16777216.0f;  // NaN compares false
    if (TOTAL && !tame) {
        TexFootprint::Lvl g;
        tex_level_footprint(w, h, u, v, g);
        l.o00 = (base + g.i00) << 2; l.o10 = (base + g.i10) << 2;
        l.o01 = (base + g.i01) << 2; l.o11 = (base + g.i11) << 2;
        l.fx = g.fx; l.fy = g.fy;
        return true;
    }
    return tame;
}
// The same footprint with the texel indices RELATIVE to the level's first texel: what two maps of one material share when their
// levels have the same extents (tex_sample_grad's `share`).  Total like tex_level_fast<true>.
struct TexLvlRel {
    uint32_t r00, r10, r01, r11;
    float fx, fy;
};
R3N_DEV void tex_level_rel(uint32_t w, uint32_t h, float u, float v, TexLvlRel &l) {
    TexLvlFast f;
    (void)tex_level_fast<true>(w, h, 0u, u, v, f);
    l.r00 = f.o00 >> 2; l.r10 = f.o10 >> 2; l.r01 = f.o01 >> 2; l.r11 = f.o11 >> 2;
    l.fx = f.fx; l.fy = f.fy;
}
// What a sample of the short path derives from (extent, mip count, coordinates, gradients) alone.  A material's maps are sampled at
// the same coordinates with the same gradients (opaque.wgsl:207-215), so a second map of the same extent -- or of HALF the extent
// with one level less, whose levels are the first map's levels from 1 on -- has the same footprints and the same blend fraction:
// with W' = W / 2 every product of the level-of-detail arithmetic is exactly halved (powers of two; the squares of operands small
// enough to round differently vanish in both), the correctly rounded sqrt of a quarter is the half, so rho' = rho / 2 bit for bit:
// level' = level - 1 for level >= 1, the same mantissa, the same clamp (level' >= mips' - 1 <=> level >= mips - 1).
struct TexShare {
    uint32_t width, height, mips;  // of the map the entry was computed for; width == 0: empty
    uint32_t level;
    float frac;
    TexLvlRel l0, l1;              // l1 only when frac > 0
};
// SRGB_SEL: 0 / 1 = the colour channels' decode table is known here (the unorm table at t.decode, the sRGB one 256 entries
// further): its offset then folds into the LDS read's immediate and a channel's address is ONE instruction (a byte-select shift);
// -1 = per-lane table pointer `rgb` (an extract and a shift-or per channel).
template <bool NEED_A, int SRGB_SEL = -1>
R3N_DEV Texel4 tex_texel_at(const TextureArgs &t, const float *__restrict__ rgb, uint32_t byte_off) {
    const uint32_t v = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(t.texels) + byte_off);  // uniform base + 32-bit offset
    const float *tab = SRGB_SEL < 0 ? rgb : t.decode + (SRGB_SEL == 1 ? 256 : 0);
    Texel4 o;
    o.rg = (f2){tab[v & 0xFFu], tab[(v >> 8) & 0xFFu]};
    o.ba = (f2){tab[(v >> 16) & 0xFFu], NEED_A ? t.decode[v >> 24] : 0.0f};
    return o;
}
template <class M, bool NEED_A, int SRGB_SEL = -1>
R3N_DEV Texel4 tex_bilinear_fast(const TextureArgs &t, const float *__restrict__ rgb, const TexLvlFast &l) {
    const Texel4 c00 = tex_texel_at<NEED_A, SRGB_SEL>(t, rgb, l.o00), c10 = tex_texel_at<NEED_A, SRGB_SEL>(t, rgb, l.o10);
    const Texel4 c01 = tex_texel_at<NEED_A, SRGB_SEL>(t, rgb, l.o01), c11 = tex_texel_at<NEED_A, SRGB_SEL>(t, rgb, l.o11);
    const float omx = 1.0f - l.fx, omy = 1.0f - l.fy;
    Texel4 o;
    o.rg = mix2<M>(mix2<M>(c00.rg, c10.rg, l.fx, omx), mix2<M>(c01.rg, c11.rg, l.fx, omx), l.fy, omy);
    if (NEED_A) {
        o.ba = mix2<M>(mix2<M>(c00.ba, c10.ba, l.fx, omx), mix2<M>(c01.ba, c11.ba, l.fx, omx), l.fy, omy);
    } else {
        const float top = M::mad(c10.ba.x, l.fx, c00.ba.x * omx), bot = M::mad(c11.ba.x, l.fx, c01.ba.x * omx);
        o.ba = (f2){M::mad(bot, l.fy, top * omy), 0.0f};
    }
    return o;
}

// textureSampleGrad(textures[id - 1], nearest ? nearest_sampler : primary_sampler, (u, v), ddx, ddy).
// The level of detail and the footprint are the same exact arithmetic under both policies (a fused operation must not move a
// sample to another mip level or texel); M only governs the blends.
// Short path (what scanned material sets hit): linear sampler, RGBA8 texels, power-of-two extents, tame coordinates, pool below
// 4 GiB.  Same
// values as the general path -- masks instead of remainders, the level's first texel from a table instead of a walk over
// the chain, 32-bit byte offsets from the uniform pool pointer -- with one branch per sample instead of one per texel.
// (Sharing one footprint between the maps of a material that have the same extent was measured: slower -- the cached
// footprint stays live across the whole fragment stage and costs an occupancy step.)
// SHORT_ONLY: the caller knows that the short path's conditions hold for this texture (the resolve's material classes,
// kernels_shade.h R3N_FEAT_TEX_GENERAL): no general path is instantiated.
template <class M = MathExact, bool NEED_A = true, bool SHORT_ONLY = false, bool SHARE = false>
R3N_DEV void tex_sample_grad(const TextureArgs &t, uint32_t id, bool nearest, float u, float v, const float ddx[2],
                             const float ddy[2], float o[4], TexShare *share = nullptr) {
    if (id == 0u || (!SHORT_ONLY && id > t.count)) { o[0] = o[1] = o[2] = o[3] = 0.0f; return; }
    const r3n_texture_desc32 d = t.descs[id - 1u];
    if (SHORT_ONLY && SHARE) {
        // 0: same extents and mip count as the entry; 1: half the extents, one level less, and the entry's level >= 1
        const bool same = share->width == d.width && share->height == d.height && share->mips == d.mips;
        const bool half = share->width == d.width << 1 && share->height == d.height << 1 && share->mips == d.mips + 1u && share->level >= 1u && d.width != 0u;
        if (!(same || half)) {
            const float W = (float)d.width, H = (float)d.height;
            const float ax = ddx[0] * W, ay = ddx[1] * H, bx = ddy[0] * W, by = ddy[1] * H;
            const float rho = exact_math::sqrt(fmaxf(ax * ax + ay * ay, bx * bx + by * by));
            uint32_t level = 0;
            float frac = 0.0f;
            if (rho > 1.0f && rho < INFINITY) {
                const uint32_t bits = __float_as_uint(rho);
                level = (bits >> 23) - 127u;
                frac = (float)(bits & 0x7FFFFFu) / 8388608.0f;
            } else if (rho == INFINITY) {
                level = d.mips;
            }
            if (level >= d.mips - 1u) { level = d.mips - 1u; frac = 0.0f; }
            share->width = d.width; share->height = d.height; share->mips = d.mips;
            share->level = level; share->frac = frac;
            tex_level_rel(tex_mip_dim(d.width, level), tex_mip_dim(d.height, level), u, v, share->l0);
            if (frac > 0.0f) tex_level_rel(tex_mip_dim(d.width, level + 1u), tex_mip_dim(d.height, level + 1u), u, v, share->l1);
        }
        const uint32_t level = share->level - ((!same && half) ? 1u : 0u);
        const float frac = share->frac;
        const float *rgb = t.decode + (d.format == 1u ? 256 : 0);
        const bool two = frac > 0.0f;
        const uint32_t b0 = d.offset + tex_level_start_pow2(d.width, d.height, level);
        TexLvlFast l0, l1;
        l0.o00 = (b0 + share->l0.r00) << 2; l0.o10 = (b0 + share->l0.r10) << 2; l0.o01 = (b0 + share->l0.r01) << 2; l0.o11 = (b0 + share->l0.r11) << 2;
        l0.fx = share->l0.fx; l0.fy = share->l0.fy;
        Texel4 r = tex_bilinear_fast<M, NEED_A>(t, rgb, l0);
        if (two) {
            const uint32_t b1 = b0 + __umul24(tex_mip_dim(d.width, level), tex_mip_dim(d.height, level));
            l1.o00 = (b1 + share->l1.r00) << 2; l1.o10 = (b1 + share->l1.r10) << 2; l1.o01 = (b1 + share->l1.r01) << 2; l1.o11 = (b1 + share->l1.r11) << 2;
            l1.fx = share->l1.fx; l1.fy = share->l1.fy;
            const Texel4 hi = tex_bilinear_fast<M, NEED_A>(t, rgb, l1);
            const float omf = 1.0f - frac;
            r.rg = mix2<M>(r.rg, hi.rg, frac, omf);
            if (NEED_A) r.ba = mix2<M>(r.ba, hi.ba, frac, omf);
            else r.ba.x = M::mad(hi.ba.x, frac, r.ba.x * omf);
        }
        o[0] = r.rg.x; o[1] = r.rg.y; o[2] = r.ba.x; o[3] = r.ba.y;
        return;
    }
    const bool pow2 = (((d.width & (d.width - 1u)) | (d.height & (d.height - 1u))) == 0u);
    if (SHORT_ONLY || (!nearest && pow2 && t.small_pool != 0u && d.format < R3N_POOL_FLOAT)) {
        // level of detail exactly as tex_footprint derives it
        const float W = (float)d.width, H = (float)d.height;
        const float ax = ddx[0] * W, ay = ddx[1] * H, bx = ddy[0] * W, by = ddy[1] * H;
        const float rho = exact_math::sqrt(fmaxf(ax * ax + ay * ay, bx * bx + by * by));
        uint32_t level = 0;
        float frac = 0.0f;
        if (rho > 1.0f && rho < INFINITY) {
            const uint32_t bits = __float_as_uint(rho);
            level = (bits >> 23) - 127u;
            frac = (float)(bits & 0x7FFFFFu) / 8388608.0f;
        } else if (rho == INFINITY) {
            level = d.mips;
        }
        if (level >= d.mips - 1u) { level = d.mips - 1u; frac = 0.0f; }
        const float *rgb = t.decode + (d.format == 1u ? 256 : 0);
        const uint32_t w0 = tex_mip_dim(d.width, level), h0 = tex_mip_dim(d.height, level);
        // (closed form in the SHORT_ONLY instantiations: no load between the level of detail and the texels; tex_sample_alpha has the note)
        const uint32_t *lo = t.level_off + (size_t)(id - 1u) * R3N_TEX_LEVELS;
        const uint32_t b0 = SHORT_ONLY ? d.offset + tex_level_start_pow2(d.width, d.height, level) : lo[level];
        TexLvlFast l0, l1;
        bool tame = tex_level_fast<SHORT_ONLY>(w0, h0, b0, u, v, l0);
        const bool two = frac > 0.0f;  // then level + 1 <= mips - 1
        if (two) tame = tex_level_fast<SHORT_ONLY>(tex_mip_dim(d.width, level + 1u), tex_mip_dim(d.height, level + 1u), SHORT_ONLY ? b0 + __umul24(w0, h0) : lo[level + 1u], u, v, l1) && tame;
        if (SHORT_ONLY || tame) {
            Texel4 r, hi;
            r = tex_bilinear_fast<M, NEED_A>(t, rgb, l0);
            if (two) hi = tex_bilinear_fast<M, NEED_A>(t, rgb, l1);
            if (two) {
                const float omf = 1.0f - frac;
                r.rg = mix2<M>(r.rg, hi.rg, frac, omf);
                if (NEED_A) r.ba = mix2<M>(r.ba, hi.ba, frac, omf);
                else r.ba.x = M::mad(hi.ba.x, frac, r.ba.x * omf);
            }
            o[0] = r.rg.x; o[1] = r.rg.y; o[2] = r.ba.x; o[3] = r.ba.y;
            return;
        }
    }
    if (SHORT_ONLY) return;  // (not reached)
    TexFootprint f;
    tex_footprint(d, nearest, u, v, ddx, ddy, f);
    const Texel4 r = tex_apply<M, NEED_A>(t, d, f);
    o[0] = r.rg.x; o[1] = r.rg.y; o[2] = r.ba.x; o[3] = r.ba.y;
}

// ALPHA of textureSampleGrad alone: what the rasterisers' cutout test reads (opaque.wgsl:231-235, depth.wgsl:100-127).  On the short
// path a texel costs one word and one division -- decode[c] IS c / 255 as an f32 division (r3n_create builds the table that way), so
// no table load -- instead of four table loads out of global memory per texel (the rasterisers do not stage the tables in LDS): 8
// texels x 4 channels per fragment before.  The alpha channel of tex_sample_grad runs through exactly these operations (a packed
// pair rounds each lane like the scalar form), so the value is the same bit for bit; every other case calls tex_sample_grad.
// `d` = t.descs[id - 1], fetched by the caller once per triangle / work item (the fragments of one share it).
// SHORT_ONLY: the caller knows the short path's conditions hold (and the linear sampler is asked for): no general path behind it,
// wild coordinates handled inside (tex_level_fast<true>).
template <class M = MathExact, bool SHORT_ONLY = false>
R3N_DEV float tex_sample_alpha(const TextureArgs &t, uint32_t id, const r3n_texture_desc32 &d, bool nearest, float u, float v, const float ddx[2], const float ddy[2]) {
    if (id == 0u || id > t.count) return 0.0f;
    const bool pow2 = (((d.width & (d.width - 1u)) | (d.height & (d.height - 1u))) == 0u);
    if (SHORT_ONLY || (!nearest && pow2 && t.small_pool != 0u && d.format < R3N_POOL_FLOAT)) {
        const float W = (float)d.width, H = (float)d.height;
        const float ax = ddx[0] * W, ay = ddx[1] * H, bx = ddy[0] * W, by = ddy[1] * H;
        const float rho = exact_math::sqrt(fmaxf(ax * ax + ay * ay, bx * bx + by * by));
        uint32_t level = 0;
        float frac = 0.0f;
        if (rho > 1.0f && rho < INFINITY) {
            const uint32_t bits = __float_as_uint(rho);
            level = (bits >> 23) - 127u;
            frac = (float)(bits & 0x7FFFFFu) / 8388608.0f;
        } else if (rho == INFINITY) {
            level = d.mips;
        }
        if (level >= d.mips - 1u) { level = d.mips - 1u; frac = 0.0f; }
        // (the levels' first texels in closed form: no load between the level of detail and the texels)
        const uint32_t w0 = tex_mip_dim(d.width, level), h0 = tex_mip_dim(d.height, level);
        // (SHORT_ONLY instantiations only: the ones with the general sampler behind them have no scalar registers to spare for the
        // closed form's intermediates -- tests/test_kernel_isa.py caught the allocator spilling the in-flight destination of the
        // work-item kernels' record prefetch -- and read the table as before)
        const uint32_t *lo = t.level_off + (size_t)(id - 1u) * R3N_TEX_LEVELS;
        const uint32_t b0 = SHORT_ONLY ? d.offset + tex_level_start_pow2(d.width, d.height, level) : lo[level];
        TexLvlFast l0, l1;
        bool tame = tex_level_fast<SHORT_ONLY>(w0, h0, b0, u, v, l0);
        const bool two = frac > 0.0f;
        if (two) tame = tex_level_fast<SHORT_ONLY>(tex_mip_dim(d.width, level + 1u), tex_mip_dim(d.height, level + 1u), SHORT_ONLY ? b0 + __umul24(w0, h0) : lo[level + 1u], u, v, l1) && tame;
        if (SHORT_ONLY || tame) {
            const char *pool = reinterpret_cast<const char *>(t.texels);
            auto byte_at = [&](uint32_t byte_off) { return *reinterpret_cast<const uint32_t *>(pool + byte_off) >> 24; };
            // every texel's alpha byte first (up to eight loads in flight together)
            const uint32_t c00 = byte_at(l0.o00), c10 = byte_at(l0.o10), c01 = byte_at(l0.o01), c11 = byte_at(l0.o11);
            uint32_t d00 = c00, d10 = c10, d01 = c01, d11 = c11;
            if (two) { d00 = byte_at(l1.o00); d10 = byte_at(l1.o10); d01 = byte_at(l1.o01); d11 = byte_at(l1.o11); }
            // A footprint of ONE value -- the inside of a leaf (255) or the empty part of its card (0), most fragments of a foliage
            // scene -- needs no filtering: with every alpha 1.0f the blends are fl(fl(1 - f) + f), which is 1.0f for every f in
            // [0, 1] (the first rounding errs by at most 2^-25, the sum 1 + e rounds back to 1; ties go to even = 1), and 0 * w + 0 * w'
            // is +0.  The value is the filtered one bit for bit; the divisions and the seven blends are skipped.  (Other uniform
            // values are NOT exact this way -- a * (1 - f) + a * f can miss a by an ulp -- and take the arithmetic below.)
            const uint32_t all_and = (c00 & c10) & (c01 & c11) & (d00 & d10) & (d01 & d11);
            const uint32_t all_or = (c00 | c10) | (c01 | c11) | (d00 | d10) | (d01 | d11);
            if (all_and == 255u) return 1.0f;
            if (all_or == 0u) return 0.0f;
            auto bilinear = [&](const TexLvlFast &l, uint32_t b00, uint32_t b10, uint32_t b01, uint32_t b11) {
                const float a00 = exact_math::unorm8(b00), a10 = exact_math::unorm8(b10), a01 = exact_math::unorm8(b01), a11 = exact_math::unorm8(b11);
                const float omx = 1.0f - l.fx, omy = 1.0f - l.fy;
                const float top = M::mad(a10, l.fx, a00 * omx), bot = M::mad(a11, l.fx, a01 * omx);
                return M::mad(bot, l.fy, top * omy);
            };
            float r = bilinear(l0, c00, c10, c01, c11);
            if (two) r = M::mad(bilinear(l1, d00, d10, d01, d11), frac, r * (1.0f - frac));
            return r;
        }
    }
    if (SHORT_ONLY) return 0.0f;  // (not reached)
    float o[4];
    tex_sample_grad<M, true>(t, id, nearest, u, v, ddx, ddy, o);
    return o[3];
}

// The three maps of a material (albedo, normal, AO / roughness / metallic: opaque.wgsl:207-351 samples them at the same coordinates
// with the same gradients) in ONE pass whose memory operations go out in batches -- the descriptors; (the level offsets until round 5: closed form now); the
// (up to 24) texels -- instead of three dependent chains of descriptor -> level offset -> texels, one behind the other: the resolve is
// bound by the LENGTH of its chain of dependent memory round trips at five waves per SIMD, not by instruction issue
// (profiles/r04_summary.md: 8 % fewer vector instructions moved its time by 1 %).
// Works on maps whose levels have the extents of the largest map's levels: the same extents and mip count, or half the extents with
// one level less (TexShare: the level of detail is then the largest map's, shifted by one).  Returns false -- for the whole
// wavefront, having done nothing -- when some lane binds a map that is neither; the caller samples one by one then.
// Every bound id is on the sampler's short path (SHORT_ONLY's contract).  out[k] of an unbound slot (id 0) is zero.
template <class M>
R3N_DEV bool tex_sample3_batched(const TextureArgs &t, const uint32_t id[3], float u, float v, const float ddx[2], const float ddy[2],
                                 float out[3][4]) {
    bool present[3];
    uint32_t w[3], h[3], mips[3], fmt[3], base[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        present[k] = id[k] != 0u;
        const uint32_t idx = present[k] ? id[k] - 1u : 0u;   // (an unbound slot reads descriptor 0: valid memory; its texel addresses are forced into the pool below)
        const uint4 dw = *reinterpret_cast<const uint4 *>(&t.descs[idx]);  // offset, width, height, mips
        base[k] = dw.x; w[k] = dw.y; h[k] = dw.z; mips[k] = dw.w;
        fmt[k] = t.descs[idx].format;
    }
    // the reference map: the widest bound one
    uint32_t W = 0u, H = 0u, MI = 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (present[k] && w[k] > W) { W = w[k]; H = h[k]; MI = mips[k]; }
    bool ok = true, any_half = false, half[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool same = w[k] == W && h[k] == H && mips[k] == MI;
        half[k] = present[k] && !same && (w[k] << 1) == W && (h[k] << 1) == H && mips[k] + 1u == MI;
        ok = ok && (!present[k] || same || half[k]);
        any_half = any_half || half[k];
    }
    if (!__all(ok)) return false;
    // level of detail and footprints of the reference map, exactly as tex_sample_grad's short path derives them
    uint32_t level = 0;
    float frac = 0.0f;
    {
        const float Wf = (float)W, Hf = (float)H;
        const float ax = ddx[0] * Wf, ay = ddx[1] * Hf, bx = ddy[0] * Wf, by = ddy[1] * Hf;
        const float rho = exact_math::sqrt(fmaxf(ax * ax + ay * ay, bx * bx + by * by));
        if (rho > 1.0f && rho < INFINITY) {
            const uint32_t bits = __float_as_uint(rho);
            level = (bits >> 23) - 127u;
            frac = (float)(bits & 0x7FFFFFu) / 8388608.0f;
        } else if (rho == INFINITY) {
            level = MI;
        }
        if (MI != 0u && level >= MI - 1u) { level = MI - 1u; frac = 0.0f; }
    }
    // f0: the footprint at the reference's level `level`, f1: at `level + 1` -- needed by a second level (frac > 0) and by a
    // half-size map when level == 0 (its level 0 has the extents of the reference's level 1; rho / 2 <= 1 there: one level, no blend)
    TexLvlRel f0, f1;
    tex_level_rel(tex_mip_dim(W, level), tex_mip_dim(H, level), u, v, f0);
    const bool need1 = frac > 0.0f || (any_half && level == 0u);
    f1 = f0;
    if (need1) tex_level_rel(tex_mip_dim(W, level + 1u), tex_mip_dim(H, level + 1u), u, v, f1);
    // per map: its first level (with the footprint that level has), whether it blends a second one
    uint32_t la[3], ba[3], bb[3];
    bool first_is_f1[3], two[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        first_is_f1[k] = half[k] && level == 0u;
        la[k] = half[k] ? (level == 0u ? 0u : level - 1u) : level;
        two[k] = frac > 0.0f && !first_is_f1[k];
    }
    // the levels' first texels in closed form (tex_level_start_pow2; a load of two table entries per map before round 5: one of the
    // three round trips of this function)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        ba[k] = base[k] + tex_level_start_pow2(w[k], h[k], la[k]);
        bb[k] = two[k] ? ba[k] + __umul24(tex_mip_dim(w[k], la[k]), tex_mip_dim(h[k], la[k])) : ba[k];
        // An unbound slot still issues its texel loads (results discarded): they go to the START of the pool plus the reference
        // map's footprint offsets, which are smaller than that map's level `level` and therefore than the pool that holds the map --
        // provably inside the allocation wherever the caller placed its textures.
        if (!present[k]) { ba[k] = 0u; bb[k] = 0u; }
    }
    // batch 3: the texels (byte offsets from the uniform pool pointer: the pool holds < 2^30 texels on the short path)
    // A footprint row's two texels are neighbours in memory unless the row wraps (Repeat at the right edge; a 1-texel level): ONE 8-byte
    // load per row (dword-aligned) instead of two 4-byte loads -- the vector L1 works per instruction and line, and this kernel's texel
    // fetches were 48 of its 76 vector-memory instructions; the lanes of a wrapping row fetch their second texel separately (rare,
    // behind a wave-uniform test).  The word behind the pool's last texel exists (r3n.hip pads the pool).
    struct __attribute__((packed, aligned(4))) W2 { uint32_t v[2]; };
    uint32_t ta[3][4], tb[3][4];
    const char *pool = reinterpret_cast<const char *>(t.texels);
    const bool any_two = __any(frac > 0.0f);
    const bool pair0 = f0.r10 == f0.r00 + 1u, pair1 = f1.r10 == f1.r00 + 1u;
    bool wrap_a = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const TexLvlRel &fa = first_is_f1[k] ? f1 : f0;
        const W2 top = *reinterpret_cast<const W2 *>(pool + ((ba[k] + fa.r00) << 2));
        const W2 bot = *reinterpret_cast<const W2 *>(pool + ((ba[k] + fa.r01) << 2));
        ta[k][0] = top.v[0]; ta[k][1] = top.v[1]; ta[k][2] = bot.v[0]; ta[k][3] = bot.v[1];
        wrap_a = wrap_a || !(first_is_f1[k] ? pair1 : pair0);
    }
    if (__any(wrap_a)) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const TexLvlRel &fa = first_is_f1[k] ? f1 : f0;
            if (!(first_is_f1[k] ? pair1 : pair0)) {
                ta[k][1] = *reinterpret_cast<const uint32_t *>(pool + ((ba[k] + fa.r10) << 2));
                ta[k][3] = *reinterpret_cast<const uint32_t *>(pool + ((ba[k] + fa.r11) << 2));
            }
        }
    }
    // (the second level's byte offsets are formed out here, in front of the wave-uniform branch: that is what puts the loads of their
    // level offsets into batch 2 instead of a round trip of their own inside the branch)
    uint32_t ob[3][4];
    bool pair_b[3], wrap_b = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) {  // (lanes without a second level read their first level's texels again)
        const bool use1 = two[k] || first_is_f1[k];
        const TexLvlRel &fb = use1 ? f1 : f0;
        ob[k][0] = (bb[k] + fb.r00) << 2; ob[k][1] = (bb[k] + fb.r10) << 2; ob[k][2] = (bb[k] + fb.r01) << 2; ob[k][3] = (bb[k] + fb.r11) << 2;
        pair_b[k] = use1 ? pair1 : pair0;
        wrap_b = wrap_b || !pair_b[k];
    }
    asm volatile("" : : "v"(ob[0][0]), "v"(ob[1][0]), "v"(ob[2][0]));
    if (any_two) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const W2 top = *reinterpret_cast<const W2 *>(pool + ob[k][0]);
            const W2 bot = *reinterpret_cast<const W2 *>(pool + ob[k][2]);
            tb[k][0] = top.v[0]; tb[k][1] = top.v[1]; tb[k][2] = bot.v[0]; tb[k][3] = bot.v[1];
        }
        if (__any(wrap_b)) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (!pair_b[k]) {
                    tb[k][1] = *reinterpret_cast<const uint32_t *>(pool + ob[k][1]);
                    tb[k][3] = *reinterpret_cast<const uint32_t *>(pool + ob[k][3]);
                }
        }
    }
    // decode + filter: the expressions of tex_texel_at / tex_bilinear_fast / tex_sample_grad
    auto decode = [&](uint32_t word, const float *tab, bool need_a) {
        Texel4 o;
        o.rg = (f2){tab[word & 0xFFu], tab[(word >> 8) & 0xFFu]};
        o.ba = (f2){tab[(word >> 16) & 0xFFu], need_a ? t.decode[word >> 24] : 0.0f};
        return o;
    };
    auto bilinear = [&](const uint32_t q[4], const float *tab, bool need_a, const TexLvlRel &f) {
        const Texel4 c00 = decode(q[0], tab, need_a), c10 = decode(q[1], tab, need_a), c01 = decode(q[2], tab, need_a), c11 = decode(q[3], tab, need_a);
        const float omx = 1.0f - f.fx, omy = 1.0f - f.fy;
        Texel4 o;
        o.rg = mix2<M>(mix2<M>(c00.rg, c10.rg, f.fx, omx), mix2<M>(c01.rg, c11.rg, f.fx, omx), f.fy, omy);
        if (need_a) {
            o.ba = mix2<M>(mix2<M>(c00.ba, c10.ba, f.fx, omx), mix2<M>(c01.ba, c11.ba, f.fx, omx), f.fy, omy);
        } else {
            const float top = M::mad(c10.ba.x, f.fx, c00.ba.x * omx), bot = M::mad(c11.ba.x, f.fx, c01.ba.x * omx);
            o.ba = (f2){M::mad(bot, f.fy, top * omy), 0.0f};
        }
        return o;
    };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool need_a = k != 2;  // albedo and the normal map are read with all four channels (tex4), slot 2 with three (tex3)
        const float *tab = t.decode + (fmt[k] == 1u ? 256 : 0);
        Texel4 r = bilinear(ta[k], tab, need_a, first_is_f1[k] ? f1 : f0);
        if (any_two) {
            const Texel4 hi = bilinear(tb[k], tab, need_a, f1);
            if (two[k]) {
                const float omf = 1.0f - frac;
                r.rg = mix2<M>(r.rg, hi.rg, frac, omf);
                if (need_a) r.ba = mix2<M>(r.ba, hi.ba, frac, omf);
                else r.ba.x = M::mad(hi.ba.x, frac, r.ba.x * omf);
            }
        }
        out[k][0] = present[k] ? r.rg.x : 0.0f; out[k][1] = present[k] ? r.rg.y : 0.0f;
        out[k][2] = present[k] ? r.ba.x : 0.0f; out[k][3] = present[k] ? r.ba.y : 0.0f;
    }
    return true;
}

// vertex_attributes.wgsl: vec2<f32> texture coordinates (attribute 3); a missing attribute reads (0, 0)
R3N_DEV void fetch_uv0(const uint32_t *__restrict__ mesh, uint32_t byte_off, uint32_t vtx, float o[2]) {
    if (byte_off == R3N_INVALID) { o[0] = o[1] = 0.0f; return; }
    const uint32_t w = byte_off / 4u + vtx * 2u;
    const r3n_words2 t = *reinterpret_cast<const r3n_words2 *>(mesh + w);
    o[0] = __uint_as_float(t.x);
    o[1] = __uint_as_float(t.y);
}
// perspective-correct interpolation of a vec2 attribute at the centre of pixel (px, py), covered or not
template <class M = MathExact>
R3N_DEV void interp_vec2(const TriSetup &ts, const float a[3][2], int px, int py, float o[2]) {
    float E[3];
    (void)edge_eval(ts, (float)px + 0.5f, (float)py + 0.5f, E);
    const float rs = M::rcp((E[0] + E[1]) + E[2]);
    const float l0 = E[0] * rs, l1 = E[1] * rs, l2 = E[2] * rs;
    const f2 r = M::mad(splat2(l2), (f2){a[2][0], a[2][1]}, M::mad(splat2(l1), (f2){a[1][0], a[1][1]}, splat2(l0) * (f2){a[0][0], a[0][1]}));
    o[0] = r.x; o[1] = r.y;
}
// (uv_transform * vec3(uv, 1)).xy; mat3x3 stored as three padded vec4 columns
template <class M = MathExact>
R3N_DEV void uv_transform(const float *m, const float uv[2], float o[2]) {
    const f2 r = M::mad((f2){m[8], m[9]}, splat2(1.0f), M::mad((f2){m[4], m[5]}, splat2(uv[1]), (f2){m[0], m[1]} * splat2(uv[0])));
    o[0] = r.x; o[1] = r.y;
}
// Fragment-stage texture coordinates of pixel (x, y) and their derivatives: differences inside the pixel's 2x2 quad
// ("fine": same row for dpdx, same column for dpdy), every operand evaluated like its own (helper) invocation.
// m = uv_transform0 or nullptr (depth.wgsl uses the raw coordinates).  Only the two quad neighbours are evaluated
// here; the pixel's own value comes from `self_raw` (its interpolated coordinates) when the caller already has them.
template <class M = MathExact>
R3N_DEV void frag_coords(const TriSetup &ts, const float uv[3][2], const float *m, int x, int y, float coords[2],
                         float ddx[2], float ddy[2], const float *self_raw = nullptr) {
    float raw[2], self[2], nx[2], ny[2];
    if (self_raw) { raw[0] = self_raw[0]; raw[1] = self_raw[1]; } else interp_vec2<M>(ts, uv, x, y, raw);
    if (m) uv_transform<M>(m, raw, self); else { self[0] = raw[0]; self[1] = raw[1]; }
    interp_vec2<M>(ts, uv, x ^ 1, y, raw);
    if (m) uv_transform<M>(m, raw, nx); else { nx[0] = raw[0]; nx[1] = raw[1]; }
    interp_vec2<M>(ts, uv, x, y ^ 1, raw);
    if (m) uv_transform<M>(m, raw, ny); else { ny[0] = raw[0]; ny[1] = raw[1]; }
    coords[0] = self[0]; coords[1] = self[1];
    // value at the odd pixel of the pair minus value at the even one
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        ddx[k] = (x & 1) ? self[k] - nx[k] : nx[k] - self[k];
        ddy[k] = (y & 1) ? self[k] - ny[k] : ny[k] - self[k];
    }
}
