// texture_decode.hip -- row N2, texture formats: everything rend3-gltf's loader hands to Renderer::add_texture_2d that
// the PBR path samples (rend3-gltf/src/lib.rs:1013-1130; util::map_ktx2_format / map_dxgi_format / map_d3d_format) is
// converted ON THE GPU into the library's texel pool when the texture array is written: R8 / RG8 / BGRA8
// expansion and BC1 / BC2 / BC3 / BC4 / BC5 / BC7 block decoding (Khronos Data Format Specification 1.3: S3TC, RGTC,
// BPTC) into RGBA8 texels; the formats whose values are not 8-bit unorm -- snorm, 16-bit, float, packed float, BC4 / BC5
// snorm, BC6H -- into four f32 per texel (second half of this file).  The reference leaves the decoding to the texture
// unit; a compute rasteriser has none, and decoding once at load keeps the per-pixel sampler (texture.h) a plain fetch.
//
// One thread per 4x4 block: 8 / 16 B in, 64 B out -- a streaming, HBM-bound kernel (80 B per block); BC7 adds ~250
// integer instructions per block.  Rounding conventions (bit-replicated 5:6:5, truncating thirds / sevenths / fifths)
// are the oracle's (oracle/bcn.c), which is pinned against an independent decoder.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/r3n.h"

#define BC7_TABLE static __device__ const
#include "bc7_tables.h"
#define BC6H_TABLE static __device__ const
#include "bc6h_tables.h"

namespace {

__device__ inline void color565(uint32_t c, uint32_t o[3]) {
    const uint32_t r = (c >> 11) & 31u, g = (c >> 5) & 63u, b = c & 31u;
    o[0] = (r << 3) | (r >> 2);
    o[1] = (g << 2) | (g >> 4);
    o[2] = (b << 3) | (b >> 2);
}
__device__ inline uint32_t pack(uint32_t r, uint32_t g, uint32_t b, uint32_t a) { return r | (g << 8) | (b << 16) | (a << 24); }

// S3TC colour block (lo = endpoints, hi = selectors) -> 16 packed RGBA8 texels
__device__ void decode_color_block(uint32_t lo, uint32_t sel, bool bc1, uint32_t out[16]) {
    const uint32_t c0 = lo & 0xFFFFu, c1 = lo >> 16;
    uint32_t p0[3], p1[3], pal[4];
    color565(c0, p0); color565(c1, p1);
    pal[0] = pack(p0[0], p0[1], p0[2], 255u);
    pal[1] = pack(p1[0], p1[1], p1[2], 255u);
    if (!bc1 || c0 > c1) {
        pal[2] = pack((2u * p0[0] + p1[0]) / 3u, (2u * p0[1] + p1[1]) / 3u, (2u * p0[2] + p1[2]) / 3u, 255u);
        pal[3] = pack((p0[0] + 2u * p1[0]) / 3u, (p0[1] + 2u * p1[1]) / 3u, (p0[2] + 2u * p1[2]) / 3u, 255u);
    } else {
        pal[2] = pack((p0[0] + p1[0]) / 2u, (p0[1] + p1[1]) / 2u, (p0[2] + p1[2]) / 2u, 255u);
        pal[3] = 0u;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t s = (sel >> (2 * i)) & 3u;
        out[i] = s == 0u ? pal[0] : (s == 1u ? pal[1] : (s == 2u ? pal[2] : pal[3]));
    }
}

// RGTC / BC3-alpha block -> 16 values
__device__ void decode_alpha_block(uint64_t blk, uint32_t out[16]) {
    const uint32_t a0 = (uint32_t)(blk & 0xFFu), a1 = (uint32_t)((blk >> 8) & 0xFFu);
    const uint64_t sel = blk >> 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t s = (uint32_t)(sel >> (3 * i)) & 7u;
        uint32_t v;
        if (s == 0u) v = a0;
        else if (s == 1u) v = a1;
        else if (a0 > a1) v = ((8u - s) * a0 + (s - 1u) * a1) / 7u;
        else if (s < 6u) v = ((6u - s) * a0 + (s - 1u) * a1) / 5u;
        else v = s == 6u ? 0u : 255u;
        out[i] = v;
    }
}

// ---- BC7
struct Bits128 {
    uint64_t lo, hi;
    uint32_t pos;
    __device__ uint32_t take(uint32_t n) {  // n <= 8
        if (n == 0u) return 0u;
        uint64_t v;
        if (pos >= 64u) v = hi >> (pos - 64u);
        else v = pos == 0u ? lo : ((lo >> pos) | (hi << (64u - pos)));
        pos += n;
        return (uint32_t)v & ((1u << n) - 1u);
    }
};
__device__ inline uint32_t bc7_weight(uint32_t bits, uint32_t i) {
    // 2-bit {0,21,43,64}, 3-bit {0,9,18,27,37,46,55,64}, 4-bit {0,4,9,13,17,21,26,30,34,38,43,47,51,55,60,64}
    const uint64_t W2 = 0x402B1500ull, W3 = 0x40372E251B120900ull, W4L = 0x1E1A15110D090400ull, W4H = 0x403C37332F2B2622ull;
    if (bits == 2u) return (uint32_t)(W2 >> (8u * i)) & 0xFFu;
    if (bits == 3u) return (uint32_t)(W3 >> (8u * i)) & 0xFFu;
    return (uint32_t)((i < 8u ? W4L >> (8u * i) : W4H >> (8u * (i - 8u)))) & 0xFFu;
}
__device__ inline uint32_t lerp7(uint32_t e0, uint32_t e1, uint32_t w) { return ((64u - w) * e0 + w * e1 + 32u) >> 6; }

__device__ void decode_bc7_block(uint64_t lo, uint64_t hi, uint32_t out[16]) {
    // per mode: subsets, partition bits, rotation bits, index-selection bits, colour bits, alpha bits, endpoint p-bits,
    // shared p-bits, index bits, second index bits -- packed 4 bits each, field k at bits [4k, 4k + 4)
    const uint64_t MODES[8] = {0x0301040043ull, 0x0310060062ull, 0x0200050063ull, 0x0201070062ull,
                               0x3200651201ull, 0x2200870201ull, 0x0401770001ull, 0x0201550062ull};
    const uint32_t first = (uint32_t)lo & 0xFFu;
    if (first == 0u) {  // reserved mode: zeros
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i] = 0u;
        return;
    }
    const uint32_t mode = (uint32_t)__builtin_ctz(first);
    const uint64_t M = MODES[mode];
    auto f = [&](int k) { return (uint32_t)(M >> (4 * k)) & 15u; };
    const uint32_t ns = f(0), cb = f(4), ab = f(5), ib = f(8), ib2 = f(9);
    Bits128 b = {lo, hi, mode + 1u};
    const uint32_t part = b.take(f(1)), rot = b.take(f(2)), isb = b.take(f(3));
    uint32_t e[6][4];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (uint32_t k = 0; k < 6u; ++k) e[k][c] = k < 2u * ns ? b.take(cb) : 0u;
#pragma unroll
    for (uint32_t k = 0; k < 6u; ++k) e[k][3] = (ab && k < 2u * ns) ? b.take(ab) : 0u;
    uint32_t cprec = cb, aprec = ab;
    if (f(6)) {
#pragma unroll
        for (uint32_t k = 0; k < 6u; ++k)
            if (k < 2u * ns) {
                const uint32_t p = b.take(1u);
#pragma unroll
                for (int c = 0; c < 4; ++c) e[k][c] = (e[k][c] << 1) | p;
            }
        ++cprec; if (ab) ++aprec;
    } else if (f(7)) {
#pragma unroll
        for (uint32_t s = 0; s < 3u; ++s)
            if (s < ns) {
                const uint32_t p = b.take(1u);
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int c = 0; c < 4; ++c) e[2 * s + k][c] = (e[2 * s + k][c] << 1) | p;
            }
        ++cprec; if (ab) ++aprec;
    }
#pragma unroll
    for (uint32_t k = 0; k < 6u; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint32_t v = e[k][c] << (8u - cprec);
            e[k][c] = v | (v >> cprec);
        }
        if (ab) {
            const uint32_t v = e[k][3] << (8u - aprec);
            e[k][3] = v | (v >> aprec);
        } else {
            e[k][3] = 255u;
        }
    }
    const uint32_t anchor1 = ns == 2u ? BC7_A2[part] : (ns == 3u ? BC7_A3A[part] : 0u);
    const uint32_t anchor2 = ns == 3u ? BC7_A3B[part] : 0u;
    uint32_t subset[16], idx[16];
#pragma unroll
    for (uint32_t i = 0; i < 16u; ++i) {
        const uint32_t s = ns == 1u ? 0u : (ns == 2u ? BC7_P2[part][i] : BC7_P3[part][i]);
        subset[i] = s;
        const bool is_anchor = i == (s == 0u ? 0u : (s == 1u ? anchor1 : anchor2));
        idx[i] = b.take(is_anchor ? ib - 1u : ib);
    }
#pragma unroll
    for (uint32_t i = 0; i < 16u; ++i) {
        const uint32_t i2 = ib2 ? b.take(i == 0u ? ib2 - 1u : ib2) : 0u;
        uint32_t cw, aw;
        if (ib2 == 0u) cw = aw = bc7_weight(ib, idx[i]);
        else if (isb) { cw = bc7_weight(ib2, i2); aw = bc7_weight(ib, idx[i]); }
        else { cw = bc7_weight(ib, idx[i]); aw = bc7_weight(ib2, i2); }
        const uint32_t s = subset[i];
        uint32_t e0[4], e1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            e0[c] = s == 0u ? e[0][c] : (s == 1u ? e[2][c] : e[4][c]);
            e1[c] = s == 0u ? e[1][c] : (s == 1u ? e[3][c] : e[5][c]);
        }
        uint32_t px[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) px[c] = lerp7(e0[c], e1[c], cw);
        px[3] = ab ? lerp7(e0[3], e1[3], aw) : 255u;
        if (rot == 1u) { const uint32_t t = px[3]; px[3] = px[0]; px[0] = t; }
        else if (rot == 2u) { const uint32_t t = px[3]; px[3] = px[1]; px[1] = t; }
        else if (rot == 3u) { const uint32_t t = px[3]; px[3] = px[2]; px[2] = t; }
        out[i] = pack(px[0], px[1], px[2], px[3]);
    }
}

__global__ __launch_bounds__(256) void k_decode_blocks(uint32_t format, uint32_t w, uint32_t h, const uint32_t *__restrict__ src,
                                                       uint32_t *__restrict__ dst) {
    const uint32_t bw = (w + 3u) / 4u, bh = (h + 3u) / 4u;
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= bw * bh) return;
    const uint32_t bx = g % bw, by = g / bw;
    uint32_t px[16];
    const bool small = format == R3N_TEXTURE_BC1_RGBA_UNORM || format == R3N_TEXTURE_BC1_RGBA_UNORM_SRGB || format == R3N_TEXTURE_BC4_R_UNORM;
    const uint32_t *s = src + (size_t)g * (small ? 2u : 4u);
    const uint32_t d0 = s[0], d1 = s[1], d2 = small ? 0u : s[2], d3 = small ? 0u : s[3];
    switch (format) {
    case R3N_TEXTURE_BC1_RGBA_UNORM: case R3N_TEXTURE_BC1_RGBA_UNORM_SRGB:
        decode_color_block(d0, d1, true, px);
        break;
    case R3N_TEXTURE_BC2_RGBA_UNORM: case R3N_TEXTURE_BC2_RGBA_UNORM_SRGB: {
        decode_color_block(d2, d3, false, px);
        const uint64_t a = (uint64_t)d0 | ((uint64_t)d1 << 32);
#pragma unroll
        for (int i = 0; i < 16; ++i) px[i] = (px[i] & 0x00FFFFFFu) | ((((uint32_t)(a >> (4 * i)) & 15u) * 17u) << 24);
        break;
    }
    case R3N_TEXTURE_BC3_RGBA_UNORM: case R3N_TEXTURE_BC3_RGBA_UNORM_SRGB: {
        uint32_t a[16];
        decode_color_block(d2, d3, false, px);
        decode_alpha_block((uint64_t)d0 | ((uint64_t)d1 << 32), a);
#pragma unroll
        for (int i = 0; i < 16; ++i) px[i] = (px[i] & 0x00FFFFFFu) | (a[i] << 24);
        break;
    }
    case R3N_TEXTURE_BC4_R_UNORM: {
        uint32_t r[16];
        decode_alpha_block((uint64_t)d0 | ((uint64_t)d1 << 32), r);
#pragma unroll
        for (int i = 0; i < 16; ++i) px[i] = pack(r[i], 0u, 0u, 255u);
        break;
    }
    case R3N_TEXTURE_BC5_RG_UNORM: {
        uint32_t r[16], gch[16];
        decode_alpha_block((uint64_t)d0 | ((uint64_t)d1 << 32), r);
        decode_alpha_block((uint64_t)d2 | ((uint64_t)d3 << 32), gch);
#pragma unroll
        for (int i = 0; i < 16; ++i) px[i] = pack(r[i], gch[i], 0u, 255u);
        break;
    }
    default:
        decode_bc7_block((uint64_t)d0 | ((uint64_t)d1 << 32), (uint64_t)d2 | ((uint64_t)d3 << 32), px);
        break;
    }
    // rows of a block are 16 B; whole, 16-byte-aligned rows (level width a multiple of 4 and an aligned level start,
    // which r3n_textures_write_encoded arranges for every texture) go out as one dwordx4 store per row
    const bool vec = (w & 3u) == 0u && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0u;
#pragma unroll
    for (uint32_t y = 0; y < 4u; ++y) {
        const uint32_t ty = by * 4u + y;
        if (ty >= h) break;
        uint32_t *row = dst + (size_t)ty * w + bx * 4u;
        if (vec) {
            *reinterpret_cast<uint4 *>(row) = make_uint4(px[y * 4u], px[y * 4u + 1u], px[y * 4u + 2u], px[y * 4u + 3u]);
        } else {
#pragma unroll
            for (uint32_t x = 0; x < 4u; ++x)
                if (bx * 4u + x < w) row[x] = px[y * 4u + x];
        }
    }
}

// uncompressed sources: one thread per texel
__global__ __launch_bounds__(256) void k_expand_texels(uint32_t format, uint64_t n, const uint8_t *__restrict__ src, uint32_t *__restrict__ dst) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t v;
    if (format == R3N_TEXTURE_R8_UNORM) v = pack(src[i], 0u, 0u, 255u);
    else if (format == R3N_TEXTURE_RG8_UNORM) v = pack(src[2 * i], src[2 * i + 1], 0u, 255u);
    else {
        const uint32_t t = reinterpret_cast<const uint32_t *>(src)[i];
        v = (format == R3N_TEXTURE_BGRA8_UNORM || format == R3N_TEXTURE_BGRA8_UNORM_SRGB)
                ? ((t & 0xFF00FF00u) | ((t >> 16) & 0xFFu) | ((t & 0xFFu) << 16)) : t;
    }
    dst[i] = v;
}

// MipmapSource::Generated (rend3/src/util/mipmap.rs:139-236 + rend3/shaders/mipmap.wgsl, K11): level l is a blit of
// level l - 1 through a Linear / ClampToEdge sampler at the destination texel centres into the texture's own format, so
// an sRGB texture is decoded, filtered and re-encoded per level.  One thread per destination texel.  float -> unorm8:
// x * 255 + 0.5, truncated.  The sRGB encode is a search in a 255-entry table of thresholds (thr[c - 1] = the smallest
// float whose code is >= c, found on the host with the same libm expression the oracle evaluates), not a device powf:
// the codes then agree with the oracle for every input.
__global__ __launch_bounds__(256) void k_generate_mip(uint32_t srgb, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh,
                                                      const uint32_t *__restrict__ src, uint32_t *__restrict__ dst,
                                                      const float *__restrict__ decode, const float *__restrict__ thr) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= dw * dh) return;
    const uint32_t x = g % dw, y = g / dw;
    const float u = ((float)x + 0.5f) / (float)dw, v = ((float)y + 0.5f) / (float)dh;
    const float tx = u * (float)sw - 0.5f, ty = v * (float)sh - 0.5f;
    const float fx0 = floorf(tx), fy0 = floorf(ty);
    const float fx = tx - fx0, fy = ty - fy0;
    const int ix = (int)fx0, iy = (int)fy0;
    auto clampi = [](int a, int hi) { return a < 0 ? 0 : (a > hi ? hi : a); };
    const uint32_t x0 = (uint32_t)clampi(ix, (int)sw - 1), x1 = (uint32_t)clampi(ix + 1, (int)sw - 1);
    const uint32_t y0 = (uint32_t)clampi(iy, (int)sh - 1), y1 = (uint32_t)clampi(iy + 1, (int)sh - 1);
    const uint32_t t[4] = {src[(size_t)y0 * sw + x0], src[(size_t)y0 * sw + x1], src[(size_t)y1 * sw + x0], src[(size_t)y1 * sw + x1]};
    uint32_t out = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool s = srgb != 0u && c < 3;
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = decode[(s ? 256u : 0u) + ((t[k] >> (8 * c)) & 0xFFu)];
        const float top = q[0] * (1.0f - fx) + q[1] * fx;
        const float bot = q[2] * (1.0f - fx) + q[3] * fx;
        const float r = top * (1.0f - fy) + bot * fy;
        uint32_t code;
        if (s) {
            uint32_t lo = 0, hi = 255;  // code = number of thresholds <= r (NaN compares false: code 0, like the formula)
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (r >= thr[mid]) lo = mid + 1u; else hi = mid;
            }
            code = lo;
        } else {
            const float e = fminf(fmaxf(r, 0.0f), 1.0f);
            code = (uint32_t)(e * 255.0f + 0.5f);
        }
        out |= code << (8 * c);
    }
    dst[g] = out;
}


// ---- formats decoded to float texels.  Value definitions and rounding: oracle/bcn.c (r3o_texture_decode_level_f32), which
// cites the specifications; the two agree bit for bit (every conversion below is exact or one correctly rounded division).
__device__ inline float pow2f(int k) { return __uint_as_float((uint32_t)(k + 127) << 23); }  // -126 <= k <= 127
__device__ inline float half_to_float(uint32_t h) {
    const uint32_t s = (h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
    if (e == 0u) return __uint_as_float(s | __float_as_uint((float)m * pow2f(-24)));
    if (e == 31u) return __uint_as_float(s | 0x7F800000u | (m << 13));
    return __uint_as_float(s | ((e + 112u) << 23) | (m << 13));
}
__device__ inline float ufloat_to_float(uint32_t v, uint32_t mb) {  // 5 exponent bits, mb mantissa bits, no sign
    const uint32_t e = v >> mb, m = v & ((1u << mb) - 1u);
    if (e == 0u) return (float)m * pow2f(-14 - (int)mb);
    if (e == 31u) return __uint_as_float(0x7F800000u | (m << (23u - mb)));
    return (float)((1u << mb) | m) * pow2f((int)e - 15 - (int)mb);
}
__device__ inline float snorm8(uint32_t c) { return fmaxf((float)(int)(int8_t)c / 127.0f, -1.0f); }
__device__ inline float snorm16(uint32_t c) { return fmaxf((float)(int)(int16_t)c / 32767.0f, -1.0f); }

// one thread per texel; `src` is read through the narrowest aligned type of the format
__global__ __launch_bounds__(256) void k_expand_texels_f32(uint32_t format, uint64_t n, const uint8_t *__restrict__ src, float4 *__restrict__ dst) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint16_t *s16 = reinterpret_cast<const uint16_t *>(src);
    const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    switch (format) {
    case R3N_TEXTURE_R8_SNORM: o.x = snorm8(src[i]); break;
    case R3N_TEXTURE_RG8_SNORM: { const uint32_t v = s16[i]; o.x = snorm8(v & 0xFFu); o.y = snorm8(v >> 8); break; }
    case R3N_TEXTURE_RGBA8_SNORM: { const uint32_t v = s32[i]; o = make_float4(snorm8(v & 0xFFu), snorm8((v >> 8) & 0xFFu), snorm8((v >> 16) & 0xFFu), snorm8(v >> 24)); break; }
    case R3N_TEXTURE_R16_FLOAT: o.x = half_to_float(s16[i]); break;
    case R3N_TEXTURE_RG16_FLOAT: { const uint32_t v = s32[i]; o.x = half_to_float(v & 0xFFFFu); o.y = half_to_float(v >> 16); break; }
    case R3N_TEXTURE_RGBA16_FLOAT: { const uint32_t a = s32[2 * i], b = s32[2 * i + 1]; o = make_float4(half_to_float(a & 0xFFFFu), half_to_float(a >> 16), half_to_float(b & 0xFFFFu), half_to_float(b >> 16)); break; }
    case R3N_TEXTURE_R32_FLOAT: o.x = __uint_as_float(s32[i]); break;
    case R3N_TEXTURE_RG32_FLOAT: o.x = __uint_as_float(s32[2 * i]); o.y = __uint_as_float(s32[2 * i + 1]); break;
    case R3N_TEXTURE_RGBA32_FLOAT: o = make_float4(__uint_as_float(s32[4 * i]), __uint_as_float(s32[4 * i + 1]), __uint_as_float(s32[4 * i + 2]), __uint_as_float(s32[4 * i + 3])); break;
    case R3N_TEXTURE_RGBA16_UNORM: { const uint32_t a = s32[2 * i], b = s32[2 * i + 1]; o = make_float4((float)(a & 0xFFFFu) / 65535.0f, (float)(a >> 16) / 65535.0f, (float)(b & 0xFFFFu) / 65535.0f, (float)(b >> 16) / 65535.0f); break; }
    case R3N_TEXTURE_RGBA16_SNORM: { const uint32_t a = s32[2 * i], b = s32[2 * i + 1]; o = make_float4(snorm16(a & 0xFFFFu), snorm16(a >> 16), snorm16(b & 0xFFFFu), snorm16(b >> 16)); break; }
    case R3N_TEXTURE_RGB10A2_UNORM: { const uint32_t v = s32[i]; o = make_float4((float)(v & 1023u) / 1023.0f, (float)((v >> 10) & 1023u) / 1023.0f, (float)((v >> 20) & 1023u) / 1023.0f, (float)(v >> 30) / 3.0f); break; }
    case R3N_TEXTURE_RG11B10_FLOAT: { const uint32_t v = s32[i]; o.x = ufloat_to_float(v & 2047u, 6u); o.y = ufloat_to_float((v >> 11) & 2047u, 6u); o.z = ufloat_to_float(v >> 22, 5u); break; }
    default: {  // R3N_TEXTURE_RGB9E5_UFLOAT
        const uint32_t v = s32[i];
        const float sc = pow2f((int)(v >> 27) - 24);
        o.x = (float)(v & 511u) * sc; o.y = (float)((v >> 9) & 511u) * sc; o.z = (float)((v >> 18) & 511u) * sc;
        break;
    }
    }
    dst[i] = o;
}

// RGTC signed block -> 16 floats (snorm8 endpoints, integer ordering, palette interpolated in f32)
__device__ void decode_snorm_block(uint64_t blk, float out[16]) {
    const int a0 = (int)(int8_t)(blk & 0xFFu), a1 = (int)(int8_t)((blk >> 8) & 0xFFu);
    const float f0 = fmaxf((float)a0 / 127.0f, -1.0f), f1 = fmaxf((float)a1 / 127.0f, -1.0f);
    const uint64_t sel = blk >> 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t k = (uint32_t)(sel >> (3 * i)) & 7u;
        float v;
        if (k == 0u) v = f0;
        else if (k == 1u) v = f1;
        else if (a0 > a1) v = ((float)(8u - k) * f0 + (float)(k - 1u) * f1) / 7.0f;
        else if (k < 6u) v = ((float)(6u - k) * f0 + (float)(k - 1u) * f1) / 5.0f;
        else v = k == 6u ? -1.0f : 1.0f;
        out[i] = v;
    }
}

// ---- BC6H (BPTC float): bit layouts from bc6h_tables.h, integer pipeline of the specification
__device__ inline uint32_t bits128(uint64_t lo, uint64_t hi, uint32_t pos, uint32_t n) {  // 1 <= n <= 16
    uint64_t v;
    if (pos >= 64u) v = hi >> (pos - 64u);
    else v = pos == 0u ? lo : ((lo >> pos) | (hi << (64u - pos)));
    return (uint32_t)v & ((1u << n) - 1u);
}
__device__ inline int sign_extend(int v, uint32_t bits) { return (v & (1 << (bits - 1u))) ? v - (1 << bits) : v; }
__device__ inline int bc6h_unquantize(int x, uint32_t epb, bool is_signed) {
    if (!is_signed) {
        if (epb >= 15u) return x;
        if (x == 0) return 0;
        if (x == (1 << epb) - 1) return 0xFFFF;
        return ((x << 16) + 0x8000) >> epb;
    }
    if (epb >= 16u) return x;
    const bool neg = x < 0;
    if (neg) x = -x;
    int u;
    if (x == 0) u = 0;
    else if (x >= (1 << (epb - 1u)) - 1) u = 0x7FFF;
    else u = ((x << 15) + 0x4000) >> (epb - 1u);
    return neg ? -u : u;
}
__device__ inline uint32_t bc6h_finalize(int v, bool is_signed) {
    if (!is_signed) return (uint32_t)((v * 31) >> 6);
    if (v < 0) return 0x8000u | (uint32_t)(((-v) * 31) >> 5);
    return (uint32_t)((v * 31) >> 5);
}
__device__ void decode_bc6h_block(uint64_t lo, uint64_t hi, bool is_signed, float4 out[16]) {
    const uint32_t first = (uint32_t)lo & 31u;
    int mode = -1;
    if ((first & 3u) < 2u) mode = (int)(first & 3u);
    else
        for (int k = 2; k < 14; ++k)
            if ((BC6H_MODE[k] & 31u) == first) mode = k;
    if (mode < 0) {  // reserved: zeros
#pragma unroll
        for (int i = 0; i < 16; ++i) out[i] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        return;
    }
    const uint32_t info = BC6H_MODE[mode];
    const uint32_t epb = (info >> 8) & 31u, regions = ((info >> 29) & 1u) + 1u;
    const bool transformed = ((info >> 28) & 1u) != 0u;
    int f[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) f[k] = 0;
    uint32_t pos = (info >> 5) & 7u;
    for (int k = 0; k < BC6H_MAX_ENTRIES; ++k) {
        const uint32_t e = BC6H_LAYOUT[mode][k];
        if (e == 0xFFFFu) break;
        const uint32_t fld = e & 15u, lsb = (e >> 4) & 15u, n = ((e >> 8) & 15u) + 1u;
        uint32_t v = bits128(lo, hi, pos, n);
        if ((e >> 12) & 1u) v = __brev(v) >> (32u - n);
        // f[] is indexed dynamically: kept in scratch; a load-time kernel
        f[fld] |= (int)(v << lsb);
        pos += n;
    }
    int ep[4][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t dbits = (info >> (13 + 5 * c)) & 31u;
        int base = f[c];
        if (is_signed) base = sign_extend(base, epb);
        ep[0][c] = base;
#pragma unroll
        for (uint32_t e = 1; e < 4u; ++e) {
            int v = f[3 * e + c];
            if (transformed) {
                v = sign_extend(v, dbits);
                v = (base + v) & ((1 << epb) - 1);
                if (is_signed) v = sign_extend(v, epb);
            } else if (is_signed) {
                v = sign_extend(v, dbits);
            }
            ep[e][c] = v;
        }
#pragma unroll
        for (uint32_t e = 0; e < 4u; ++e) ep[e][c] = bc6h_unquantize(ep[e][c], epb, is_signed);
    }
    const uint32_t part = (uint32_t)f[12] & 31u;
    const uint32_t ib = regions == 2u ? 3u : 4u;
    const uint32_t anchor = regions == 2u ? BC7_A2[part] : 0u;
    pos = regions == 2u ? 82u : 65u;
    for (uint32_t i = 0; i < 16u; ++i) {
        const uint32_t sub = regions == 2u ? BC7_P2[part][i] : 0u;
        const uint32_t n = (i == 0u || (regions == 2u && i == anchor)) ? ib - 1u : ib;
        const uint32_t idx = bits128(lo, hi, pos, n);
        pos += n;
        const int w = (int)bc7_weight(ib, idx);
        float ch[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int e0 = sub ? ep[2][c] : ep[0][c], e1 = sub ? ep[3][c] : ep[1][c];
            ch[c] = half_to_float(bc6h_finalize((e0 * (64 - w) + e1 * w + 32) >> 6, is_signed));
        }
        out[i] = make_float4(ch[0], ch[1], ch[2], 1.0f);
    }
}

__global__ __launch_bounds__(256) void k_decode_blocks_f32(uint32_t format, uint32_t w, uint32_t h, const uint32_t *__restrict__ src,
                                                           float4 *__restrict__ dst) {
    const uint32_t bw = (w + 3u) / 4u, bh = (h + 3u) / 4u;
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= bw * bh) return;
    const uint32_t bx = g % bw, by = g / bw;
    const bool small = format == R3N_TEXTURE_BC4_R_SNORM;
    const uint32_t *s = src + (size_t)g * (small ? 2u : 4u);
    const uint64_t lo = (uint64_t)s[0] | ((uint64_t)s[1] << 32), hi = small ? 0ull : ((uint64_t)s[2] | ((uint64_t)s[3] << 32));
    float4 px[16];
    if (format == R3N_TEXTURE_BC4_R_SNORM || format == R3N_TEXTURE_BC5_RG_SNORM) {
        float r[16], gch[16];
        decode_snorm_block(lo, r);
        if (!small) decode_snorm_block(hi, gch);
#pragma unroll
        for (int i = 0; i < 16; ++i) px[i] = make_float4(r[i], small ? 0.0f : gch[i], 0.0f, 1.0f);
    } else {
        decode_bc6h_block(lo, hi, format == R3N_TEXTURE_BC6H_RGB_FLOAT, px);
    }
#pragma unroll
    for (uint32_t y = 0; y < 4u; ++y) {
        const uint32_t ty = by * 4u + y;
        if (ty >= h) break;
#pragma unroll
        for (uint32_t x = 0; x < 4u; ++x)
            if (bx * 4u + x < w) dst[(size_t)ty * w + bx * 4u + x] = px[y * 4u + x];
    }
}


// MipmapSource::Generated for the float-decoded formats the loader generates chains for (single-level R16Float / Rg16Float /
// Rgba16Float / Rgb10a2Unorm files): the blit of k_generate_mip on float texels, then the render-target write of the format --
// f32 -> binary16 round to nearest even, done in integers so that every bit pattern (NaN payloads too) is the oracle's
// (oracle/bcn.c float_to_half_rne); unorm: clamp, x * (2^n - 1) + 0.5 truncated.  Channels the format does not store keep (0, 0, 1).
__device__ inline uint32_t float_to_half_rne(float f) {
    const uint32_t x = __float_as_uint(f);
    const uint32_t sign = (x >> 16) & 0x8000u, mag = x & 0x7FFFFFFFu;
    if (mag >= 0x7F800000u) return sign | 0x7C00u | (mag > 0x7F800000u ? (0x200u | ((mag >> 13) & 0x3FFu)) : 0u);
    if (mag >= 0x477FF000u) return sign | 0x7C00u;
    if (mag < 0x33000001u) return sign;
    const int e = (int)(mag >> 23) - 127;
    const uint32_t m = (mag & 0x7FFFFFu) | 0x800000u;
    const int shift = e >= -14 ? 13 : 13 + (-14 - e);
    const uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1u);
    uint32_t q = m >> shift;
    if (rest > half || (rest == half && (q & 1u))) ++q;
    return sign | (e >= -14 ? ((uint32_t)(e + 14) << 10) + q : q);
}
__device__ inline float quantize_channel(uint32_t format, int c, float r) {
    if (format == R3N_TEXTURE_RGB10A2_UNORM) {
        const float e = fminf(fmaxf(r, 0.0f), 1.0f);
        return c < 3 ? (float)(uint32_t)(e * 1023.0f + 0.5f) / 1023.0f : (float)(uint32_t)(e * 3.0f + 0.5f) / 3.0f;
    }
    return half_to_float(float_to_half_rne(r));
}
__global__ __launch_bounds__(256) void k_generate_mip_f32(uint32_t format, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh,
                                                          const float4 *__restrict__ src, float4 *__restrict__ dst) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= dw * dh) return;
    const uint32_t x = g % dw, y = g / dw;
    const float u = ((float)x + 0.5f) / (float)dw, v = ((float)y + 0.5f) / (float)dh;
    const float tx = u * (float)sw - 0.5f, ty = v * (float)sh - 0.5f;
    const float fx0 = floorf(tx), fy0 = floorf(ty);
    const float fx = tx - fx0, fy = ty - fy0;
    const int ix = (int)fx0, iy = (int)fy0;
    auto clampi = [](int a, int hi) { return a < 0 ? 0 : (a > hi ? hi : a); };
    const uint32_t x0 = (uint32_t)clampi(ix, (int)sw - 1), x1 = (uint32_t)clampi(ix + 1, (int)sw - 1);
    const uint32_t y0 = (uint32_t)clampi(iy, (int)sh - 1), y1 = (uint32_t)clampi(iy + 1, (int)sh - 1);
    const float4 t4[4] = {src[(size_t)y0 * sw + x0], src[(size_t)y0 * sw + x1], src[(size_t)y1 * sw + x0], src[(size_t)y1 * sw + x1]};
    const int stored = format == R3N_TEXTURE_R16_FLOAT ? 1 : (format == R3N_TEXTURE_RG16_FLOAT ? 2 : 4);
    float o[4] = {0.0f, 0.0f, 0.0f, 1.0f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c >= stored) continue;
        const float q0 = c == 0 ? t4[0].x : (c == 1 ? t4[0].y : (c == 2 ? t4[0].z : t4[0].w));
        const float q1 = c == 0 ? t4[1].x : (c == 1 ? t4[1].y : (c == 2 ? t4[1].z : t4[1].w));
        const float q2 = c == 0 ? t4[2].x : (c == 1 ? t4[2].y : (c == 2 ? t4[2].z : t4[2].w));
        const float q3 = c == 0 ? t4[3].x : (c == 1 ? t4[3].y : (c == 2 ? t4[3].z : t4[3].w));
        const float top = q0 * (1.0f - fx) + q1 * fx;
        const float bot = q2 * (1.0f - fx) + q3 * fx;
        o[c] = quantize_channel(format, c, top * (1.0f - fy) + bot * fy);
    }
    dst[g] = make_float4(o[0], o[1], o[2], o[3]);
}

}  // namespace

extern "C" int r3n_internal_generate_mip(uint32_t srgb, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh, const uint32_t *src,
                                         uint32_t *dst, const float *decode, const float *thr, hipStream_t stream) {
    hipLaunchKernelGGL(k_generate_mip, dim3((dw * dh + 255u) / 256u), dim3(256), 0, stream, srgb, sw, sh, dw, dh, src, dst, decode, thr);
    return (int)hipGetLastError();
}

// bytes of one w x h level in `format`; 0 for an unknown format
extern "C" uint64_t r3n_internal_level_bytes(uint32_t format, uint32_t w, uint32_t h) {
    switch (format) {
    case R3N_TEXTURE_RGBA8_UNORM: case R3N_TEXTURE_RGBA8_UNORM_SRGB: case R3N_TEXTURE_BGRA8_UNORM: case R3N_TEXTURE_BGRA8_UNORM_SRGB:
        return (uint64_t)w * h * 4u;
    case R3N_TEXTURE_R8_UNORM: return (uint64_t)w * h;
    case R3N_TEXTURE_RG8_UNORM: return (uint64_t)w * h * 2u;
    case R3N_TEXTURE_BC1_RGBA_UNORM: case R3N_TEXTURE_BC1_RGBA_UNORM_SRGB: case R3N_TEXTURE_BC4_R_UNORM:
        return (uint64_t)((w + 3u) / 4u) * ((h + 3u) / 4u) * 8u;
    case R3N_TEXTURE_BC2_RGBA_UNORM: case R3N_TEXTURE_BC2_RGBA_UNORM_SRGB: case R3N_TEXTURE_BC3_RGBA_UNORM:
    case R3N_TEXTURE_BC3_RGBA_UNORM_SRGB: case R3N_TEXTURE_BC5_RG_UNORM: case R3N_TEXTURE_BC7_RGBA_UNORM:
    case R3N_TEXTURE_BC7_RGBA_UNORM_SRGB: case R3N_TEXTURE_BC5_RG_SNORM: case R3N_TEXTURE_BC6H_RGB_UFLOAT: case R3N_TEXTURE_BC6H_RGB_FLOAT:
        return (uint64_t)((w + 3u) / 4u) * ((h + 3u) / 4u) * 16u;
    case R3N_TEXTURE_BC4_R_SNORM: return (uint64_t)((w + 3u) / 4u) * ((h + 3u) / 4u) * 8u;
    case R3N_TEXTURE_R8_SNORM: return (uint64_t)w * h;
    case R3N_TEXTURE_RG8_SNORM: case R3N_TEXTURE_R16_FLOAT: return (uint64_t)w * h * 2u;
    case R3N_TEXTURE_RGBA8_SNORM: case R3N_TEXTURE_RG16_FLOAT: case R3N_TEXTURE_R32_FLOAT: case R3N_TEXTURE_RGB10A2_UNORM:
    case R3N_TEXTURE_RG11B10_FLOAT: case R3N_TEXTURE_RGB9E5_UFLOAT:
        return (uint64_t)w * h * 4u;
    case R3N_TEXTURE_RGBA16_FLOAT: case R3N_TEXTURE_RG32_FLOAT: case R3N_TEXTURE_RGBA16_UNORM: case R3N_TEXTURE_RGBA16_SNORM:
        return (uint64_t)w * h * 8u;
    case R3N_TEXTURE_RGBA32_FLOAT: return (uint64_t)w * h * 16u;
    default: return 0;
    }
}

// 1: the format decodes to four f32 per texel (r3n_internal_decode_level_f32), 0: to RGBA8
extern "C" int r3n_internal_format_is_float(uint32_t format) { return format >= R3N_TEXTURE_R8_SNORM && format < R3N_TEXTURE_FORMAT_COUNT; }
// required alignment of a level's first byte in the payload (the decoders read it through that type)
extern "C" uint32_t r3n_internal_format_align(uint32_t format) {
    switch (format) {
    case R3N_TEXTURE_R8_UNORM: case R3N_TEXTURE_RG8_UNORM: case R3N_TEXTURE_R8_SNORM: return 1u;
    case R3N_TEXTURE_RG8_SNORM: case R3N_TEXTURE_R16_FLOAT: return 2u;
    default: return 4u;
    }
}

// one level: device source (its own format) -> w * h RGBA8 texels at `dst` (device)
extern "C" int r3n_internal_decode_level(uint32_t format, uint32_t w, uint32_t h, const void *src, uint32_t *dst, hipStream_t stream) {
    if (format >= R3N_TEXTURE_BC1_RGBA_UNORM) {
        const uint32_t blocks = ((w + 3u) / 4u) * ((h + 3u) / 4u);
        hipLaunchKernelGGL(k_decode_blocks, dim3((blocks + 255u) / 256u), dim3(256), 0, stream, format, w, h,
                           static_cast<const uint32_t *>(src), dst);
    } else {
        const uint64_t n = (uint64_t)w * h;
        hipLaunchKernelGGL(k_expand_texels, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, stream, format, n,
                           static_cast<const uint8_t *>(src), dst);
    }
    return (int)hipGetLastError();
}

// one level: device source (its own format) -> w * h float texels at `dst` (device, 16-byte aligned)
extern "C" int r3n_internal_decode_level_f32(uint32_t format, uint32_t w, uint32_t h, const void *src, float *dst, hipStream_t stream) {
    if (format >= R3N_TEXTURE_BC4_R_SNORM) {
        const uint32_t blocks = ((w + 3u) / 4u) * ((h + 3u) / 4u);
        hipLaunchKernelGGL(k_decode_blocks_f32, dim3((blocks + 255u) / 256u), dim3(256), 0, stream, format, w, h,
                           static_cast<const uint32_t *>(src), reinterpret_cast<float4 *>(dst));
    } else {
        const uint64_t n = (uint64_t)w * h;
        hipLaunchKernelGGL(k_expand_texels_f32, dim3((unsigned)((n + 255u) / 256u)), dim3(256), 0, stream, format, n,
                           static_cast<const uint8_t *>(src), reinterpret_cast<float4 *>(dst));
    }
    return (int)hipGetLastError();
}

// 1: MipmapSource::Generated is available for this float-decoded format (the loader generates chains for single-level files of it)
extern "C" int r3n_internal_format_generates_mips_f32(uint32_t format) {
    return format == R3N_TEXTURE_R16_FLOAT || format == R3N_TEXTURE_RG16_FLOAT || format == R3N_TEXTURE_RGBA16_FLOAT || format == R3N_TEXTURE_RGB10A2_UNORM;
}
extern "C" int r3n_internal_generate_mip_f32(uint32_t format, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh, const float *src, float *dst,
                                             hipStream_t stream) {
    hipLaunchKernelGGL(k_generate_mip_f32, dim3((dw * dh + 255u) / 256u), dim3(256), 0, stream, format, sw, sh, dw, dh,
                       reinterpret_cast<const float4 *>(src), reinterpret_cast<float4 *>(dst));
    return (int)hipGetLastError();
}
