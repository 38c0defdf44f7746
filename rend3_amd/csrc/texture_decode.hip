// texture_decode.hip -- row N2, texture formats: everything rend3-gltf's loader hands to Renderer::add_texture_2d that
// the PBR path samples (rend3-gltf/src/lib.rs:1013-1130; util::map_ktx2_format / map_dxgi_format / map_d3d_format) is
// converted ON THE GPU into the library's RGBA8 texel pool when the texture array is written: R8 / RG8 / BGRA8
// expansion and BC1 / BC2 / BC3 / BC4 / BC5 / BC7 block decoding (Khronos Data Format Specification 1.3: S3TC, RGTC,
// BPTC).  The reference leaves the decoding to the texture unit; a compute rasteriser has none, and decoding once at
// load keeps the per-pixel sampler (texture.h) a plain RGBA8 fetch.
//
// One thread per 4x4 block: 8 / 16 B in, 64 B out -- a streaming, HBM-bound kernel (80 B per block); BC7 adds ~250
// integer instructions per block.  Rounding conventions (bit-replicated 5:6:5, truncating thirds / sevenths / fifths)
// are the oracle's (oracle/bcn.c), which is pinned against an independent decoder.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/r3n.h"

#define BC7_TABLE static __device__ const
#include "bc7_tables.h"

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
    case R3N_TEXTURE_BC7_RGBA_UNORM_SRGB:
        return (uint64_t)((w + 3u) / 4u) * ((h + 3u) / 4u) * 16u;
    default: return 0;
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
