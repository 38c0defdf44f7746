// kernels_raster.h -- software rasteriser (K5/K6 coverage + depth), Hi-Z pyramid (K3), deferred PBR
// resolve (K6 shading) and tonemap (K7).
//
// Reference behaviour restated (file:line):
//   forward.rs:318-371  pipeline state: cull Back (forward) / Front (depth), depth GreaterEqual + write
//   depth.wgsl:51-127, opaque.wgsl:91-135 (VS), :203-551 (FS), math/brdf.wgsl, shadow/pcf.wgsl
//   hi_z.wgsl:19-32, hi_z.rs:161-234
//   blit.wgsl:22-31, tonemapping.rs:44 (Rgba8UnormSrgb target => exact sRGB OETF)
//
// Design: a 64-bit visibility buffer (depth bits << 32 | canonical triangle slot + 1) written
// with 64-bit atomic max -- reverse-Z GreaterEqual + depth write is exactly "max" -- so opaque shading
// runs once per pixel for the nearest fragment instead of once per rasterised fragment.  Triangles up to
// 8x8 px are scanned by one thread; larger ones are split into <=64x64 px work items scanned by one
// wavefront each (8x8 pixel blocks per step).
#pragma once
#include "device_math.h"
#include "texture.h"

struct RasterArgs {
    const r3n_camera_header240 *hdr;
    const r3n_object128 *objects;
    const uint32_t *mesh;
    const r3n_baked128 *baked;
    const r3n_material208 *materials;
    const uint8_t *material_keys;
    uint32_t n_materials;
    const uint32_t *tri_base;              // canonical slot base per object (forward only)
    const r3n_tri_ref *list;               // compacted triangle list of this camera/source
    const uint32_t *sub_counts;            // [3][R3N_SUBQ] triangles per (material key, sub-list) of that list
    uint32_t subcap;                       // entries reserved per (material key, sub-list)
    uint32_t key;                          // region to draw
    uint32_t vp_x, vp_y, vp_w, vp_h;       // viewport inside the target
    uint32_t target_pitch;                 // elements per row of the target
    unsigned long long *vis;               // forward target (u64 per sample, samples per pixel contiguous) or null
    uint32_t *depth;                       // depth-only target (f32 bits per pixel) or null
    r3n_big_item *big_items;               // R3N_BIGQ sub-queues of big_capacity entries each
    uint32_t *big_count;                   // [R3N_BIGQ]
    uint32_t big_capacity;
    r3n_big_uv *big_uv;                    // same indexing as big_items; textured cutout triangles only
    TextureArgs tex;
    // transparent pass (row N3): fragments that pass the depth test are appended here instead of written as keys
    unsigned long long *frag_keys;         // (pixel * samples + sample) << 32 | draw order of the triangle
    uint32_t *frag_vals;                   // canonical slot + 1
    uint32_t *frag_count;
    uint32_t frag_capacity;
    uint32_t row_begin, row_end;           // rows this rank resolves
};

// opaque.wgsl:214-235 / depth.wgsl:98-125 (untextured paths): alpha the cutout test compares with the threshold
// tex_alpha = alpha of the albedo texture sample, or 1 without one
R3N_DEV float cutout_alpha(uint32_t mat_flags, float mat_alpha, float tex_alpha, float vertex_alpha) {
    float alpha = 1.0f;
    if (mat_flags & R3N_FLAGS_ALBEDO_ACTIVE) {
        alpha = tex_alpha;
        if (mat_flags & R3N_FLAGS_ALBEDO_BLEND) alpha *= vertex_alpha;
    }
    alpha *= mat_alpha;
    return alpha;
}

R3N_DEV float fetch_color_alpha(const r3n_object128 &ob, const uint32_t *__restrict__ mesh, uint32_t vtx) {
    const uint32_t off = ob.vertex_attribute_start_offsets[5];
    if (off == R3N_INVALID) return 1.0f;
    const uint32_t w = mesh[off / 4u + vtx];
    return (float)((w >> 24) & 0xFFu) / 255.0f;
}

// Everything needed to scan one triangle.
struct TriWork {
    TriSetup ts;
    float va[3];
    uint32_t material;                  // material index
    uint32_t mat_flags;                 // cutout key only: material flags, albedo alpha, alpha_cutout
    float mat_alpha, mat_cutoff;
    uint32_t slot1;  // canonical slot + 1 (forward)
    float uv[3][2];                     // cutout key + albedo texture only
    bool alpha_tex;                     // the cutout alpha samples the albedo texture
    bool cutout;
    int x0, y0, x1, y1;
};

// TEX: the launch may meet cutout materials whose alpha comes from the albedo texture (row N2).  The lean variant
// (no texture code, fewer registers) is launched whenever the world has no textures or the key is not cutout.
template <bool DEPTH_ONLY, bool TEX>
R3N_DEV bool prepare_triangle(const RasterArgs &a, uint32_t obj, uint32_t tri, bool positive_visible, TriWork &tw) {
    const r3n_object128 &ob = a.objects[obj];
    if (ob.enabled == 0u) return false;  // opaque.wgsl:104-112 / depth.wgsl:64-72
    const uint32_t first = ob.first_index + tri * 3u;
    const uint32_t pos_off = ob.vertex_attribute_start_offsets[0];
    uint32_t idx[3];
    float p[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        idx[k] = a.mesh[first + (uint32_t)k];
        float v[3];
        fetch_vec3(a.mesh, pos_off, idx[k], v);
        mul_point(a.baked[obj].model_view_proj, v, p[k]);
    }
    const float half_w = (float)a.vp_w / 2.0f, half_h = (float)a.vp_h / 2.0f;
    setup_triangle(p, half_w, half_h, positive_visible, tw.ts);
    if (!tw.ts.valid) return false;
    if (!tri_bounds(p, half_w, half_h, (int)a.vp_w, (int)a.vp_h, tw.x0, tw.y0, tw.x1, tw.y1)) return false;
    tw.cutout = a.key == R3N_KEY_CUTOUT;
    tw.material = ob.material_index < a.n_materials ? ob.material_index : 0u;
    tw.va[0] = tw.va[1] = tw.va[2] = 1.0f;
    tw.mat_flags = 0u; tw.mat_alpha = 1.0f; tw.mat_cutoff = 0.0f;
    tw.alpha_tex = false;
#pragma unroll
    for (int k = 0; k < 3; ++k) tw.uv[k][0] = tw.uv[k][1] = 0.0f;
    if (tw.cutout) {
#pragma unroll
        for (int k = 0; k < 3; ++k) tw.va[k] = fetch_color_alpha(ob, a.mesh, idx[k]);
        const r3n_material208 &m = a.materials[tw.material];
        tw.mat_flags = m.flags; tw.mat_alpha = m.albedo[3]; tw.mat_cutoff = m.alpha_cutout;
        tw.alpha_tex = TEX && (m.flags & R3N_FLAGS_ALBEDO_ACTIVE) && m.textures[0] != 0u;
        if (TEX && tw.alpha_tex) {
#pragma unroll
            for (int k = 0; k < 3; ++k) fetch_uv0(a.mesh, ob.vertex_attribute_start_offsets[3], idx[k], tw.uv[k]);
        }
    }
    if (!DEPTH_ONLY) tw.slot1 = a.tri_base[obj] + tri + 1u;
    return true;
}

#ifndef R3N_PREREAD_SMALL
#define R3N_PREREAD_SMALL 0
#endif
#ifndef R3N_PREREAD_MS
#define R3N_PREREAD_MS 1        // work-item kernel on a multisampled viewport: read the pixel's keys before the atomics
#endif
#ifndef R3N_PREREAD_VIEWPORT
#define R3N_PREREAD_VIEWPORT 0  // the same at one sample per pixel: the kernel alone gains (168 -> 152 us) but the frame with
                                // frames in flight loses (1.18 -> 1.20 ms): off
#endif
#ifndef R3N_PREREAD_BIG
#define R3N_PREREAD_BIG 0
#endif
// PREREAD: plain load + compare before the atomic.  It filters occluded fragments cheaply (the load may be
// stale, which is only conservative because keys grow monotonically) but puts a dependent load in front of every
// atomic; without it the atomic is fire-and-forget.
// Alpha of the albedo texture at pixel (x, y) for the cutout test.  Forward (opaque.wgsl:207-215): coordinates through
// uv_transform0, sampler chosen by FLAGS_NEAREST.  Depth-only (depth.wgsl:108-118, quirks reproduced): raw coords0,
// always the primary sampler, and uvdy = dpdx(coords).
template <bool DEPTH_ONLY, bool TEX>
R3N_DEV float cutout_texture_alpha(const RasterArgs &a, const TriWork &tw, int x, int y) {
    if (!TEX || !tw.alpha_tex) return 1.0f;
    const r3n_material208 &m = a.materials[tw.material];
    float coords[2], ddx[2], ddy[2], texel[4];
    if (DEPTH_ONLY) {
        frag_coords(tw.ts, tw.uv, nullptr, x, y, coords, ddx, ddy);
        tex_sample_grad(a.tex, m.textures[0], false, coords[0], coords[1], ddx, ddx, texel);
    } else {
        frag_coords(tw.ts, tw.uv, m.uv_transform0, x, y, coords, ddx, ddy);
        tex_sample_grad(a.tex, m.textures[0], (m.flags & R3N_FLAGS_NEAREST) != 0u, coords[0], coords[1], ddx, ddy, texel);
    }
    return texel[3];
}

// Multisampling (row N4; forward.rs:358 MultisampleState{count}): coverage and depth at the standard 4x sample
// positions (the D3D / Vulkan standard locations every wgpu backend uses); the fragment -- here only its cutout
// alpha -- once per pixel at the pixel centre, covered or not (no centroid qualifier in opaque.wgsl).
__device__ static const float k_sample_pos4[4][2] = {{0.375f, 0.125f}, {0.875f, 0.375f}, {0.125f, 0.625f}, {0.625f, 0.875f}};

// Depth / visibility atomics, explicitly in the GLOBAL address space.  k_raster_big pins its kernel arguments through
// an asm statement, after which the compiler no longer knows the targets are global and would emit FLAT atomics, which
// also count on lgkmcnt -- the counter the next record's scalar prefetch is waited on.
R3N_DEV void global_max_u32(uint32_t *p, uint32_t v) {
    typedef __attribute__((address_space(1))) uint32_t *gp_t;
    (void)__hip_atomic_fetch_max((gp_t)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
R3N_DEV void global_max_u64(unsigned long long *p, unsigned long long v) {
    typedef __attribute__((address_space(1))) unsigned long long *gp_t;
    (void)__hip_atomic_fetch_max((gp_t)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool DEPTH_ONLY, bool PREREAD, int S = 1, bool TEX = false, bool BLEND = false>
R3N_DEV void shade_pixel(const RasterArgs &a, const TriWork &tw, int x, int y) {
    if (BLEND) {
        // Transparent pass: depth test GreaterEqual against the final opaque depth, depth write off (pbr/routine.rs:
        // 113-118); what passes is recorded for the ordered blend (k_blend_apply).  tw.material = draw order.
        if ((uint32_t)y < a.row_begin || (uint32_t)y >= a.row_end) return;
        const size_t pix = (size_t)y * a.target_pitch + (size_t)x;
#pragma unroll
        for (int sm = 0; sm < S; ++sm) {
            float E[3];
            const float sx = S == 1 ? 0.5f : k_sample_pos4[sm][0], sy = S == 1 ? 0.5f : k_sample_pos4[sm][1];
            if (!edge_eval(tw.ts, (float)x + sx, (float)y + sy, E)) continue;
            const float z = frag_depth(tw.ts, E);
            if (!(z >= 0.0f && z <= 1.0f)) continue;
            const size_t ps = pix * (size_t)S + (size_t)sm;
            const float dz = __uint_as_float((uint32_t)(a.vis[ps] >> 32));
            if (!(z >= dz)) continue;
            const uint32_t at = atomicAdd(a.frag_count, 1u);
            if (at < a.frag_capacity) {
                a.frag_keys[at] = ((unsigned long long)ps << 32) | (unsigned long long)tw.material;
                a.frag_vals[at] = tw.slot1;
            }
        }
        return;
    }
    if (S == 1) {
        float E[3];
        if (!edge_eval(tw.ts, (float)x + 0.5f, (float)y + 0.5f, E)) return;
        float z = frag_depth(tw.ts, E);
        if (!(z >= 0.0f && z <= 1.0f)) return;  // depth clip (unclipped_depth: false, forward.rs:343)
        if (z == 0.0f) z = 0.0f;                // canonicalise -0
        if (tw.cutout) {
            const float rs = 1.0f / ((E[0] + E[1]) + E[2]);
            const float al = ((E[0] * rs) * tw.va[0] + (E[1] * rs) * tw.va[1]) + (E[2] * rs) * tw.va[2];
            if (cutout_alpha(tw.mat_flags, tw.mat_alpha, cutout_texture_alpha<DEPTH_ONLY, TEX>(a, tw, x, y), al) < tw.mat_cutoff) return;  // opaque.wgsl:231-235 / depth.wgsl:123-125
        }
        const size_t pix = (size_t)(a.vp_y + (uint32_t)y) * a.target_pitch + a.vp_x + (uint32_t)x;
        const uint32_t zb = __float_as_uint(z);
#if R3N_ABLATE == 2
        asm volatile("" : : "v"(zb), "v"(pix));
        return;
#endif
        if (DEPTH_ONLY) {
            if (!PREREAD || zb > a.depth[pix]) global_max_u32(&a.depth[pix], zb);
        } else {
            const unsigned long long key = ((unsigned long long)zb << 32) | (unsigned long long)tw.slot1;
            if (!PREREAD || key > a.vis[pix]) global_max_u64(&a.vis[pix], key);
        }
    } else {
        uint32_t mask = 0u;
        float zs[S];
#pragma unroll
        for (int sm = 0; sm < S; ++sm) {
            float E[3];
            zs[sm] = 0.0f;
            if (!edge_eval(tw.ts, (float)x + k_sample_pos4[sm][0], (float)y + k_sample_pos4[sm][1], E)) continue;
            float z = frag_depth(tw.ts, E);
            if (!(z >= 0.0f && z <= 1.0f)) continue;
            if (z == 0.0f) z = 0.0f;
            zs[sm] = z;
            mask |= 1u << sm;
        }
        if (!mask) return;
        if (tw.cutout) {
            float E[3];
            (void)edge_eval(tw.ts, (float)x + 0.5f, (float)y + 0.5f, E);
            const float rs = 1.0f / ((E[0] + E[1]) + E[2]);
            const float al = ((E[0] * rs) * tw.va[0] + (E[1] * rs) * tw.va[1]) + (E[2] * rs) * tw.va[2];
            if (cutout_alpha(tw.mat_flags, tw.mat_alpha, cutout_texture_alpha<DEPTH_ONLY, TEX>(a, tw, x, y), al) < tw.mat_cutoff) return;
        }
        const size_t pix = ((size_t)(a.vp_y + (uint32_t)y) * a.target_pitch + a.vp_x + (uint32_t)x) * (size_t)S;
#if R3N_ABLATE == 2
        asm volatile("" : : "v"(zs[0]), "v"(zs[1]), "v"(zs[2]), "v"(zs[3]), "v"(mask), "v"(pix));
        return;
#endif
        // PREREAD (the work-item kernel under MSAA): the pixel's four keys are 32 contiguous bytes; read them once and
        // skip the atomics that cannot win.  Memory-side atomics are 65 % of that kernel at 4 samples (ablation,
        // bench scene: 1.43 ms -> 0.50 ms without them); the read removes the overdrawn ones: 1.43 -> 1.04 ms.  Keys only
        // grow, so a key that loses against a stale read loses against the current value too.
        unsigned long long cur[S];
        if (PREREAD) {
#pragma unroll
            for (int sm = 0; sm < S; ++sm) cur[sm] = a.vis[pix + (size_t)sm];
        }
#pragma unroll
        for (int sm = 0; sm < S; ++sm)
            if (mask & (1u << sm)) {
                const unsigned long long key = ((unsigned long long)__float_as_uint(zs[sm]) << 32) | (unsigned long long)tw.slot1;
                if (!PREREAD || key > cur[sm]) global_max_u64(&a.vis[pix + (size_t)sm], key);
            }
    }
}

#ifndef R3N_SMALL_MAX
#define R3N_SMALL_MAX 8
#endif
// Work items cover at most R3N_TILE x R3N_TILE px.  Measured on the bench scene (us per frame, shadow big / viewport
// big): tile 64 coarse-only 478 / 218, tile 32 coarse-only 475 / 204, tile 32 + fine 423 / 185, tile 16 + fine 482 / 191.
#ifndef R3N_TILE
#define R3N_TILE 32
#endif
#ifndef R3N_SMALL_OCC
#define R3N_SMALL_OCC 1  // min waves per SIMD asked of k_raster_small (launch bound)
#endif
#ifndef R3N_XCD_REMAP
#define R3N_XCD_REMAP 0  // resolve tiles in contiguous per-XCD bands: measured, no gain (shade 595 -> 593 us, frame 1.184 -> 1.193 ms)
#endif
#ifndef R3N_MS_OCC
#define R3N_MS_OCC 5   // the same for the multisampled record-based resolve (lean split pass: 0.86 ms at 5, 0.92 at 4)
#endif
#ifndef R3N_TEX_OCC
#define R3N_TEX_OCC 5  // min waves per SIMD asked of the textured record-based resolve (launch bound)
#endif
#ifndef R3N_SKIP_OCCLUDED
#define R3N_SKIP_OCCLUDED 1
#endif
#ifndef R3N_ABLATE
#define R3N_ABLATE 0  // diagnostics only (tools/variants.py): 1 no scan steps, 2 no atomics, 3 no block test / scan
#endif
#ifndef R3N_FINE
#define R3N_FINE 1    // regions of the tile size are scanned four 4x4 blocks per step instead of one 8x8 block
#endif

// Stage 1: one thread per list entry.  Small triangles are scanned in place; larger ones are split into
// <=64x64 px items for stage 2.
template <bool DEPTH_ONLY, int S = 1, bool TEX = false>
__global__ __launch_bounds__(256, R3N_SMALL_OCC) void k_raster_small(RasterArgs a) {
    // block b walks sub-list (b % R3N_SUBQ) of the region, and appends to work sub-queue (b % R3N_BIGQ)
    const uint32_t q = blockIdx.x % R3N_SUBQ;
    const uint32_t n = a.sub_counts[a.key * R3N_SUBQ + q];
    const r3n_tri_ref *list = a.list + (size_t)(a.key * R3N_SUBQ + q) * a.subcap;
    // producer side of the work queue: the sub-queue is chosen per WAVE.  readfirstlane makes the counter address
    // provably wave-uniform, which is what lets the compiler fold a wave's appends into ONE atomic (with a
    // per-lane address every lane issues its own returning atomic -- measured 4x slower kernels)
    const uint32_t bq = __builtin_amdgcn_readfirstlane((blockIdx.x * 4u + (threadIdx.x >> 6)) % R3N_BIGQ);
    const bool positive_visible = (a.hdr->flags & R3N_PCU_POSITIVE_AREA_VISIBLE) != 0u;
    const uint32_t stride = (gridDim.x / R3N_SUBQ) * 256u;
    for (uint32_t i = (blockIdx.x / R3N_SUBQ) * 256u + threadIdx.x; i < n; i += stride) {
        const r3n_tri_ref ref = list[i];
        TriWork tw;
        if (!prepare_triangle<DEPTH_ONLY, TEX>(a, ref.object, ref.triangle, positive_visible, tw)) continue;
        const int bw = tw.x1 - tw.x0 + 1, bh = tw.y1 - tw.y0 + 1;
        if (bw <= R3N_SMALL_MAX && bh <= R3N_SMALL_MAX) {
            for (int y = tw.y0; y <= tw.y1; ++y)
                for (int x = tw.x0; x <= tw.x1; ++x) shade_pixel<DEPTH_ONLY, R3N_PREREAD_SMALL != 0, S, TEX>(a, tw, x, y);
        } else {
            const uint32_t tx = (uint32_t)(bw + (R3N_TILE - 1)) / R3N_TILE, ty = (uint32_t)(bh + (R3N_TILE - 1)) / R3N_TILE;
            const uint32_t cnt = tx * ty;
            const uint32_t start = atomicAdd(&a.big_count[bq], cnt);
            r3n_big_item *big = a.big_items + (size_t)bq * a.big_capacity;
            for (uint32_t t = 0; t < cnt; ++t) {
                const uint32_t ix = t % tx, iy = t / tx;
                const int rx0 = tw.x0 + (int)ix * R3N_TILE, ry0 = tw.y0 + (int)iy * R3N_TILE;
                const int rx1 = min(rx0 + (R3N_TILE - 1), tw.x1), ry1 = min(ry0 + (R3N_TILE - 1), tw.y1);
                if (start + t < a.big_capacity) {
                    r3n_big_item it;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) it.e[i][c] = tw.ts.e[i][c];
                        it.z[i] = tw.ts.z[i];
                        it.va[i] = tw.va[i];
                    }
                    it.det = tw.ts.det;
                    it.slot1 = DEPTH_ONLY ? 0u : tw.slot1;
                    it.material = tw.material;
                    it.xy0 = (uint32_t)rx0 | ((uint32_t)ry0 << 16);
                    it.xy1 = (uint32_t)rx1 | ((uint32_t)ry1 << 16);
                    big[start + t] = it;
                    if (TEX && tw.alpha_tex) {
                        r3n_big_uv bu;
#pragma unroll
                        for (int k = 0; k < 3; ++k) { bu.uv[k][0] = tw.uv[k][0]; bu.uv[k][1] = tw.uv[k][1]; }
                        bu._pad[0] = bu._pad[1] = 0u;
                        a.big_uv[(size_t)bq * a.big_capacity + start + t] = bu;
                    }
                } else {
                    // queue full: never drop work -- scan the region here (slow path)
                    for (int y = ry0; y <= ry1; ++y)
                        for (int x = rx0; x <= rx1; ++x) shade_pixel<DEPTH_ONLY, R3N_PREREAD_SMALL != 0, S, TEX>(a, tw, x, y);
                }
            }
        }
    }
}

// Transparent pass, stage 1 (row N3): one thread per triangle of the blend-key objects, in DRAW ORDER -- objects back
// to front (blend_order, sorted on the host like batching.rs:146-176), triangles in index order; g is therefore
// the triangle's draw order.  Triangles that passed this frame's cull (cull.wgsl:372-378 writes exactly those into
// the non-atomic residual list) are set up and split into <= R3N_TILE^2 px work items; stage 2 is k_raster_big in
// BLEND mode.
struct BlendSetupArgs {
    const uint32_t *order;             // blend objects, back to front
    const uint32_t *rank_base;         // n + 1: exclusive scan of their triangle counts
    uint32_t n_objects;
    const unsigned long long *mask;    // this frame's cull result bits (viewport)
    const uint32_t *slot_base;         // first bit of each object in `mask`, or INVALID when it was not batched
};
__global__ __launch_bounds__(256) void k_blend_setup(RasterArgs a, BlendSetupArgs b) {
    const uint32_t total = b.rank_base[b.n_objects];
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;
    if (g >= total) return;
    uint32_t lo = 0, hi = b.n_objects;  // last k with rank_base[k] <= g
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (b.rank_base[mid] <= g) lo = mid; else hi = mid;
    }
    const uint32_t obj = b.order[lo], tri = g - b.rank_base[lo];
    const uint32_t sb = b.slot_base[obj];
    if (sb == R3N_INVALID) return;
    const uint32_t bit = sb + tri;
    if (!((b.mask[bit >> 6] >> (bit & 63u)) & 1ull)) return;
    const bool positive_visible = (a.hdr->flags & R3N_PCU_POSITIVE_AREA_VISIBLE) != 0u;
    TriWork tw;
    if (!prepare_triangle<false, false>(a, obj, tri, positive_visible, tw)) return;
    const int bw = tw.x1 - tw.x0 + 1, bh = tw.y1 - tw.y0 + 1;
    const uint32_t tx = (uint32_t)(bw + (R3N_TILE - 1)) / R3N_TILE, ty = (uint32_t)(bh + (R3N_TILE - 1)) / R3N_TILE;
    const uint32_t cnt = tx * ty;
    const uint32_t bq = (blockIdx.x * 4u + (threadIdx.x >> 6)) % R3N_BIGQ;
    const uint32_t start = atomicAdd(&a.big_count[bq], cnt);
    r3n_big_item *big = a.big_items + (size_t)bq * a.big_capacity;
    for (uint32_t t = 0; t < cnt; ++t) {
        const uint32_t ix = t % tx, iy = t / tx;
        const int rx0 = tw.x0 + (int)ix * R3N_TILE, ry0 = tw.y0 + (int)iy * R3N_TILE;
        const int rx1 = min(rx0 + (R3N_TILE - 1), tw.x1), ry1 = min(ry0 + (R3N_TILE - 1), tw.y1);
        if (start + t >= a.big_capacity) {  // the caller sees big_count > capacity and fails the frame
            continue;
        }
        r3n_big_item it;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) it.e[i][c] = tw.ts.e[i][c];
            it.z[i] = tw.ts.z[i];
            it.va[i] = 1.0f;
        }
        it.det = tw.ts.det;
        it.slot1 = tw.slot1;
        it.material = g;  // draw order
        it.xy0 = (uint32_t)rx0 | ((uint32_t)ry0 << 16);
        it.xy1 = (uint32_t)rx1 | ((uint32_t)ry1 << 16);
        big[start + t] = it;
    }
}

// Upper bound of edge function i over the pixel centres of an SxS block whose first pixel is (bx,by).  Each
// f32 operation is monotone, so evaluating the same expression at the extreme corner gives the exact maximum of
// the per-pixel values: a block with a negative maximum holds no covered pixel.
// MS: the evaluation points are the 4x sample positions, which span [0.125, 0.875] of a pixel in x and y.
template <int S = 8, bool MS = false>
R3N_DEV bool block_may_cover(const TriSetup &ts, int bx, int by, int rx1, int ry1) {
    const float lo = MS ? 0.125f : 0.5f, hi = MS ? 0.875f : 0.5f;
    const float x_lo = (float)bx + lo, x_hi = (float)min(bx + (S - 1), rx1) + hi;
    const float y_lo = (float)by + lo, y_hi = (float)min(by + (S - 1), ry1) + hi;
    bool may = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float A = ts.e[i][0], B = ts.e[i][1];
        const float px = A > 0.0f ? x_hi : x_lo, py = B > 0.0f ? y_hi : y_lo;
        const float v = (A * px + B * py) + ts.e[i][2];
        may = may && !(v < 0.0f);  // NaN keeps the block (per-pixel evaluation rejects it)
    }
    return may;
}

// Stage 2: scan of the work items (<= 64x64 px each): one wavefront per item, lane = 8x8 block for the rejection
// test, then lane = pixel for every surviving block.  The item record is loaded once (lane j reads dword j) and
// broadcast through SGPRs with readlane, so the triangle's constants cost no vector registers or vector ALU work;
// the next item's record is in flight while the current one is scanned.  Waves walk the concatenation of the
// R3N_BIGQ producer sub-queues with a stride of the wave count, which spreads neighbouring (similar-cost) items
// over different waves.
#ifdef R3N_WAVE_TRACE
// diagnostics build only (tools/wave_trace.py): per wave of the shadow-view launches {start, end (s_memrealtime, 100 MHz
// ticks, low 32 bits), items, scan steps}, indexed by the cascade's atlas quadrant
__device__ uint32_t g_wave_trace[4][32768][4];
#endif
template <bool DEPTH_ONLY, int S = 1, bool TEX = false, bool BLEND = false>
__global__ __launch_bounds__(256) void k_raster_big(RasterArgs a) {
#ifdef R3N_WAVE_TRACE
    const uint32_t trace_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
    uint32_t trace_items = 0, trace_steps = 0;
#endif
    // Pin the kernel arguments the scan uses into SGPRs here: hipcc otherwise sinks the wait for their s_load
    // into the scan loop, and an `s_waitcnt lgkmcnt(0)` there would also wait for the record prefetch below.
    asm volatile("" : "+s"(a.target_pitch), "+s"(a.vp_x), "+s"(a.vp_y), "+s"(a.depth), "+s"(a.vis), "+s"(a.key),
                 "+s"(a.materials), "+s"(a.big_items), "+s"(a.big_count), "+s"(a.big_capacity));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    const uint32_t nwaves = gridDim.x * 4u;
    const int lx = (int)(lane & 7u), ly = (int)(lane >> 3);
    const uint32_t cap = a.big_capacity;
    // Sub-queue bounds live in registers: lane q < R3N_BIGQ holds [excl, incl), the flat indices of sub-queue q in the
    // concatenation (one vector load + a wave scan at kernel start, before any atomic is in flight).  Locating an item
    // is then a ballot + readlane, with no memory access in the item loop.
    const uint32_t qcnt_l = lane < R3N_BIGQ ? min(a.big_count[lane], cap) : 0u;
    uint32_t incl = qcnt_l;
#pragma unroll
    for (uint32_t d = 1; d < R3N_BIGQ; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const uint32_t excl = incl - qcnt_l;
    auto locate = [&](uint32_t flat) -> const uint32_t * {  // nullptr past the end
        const unsigned long long m = __ballot(flat >= excl && flat < incl);  // lanes >= R3N_BIGQ hold an empty range
        if (!m) return nullptr;
        const uint32_t q = (uint32_t)__builtin_ctzll(m);
        const uint32_t qb = __builtin_amdgcn_readlane(excl, q);
        return reinterpret_cast<const uint32_t *>(a.big_items + (size_t)q * cap + (flat - qb));
    };
    typedef __attribute__((address_space(4))) const uint32_t *sptr_t;
    // The record is read with SCALAR loads (constant address space + wave-uniform address => s_load into SGPRs).
    // A vector load would share the vmcnt counter with the scan's fire-and-forget atomics, and waiting for the
    // record would then wait for every outstanding atomic of the previous item (measured: 2.4 us per item).
    // Scalar-cache coherence is not an issue: the queue was written by the previous kernel on this stream.
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t flat = wave_global;
    const uint32_t *rec = locate(flat);
    u32x16 da = {}, na = {};
    u32x4 db = {}, nb = {};
    if (rec) {
        sptr_t sp = (sptr_t)(unsigned long long)rec;
        da = *reinterpret_cast<__attribute__((address_space(4))) const u32x16 *>(sp);
        db = *reinterpret_cast<__attribute__((address_space(4))) const u32x4 *>(sp + 16);
    }
    while (rec) {
        // The next record's scalar loads stay in flight while this item is scanned (the records come from HBM /
        // Infinity Cache: without the overlap every item costs a full memory latency per wave).  hipcc sinks a
        // plain load to its first use, so the prefetch is an asm load it does not track; the matching wait
        // statement at the end of the iteration names both destinations (cdna_hip_programming.md section 5.7 (ii)).
        flat += nwaves;
        const uint32_t *nrec = locate(flat);
        if (nrec)
            asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x40"
                         : "=&s"(na), "=&s"(nb) : "s"(nrec) : "memory");
        uint32_t d[20];
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = da[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[16 + j] = db[j];
        auto bu = [&](int j) { return d[j]; };
        auto bf = [&](int j) { return __uint_as_float(d[j]); };
        TriWork w;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) w.ts.e[i][c] = bf(3 * i + c);
            w.ts.z[i] = bf(9 + i);
            w.va[i] = bf(13 + i);
        }
        w.ts.det = bf(12);
        w.ts.valid = true;
        w.slot1 = bu(16);
        w.cutout = !BLEND && a.key == R3N_KEY_CUTOUT;  // launch-uniform
        w.material = bu(17);
        w.mat_flags = 0u; w.mat_alpha = 1.0f; w.mat_cutoff = 0.0f;
        w.alpha_tex = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) w.uv[k][0] = w.uv[k][1] = 0.0f;
        if (w.cutout) {
            sptr_t mp = (sptr_t)(unsigned long long)(a.materials + w.material);
            w.mat_alpha = __uint_as_float(mp[offsetof(r3n_material208, albedo) / 4 + 3]);
            w.mat_cutoff = __uint_as_float(mp[offsetof(r3n_material208, alpha_cutout) / 4]);
            w.mat_flags = mp[offsetof(r3n_material208, flags) / 4];
            w.alpha_tex = TEX && (w.mat_flags & R3N_FLAGS_ALBEDO_ACTIVE) && mp[0] != 0u;
            if (TEX && w.alpha_tex) {  // rec - big_items = item index; the uv record has the same index
                const size_t item = (size_t)(reinterpret_cast<const r3n_big_item *>(rec) - a.big_items);
                sptr_t up = (sptr_t)(unsigned long long)(a.big_uv + item);
#pragma unroll
                for (int k = 0; k < 3; ++k) { w.uv[k][0] = __uint_as_float(up[2 * k]); w.uv[k][1] = __uint_as_float(up[2 * k + 1]); }
            }
        }
        const uint32_t kxy0 = bu(18), kxy1 = bu(19);
        const int rx0 = (int)(kxy0 & 0xFFFFu), ry0 = (int)(kxy0 >> 16);
        const int rx1 = (int)(kxy1 & 0xFFFFu), ry1 = (int)(kxy1 >> 16);
#if R3N_ABLATE == 3
        asm volatile("" : : "s"(rx0), "s"(ry0), "s"(rx1), "s"(ry1), "s"(w.ts.e[0][0]), "s"(w.ts.z[2]));
        if (false) {
#else
        if (R3N_FINE && rx1 - rx0 < 32 && ry1 - ry0 < 32) {
#endif
            // fine mode (regions up to 32x32 px): lane = 4x4 block for the rejection test; every step then scans
            // FOUR surviving blocks, 16 lanes each -- small triangles fill the wave far better than with 8x8 blocks
            const int cbx = rx0 + lx * 4, cby = ry0 + ly * 4;
            const bool cand = cbx <= rx1 && cby <= ry1 && block_may_cover<4, (S > 1)>(w.ts, cbx, cby, rx1, ry1);
            unsigned long long blocks = __ballot(cand);
#if R3N_ABLATE == 1
            asm volatile("" : : "s"(blocks));
            blocks = 0ull;
#endif
            const uint32_t grp = lane >> 4;
            const int px = (int)(lane & 3u), py = (int)((lane >> 2) & 3u);
            while (blocks) {
#ifdef R3N_WAVE_TRACE
                ++trace_steps;
#endif
                int bsel[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bsel[g] = blocks ? __builtin_ctzll(blocks) : 64;
                    blocks &= blocks - 1ull;
                }
                const int b = grp == 0u ? bsel[0] : (grp == 1u ? bsel[1] : (grp == 2u ? bsel[2] : bsel[3]));
                const int x = rx0 + (b & 7) * 4 + px, y = ry0 + (b >> 3) * 4 + py;
                if (b < 64 && x <= rx1 && y <= ry1) shade_pixel<DEPTH_ONLY, (R3N_PREREAD_BIG != 0) || (!DEPTH_ONLY && (S > 1 ? R3N_PREREAD_MS != 0 : R3N_PREREAD_VIEWPORT != 0)), S, TEX, BLEND>(a, w, x, y);
            }
        } else if (R3N_ABLATE != 3) {
            const int cbx = rx0 + lx * 8, cby = ry0 + ly * 8;
            const bool cand = cbx <= rx1 && cby <= ry1 && block_may_cover<8, (S > 1)>(w.ts, cbx, cby, rx1, ry1);
            unsigned long long blocks = __ballot(cand);
            while (blocks) {
#ifdef R3N_WAVE_TRACE
                ++trace_steps;
#endif
                const int b = __builtin_ctzll(blocks);
                blocks &= blocks - 1ull;
                const int x = rx0 + (b & 7) * 8 + lx, y = ry0 + (b >> 3) * 8 + ly;
                if (x <= rx1 && y <= ry1) shade_pixel<DEPTH_ONLY, (R3N_PREREAD_BIG != 0) || (!DEPTH_ONLY && (S > 1 ? R3N_PREREAD_MS != 0 : R3N_PREREAD_VIEWPORT != 0)), S, TEX, BLEND>(a, w, x, y);
            }
        }
#ifdef R3N_WAVE_TRACE
        ++trace_items;
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(na), "+s"(nb) : : "memory");
        rec = nrec;
        da = na;
        db = nb;
    }
#ifdef R3N_WAVE_TRACE
    if (DEPTH_ONLY && lane == 0u && wave_global < 32768u) {
        const uint32_t quad = (a.vp_x ? 1u : 0u) + (a.vp_y ? 2u : 0u);
        uint32_t *t = g_wave_trace[quad][wave_global];
        t[0] = trace_t0; t[1] = (uint32_t)__builtin_amdgcn_s_memrealtime(); t[2] = trace_items; t[3] = trace_steps;
    }
#endif
}

// ------------------------------------------------------------------------------------------------ clears
__global__ __launch_bounds__(256) void k_fill_u64(unsigned long long *__restrict__ p, unsigned long long v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) p[i] = v;
}

// ------------------------------------------------------------------------------------------------ K3 Hi-Z
// Level 0: depth plane = high 32 bits of the visibility keys (background 0.0 = infinitely far).
__global__ __launch_bounds__(256) void k_hiz_mip0(const unsigned long long *__restrict__ vis, float *__restrict__ mip0,
                                                  size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u)
        mip0[i] = __uint_as_float((uint32_t)(vis[i] >> 32));
}

// hi_z.wgsl:19-32: dst = min over the 2x2 (3 wide/high when the source dimension is odd) source texels;
// out-of-range source loads read 0.0.
__global__ __launch_bounds__(256) void k_hiz_downsample(const float *__restrict__ src, float *__restrict__ dst,
                                                        uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh) {
    const uint32_t x = blockIdx.x * 16u + (threadIdx.x & 15u);
    const uint32_t y = blockIdx.y * 16u + (threadIdx.x >> 4);
    if (x >= dw || y >= dh) return;
    const uint32_t nx = 2u + (sw & 1u), ny = 2u + (sh & 1u);
    float nearest = 1.0f;
    for (uint32_t ix = 0; ix < nx; ++ix)
        for (uint32_t iy = 0; iy < ny; ++iy) {
            const uint32_t sx = 2u * x + ix, sy = 2u * y + iy;
            const float v = (sx < sw && sy < sh) ? src[(size_t)sy * sw + sx] : 0.0f;
            nearest = fminf(nearest, v);
        }
    dst[(size_t)y * dw + x] = nearest;
}

// Fused head of the pyramid: one block reduces a 32x32 depth tile through mip0 .. mip`levels` (levels <= 4) in one
// pass -- registers for the first 2x2, LDS for the rest.  Only used for levels whose SOURCE dimensions are even (then
// hi_z.wgsl's window is a plain 2x2 and tiles are independent); the host picks `levels` accordingly.
// With multisampling mip 0 is the depth resolve of resolve_depth_min.wgsl:19-27 (nearest = 1.0, min over the samples).
__global__ __launch_bounds__(256) void k_hiz_head(const unsigned long long *__restrict__ vis, float *__restrict__ pyr,
                                                  r3n_hiz_desc d, uint32_t levels, uint32_t samples) {
    __shared__ float t[16][17];
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    const uint32_t x0 = (blockIdx.x * 16u + tx) * 2u, y0 = (blockIdx.y * 16u + ty) * 2u;
    float q[2][2];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const uint32_t x = x0 + (uint32_t)dx, y = y0 + (uint32_t)dy;
            float v = 0.0f;
            if (x < d.width && y < d.height) {
                const size_t pix = (size_t)y * d.width + x;
                if (samples == 1u) {
                    v = __uint_as_float((uint32_t)(vis[pix] >> 32));
                } else {
                    v = 1.0f;
                    for (uint32_t sm = 0; sm < samples; ++sm)
                        v = fminf(v, __uint_as_float((uint32_t)(vis[pix * samples + sm] >> 32)));
                }
                pyr[pix] = v;
            }
            q[dy][dx] = v;
        }
    if (levels == 0u) return;
    // hi_z.wgsl:22-30 order: nearest = 1.0; x outer, y inner
    float m = fminf(fminf(fminf(fminf(1.0f, q[0][0]), q[1][0]), q[0][1]), q[1][1]);
    {
        const uint32_t w1 = mip_dim(d.width, 1), h1 = mip_dim(d.height, 1), x = x0 >> 1, y = y0 >> 1;
        if (x < w1 && y < h1) pyr[d.offset[1] + (size_t)y * w1 + x] = m;
    }
    t[ty][tx] = m;
    for (uint32_t l = 2; l <= levels; ++l) {
        __syncthreads();
        const uint32_t n = 16u >> (l - 1u);  // tile edge at level l
        float r = 0.0f;
        const bool active = tx < n && ty < n;
        if (active) {
            const uint32_t s = 1u << (l - 2u);  // stride of level l-1 entries in the LDS tile
            const float a = t[(2u * ty) * s][(2u * tx) * s], b = t[(2u * ty + 1u) * s][(2u * tx) * s];
            const float c = t[(2u * ty) * s][(2u * tx + 1u) * s], e = t[(2u * ty + 1u) * s][(2u * tx + 1u) * s];
            r = fminf(fminf(fminf(fminf(1.0f, a), b), c), e);
        }
        __syncthreads();
        if (active) {
            const uint32_t s = 1u << (l - 1u);
            t[ty * s][tx * s] = r;
            const uint32_t wl = mip_dim(d.width, l), hl = mip_dim(d.height, l);
            const uint32_t x = blockIdx.x * n + tx, y = blockIdx.y * n + ty;
            if (x < wl && y < hl) pyr[d.offset[l] + (size_t)y * wl + x] = r;
        }
    }
}

// Tail of the pyramid (levels first .. mips-1, at most a few thousand texels each): one block walks the levels,
// generic odd-dimension windows, a barrier between levels.  Every level is written to the pyramid in memory; a level
// that fits is ALSO kept in LDS and the next level reads it from there, so the chain of ~10 dependent levels costs
// LDS latencies instead of ~10 global-memory round trips (this kernel sits on the frame's critical path:
// pass-1 raster -> Hi-Z -> cull -> pass-2 raster -> resolve).
#define R3N_HIZ_LDS_A 8192u
#define R3N_HIZ_LDS_B 2304u
__global__ __launch_bounds__(1024) void k_hiz_tail(float *__restrict__ pyr, r3n_hiz_desc d, uint32_t first) {
    __shared__ float lds_a[R3N_HIZ_LDS_A];
    __shared__ float lds_b[R3N_HIZ_LDS_B];
    const float *src_lds = nullptr;  // previous level, when it was kept in LDS
    bool to_a = true;
    for (uint32_t l = first; l < d.mips; ++l) {
        const uint32_t sw = mip_dim(d.width, l - 1u), sh = mip_dim(d.height, l - 1u);
        const uint32_t dw = mip_dim(d.width, l), dh = mip_dim(d.height, l);
        const float *src = pyr + d.offset[l - 1u];
        float *dst = pyr + d.offset[l];
        float *dst_lds = dw * dh <= (to_a ? R3N_HIZ_LDS_A : R3N_HIZ_LDS_B) ? (to_a ? lds_a : lds_b) : nullptr;
        const uint32_t nx = 2u + (sw & 1u), ny = 2u + (sh & 1u);
        for (uint32_t i = threadIdx.x; i < dw * dh; i += 1024u) {
            const uint32_t x = i % dw, y = i / dw;
            float nearest = 1.0f;
            for (uint32_t ix = 0; ix < nx; ++ix)
                for (uint32_t iy = 0; iy < ny; ++iy) {
                    const uint32_t sx = 2u * x + ix, sy = 2u * y + iy;
                    float v = 0.0f;
                    if (sx < sw && sy < sh) v = src_lds ? src_lds[sy * sw + sx] : src[(size_t)sy * sw + sx];
                    nearest = fminf(nearest, v);
                }
            dst[i] = nearest;
            if (dst_lds) dst_lds[i] = nearest;
        }
        __syncthreads();  // the next level reads what this one wrote (same workgroup, same CU)
        src_lds = dst_lds;
        to_a = !to_a;
    }
}

// ------------------------------------------------------------------------------------------------ K6 resolve
struct TriRecord;
struct ShadeArgs {
    const unsigned long long *vis;
    uint32_t width, height, row_begin, row_end;
    const r3n_frame_uniforms496 *fu;
    const r3n_camera_header240 *hdr;
    const r3n_object128 *objects;
    const uint32_t *mesh;
    const r3n_baked128 *baked;
    const r3n_material208 *materials;
    uint32_t n_materials;
    const uint32_t *tri_base;
    const uint32_t *slot_table;  // slot_table[b] = object owning canonical slot b << R3N_SLOT_TABLE_SHIFT
    uint32_t slot_table_size;
    const uint8_t *dir_buf;    // count @0, records @16
    const uint8_t *point_buf;  // count @0, records @16
    const float *atlas;
    uint32_t atlas_w, atlas_h;
    float clear[4];
    ushort4 *hdr_out;          // Rgba16Float
    uchar4 *ldr_out;           // Rgba8UnormSrgb: the tonemap blit fused into the resolve (one HDR round trip less)
    const unsigned char *srgb_lut;
    bool out_bgr;              // Bgra8* output target
    TextureArgs tex;
    ushort4 *samples_out;      // S == 4 and a transparent pass follows: the per-sample colours (else null)
    TriRecord *tri_rec;        // S == 1: per-triangle vertex-stage records by canonical slot (else null)
    unsigned char *seen;       // ... and which slots own a pixel this frame
    uint32_t total_tris;
    // S == 4, split resolve: pixels whose samples belong to more than one triangle hand their extra triangles to a
    // second, dense pass (R3N_EDGEQ sub-lists of pixel << 3 | leader sample << 1 | last-entry-of-the-pixel)
    uint32_t *edge_list, *edge_count;
    uint32_t edge_capacity;    // entries per sub-list
};
#define R3N_EDGEQ 32u

struct LdsDirLight {
    float m[16];      // light.view_proj * uniforms.inv_view (opaque.wgsl:491)
    float l[3];       // normalize(view_mat3 * -direction)   (opaque.wgsl:519)
    float color[3];
    float inv_res[2], offset[2], size[2];
    float sane;       // 1: |colour| <= 1e6 (lets the fragment stage skip fully occluded lights), else 0
};
struct LdsPointLight {
    float vpos[3];    // (uniforms.view * position).xyz (opaque.wgsl:528)
    float color[3];
    float radius;
};

// shadow/pcf.wgsl + comparison sampler (samplers.rs:24,42-57): bilinear, GreaterEqual, Repeat
R3N_DEV float sample_compare(const float *__restrict__ atlas, uint32_t aw, uint32_t ah, float u, float v, float ref,
                             int ox, int oy) {
    const float tx = (u * (float)aw - 0.5f) + (float)ox;
    const float ty = (v * (float)ah - 0.5f) + (float)oy;
    const float fx0 = floorf(tx), fy0 = floorf(ty);
    float fx = tx - fx0, fy = ty - fy0;
    const long long ix = (fx0 == fx0 && fabsf(fx0) < 1e9f) ? (long long)fx0 : 0ll;
    const long long iy = (fy0 == fy0 && fabsf(fy0) < 1e9f) ? (long long)fy0 : 0ll;
    if (!(fx == fx)) fx = 0.0f;
    if (!(fy == fy)) fy = 0.0f;
    const long long w = (long long)aw, h = (long long)ah;
    const uint32_t x0 = (uint32_t)(((ix % w) + w) % w), x1 = (uint32_t)((((ix + 1) % w) + w) % w);
    const uint32_t y0 = (uint32_t)(((iy % h) + h) % h), y1 = (uint32_t)((((iy + 1) % h) + h) % h);
    const float c00 = ref >= atlas[(size_t)y0 * aw + x0] ? 1.0f : 0.0f;
    const float c10 = ref >= atlas[(size_t)y0 * aw + x1] ? 1.0f : 0.0f;
    const float c01 = ref >= atlas[(size_t)y1 * aw + x0] ? 1.0f : 0.0f;
    const float c11 = ref >= atlas[(size_t)y1 * aw + x1] ? 1.0f : 0.0f;
    const float top = c00 * (1.0f - fx) + c10 * fx;
    const float bot = c01 * (1.0f - fx) + c11 * fx;
    return top * (1.0f - fy) + bot * fy;
}

// Texel coordinates + bilinear weights of one comparison tap, exactly as sample_compare derives them.
struct PcfTap {
    int ix, iy;  // |floor| < 1e9 fits
    float fx, fy;
};
R3N_DEV PcfTap pcf_tap(uint32_t aw, uint32_t ah, float u, float v, int ox, int oy) {
    const float tx = (u * (float)aw - 0.5f) + (float)ox;
    const float ty = (v * (float)ah - 0.5f) + (float)oy;
    const float fx0 = floorf(tx), fy0 = floorf(ty);
    PcfTap t;
    t.fx = tx - fx0; t.fy = ty - fy0;
    t.ix = (fx0 == fx0 && fabsf(fx0) < 1e9f) ? (int)fx0 : 0;
    t.iy = (fy0 == fy0 && fabsf(fy0) < 1e9f) ? (int)fy0 : 0;
    if (!(t.fx == t.fx)) t.fx = 0.0f;
    if (!(t.fy == t.fy)) t.fy = 0.0f;
    return t;
}
// Repeat addressing (samplers.rs:24): the texel index is almost always already inside the atlas
R3N_DEV uint32_t wrap_texel(int v, uint32_t n) {
    if ((uint32_t)v < n) return (uint32_t)v;
    const long long w = (long long)n;
    return (uint32_t)((((long long)v % w) + w) % w);
}
R3N_DEV float pcf_texel_cmp(const float *__restrict__ atlas, uint32_t aw, uint32_t ah, int x, int y, float ref) {
    return ref >= atlas[(size_t)wrap_texel(y, ah) * aw + wrap_texel(x, aw)] ? 1.0f : 0.0f;
}

// shadow/pcf.wgsl: mean of 5 bilinear comparison taps (centre, +-1 texel in x and y).  The 5 taps touch 20 texels
// of which only 12 are distinct (a 4x4 block without its corners): the comparisons are fetched once and every tap
// then applies its own weights -- same values, same operation order as five independent sample_compare calls.
R3N_DEV float shadow_pcf5(const float *__restrict__ atlas, uint32_t aw, uint32_t ah, float u, float v, float ref) {
    const PcfTap c = pcf_tap(aw, ah, u, v, 0, 0);
    // cmp[dy][dx] for texel (c.ix - 1 + dx, c.iy - 1 + dy); corners are never needed on the regular path
    uint32_t xs[4];
    const float *rows[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        xs[d] = wrap_texel(c.ix - 1 + d, aw);
        rows[d] = atlas + (size_t)wrap_texel(c.iy - 1 + d, ah) * aw;
    }
    float cmp[4][4];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
            const bool corner = (dx == 0 || dx == 3) && (dy == 0 || dy == 3);
            cmp[dy][dx] = corner ? 0.0f : (ref >= rows[dy][xs[dx]] ? 1.0f : 0.0f);
        }
    const int offs[5][2] = {{0, 0}, {0, 1}, {0, -1}, {1, 0}, {-1, 0}};
    float r = 0.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const PcfTap t = k == 0 ? c : pcf_tap(aw, ah, u, v, offs[k][0], offs[k][1]);
        const int rx = t.ix - c.ix + 1, ry = t.iy - c.iy + 1;
        float c00, c10, c01, c11;
        // regular case: the tap's 2x2 footprint lies inside the fetched block (and off its corners)
        if (rx == offs[k][0] + 1 && ry == offs[k][1] + 1) {
            c00 = cmp[offs[k][1] + 1][offs[k][0] + 1]; c10 = cmp[offs[k][1] + 1][offs[k][0] + 2];
            c01 = cmp[offs[k][1] + 2][offs[k][0] + 1]; c11 = cmp[offs[k][1] + 2][offs[k][0] + 2];
        } else {  // adding the integer offset rounded across a texel boundary (or NaN input): fetch directly
            c00 = pcf_texel_cmp(atlas, aw, ah, t.ix, t.iy, ref);     c10 = pcf_texel_cmp(atlas, aw, ah, t.ix + 1, t.iy, ref);
            c01 = pcf_texel_cmp(atlas, aw, ah, t.ix, t.iy + 1, ref); c11 = pcf_texel_cmp(atlas, aw, ah, t.ix + 1, t.iy + 1, ref);
        }
        const float top = c00 * (1.0f - t.fx) + c10 * t.fx;
        const float bot = c01 * (1.0f - t.fx) + c11 * t.fx;
        r = r + (top * (1.0f - t.fy) + bot * t.fy);
    }
    return r * 0.2f;
}

struct PixelData {
    float albedo[4], diffuse[3], roughness, normal[3], f0[3], emissive[3], ao;
};

#define R3N_PI 3.14159265359f

// opaque.wgsl:440-468
R3N_DEV void surface_shading(const float l[3], const float intensity[3], const PixelData &px, const float v[3],
                             float occlusion, float out[3]) {
    float h[3] = {v[0] + l[0], v[1] + l[1], v[2] + l[2]};
    normalize3(h);
    const float nov = fabsf(dot3(px.normal, v)) + 0.00001f;
    const float nol = sat(dot3(px.normal, l));
    const float noh = sat(dot3(px.normal, h));
    const float loh = sat(dot3(l, h));
    const float c165[3] = {16.5f, 16.5f, 16.5f};
    const float f90 = sat(dot3(px.f0, c165));
    const float a = px.roughness, a2 = a * a;
    const float f = (noh * a2 - noh) * noh + 1.0f;
    const float d = a2 / ((R3N_PI * f) * f);
    const float x = 1.0f - loh, x2 = x * x, x5 = (x2 * x2) * x;
    const float ggxl = nov * sqrtf((-nol * a2 + nol) * nol + a2);
    const float ggxv = nol * sqrtf((-nov * a2 + nov) * nov + a2);
    const float vis = 0.5f / (ggxl + ggxv);
    const float k = nol * occlusion;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float fres = px.f0[c] + (f90 - px.f0[c]) * x5;
        const float fr = (d * vis) * fres;
        const float fd = px.diffuse[c] * (1.0f / R3N_PI);
        const float color = fd + fr;
        out[c] = (color * intensity[c]) * k;
    }
}

R3N_DEV float srgb_to_linear(float e) { return e > 0.04045f ? powf((e + 0.055f) / 1.055f, 2.4f) : e / 12.92f; }

R3N_DEV float srgb_oetf(float x) {
    if (!(x > 0.0f)) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    if (x <= 0.0031308f) return x * 12.92f;
    return 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
}
// The HDR target is Rgba16Float, so the OETF input is one of 65536 half values and only those in (0, 1) need the
// formula: 15360 bit patterns.  k_build_srgb_lut evaluates the exact expression once per pattern at context
// creation; the per-pixel path is then a byte gather instead of three powf (measured: 73 us of the 4K resolve).
#define R3N_SRGB_LUT_SIZE 0x3C00u  // half bits of 1.0
__global__ __launch_bounds__(256) void k_build_srgb_lut(unsigned char *__restrict__ lut) {
    const uint32_t h = blockIdx.x * 256u + threadIdx.x;
    if (h >= R3N_SRGB_LUT_SIZE) return;
    const float x = (float)__builtin_bit_cast(_Float16, (unsigned short)h);
    lut[h] = (unsigned char)(srgb_oetf(x) * 255.0f + 0.5f);
}
// (unsigned char)(srgb_oetf(x) * 255 + 0.5) for the half with bit pattern h
R3N_DEV unsigned char srgb8_of_half(const unsigned char *__restrict__ lut, unsigned short h) {
    if (h & 0x8000u) return 0;       // negative, -0, negative NaN: !(x > 0)
    if (h > 0x7C00u) return 0;       // NaN
    if (h >= R3N_SRGB_LUT_SIZE) return 255;  // x >= 1 (and +inf)
    return lut[h];
}
// blit.wgsl fs_main_scene into an Rgba8UnormSrgb target: exact OETF of the Rgba16Float-rounded value
// bgr: the target is a Bgra8* format (blue first in memory)
R3N_DEV uchar4 tonemap_half4(const unsigned char *__restrict__ lut, ushort4 h, bool bgr = false) {
    const float al = (float)__builtin_bit_cast(_Float16, h.w);
    const float a = (!(al > 0.0f)) ? 0.0f : (al >= 1.0f ? 1.0f : al);
    const unsigned char r = srgb8_of_half(lut, h.x), g = srgb8_of_half(lut, h.y), b = srgb8_of_half(lut, h.z);
    return make_uchar4(bgr ? b : r, g, bgr ? r : b, (unsigned char)(a * 255.0f + 0.5f));
}
// math/color.wgsl:13-19 srgb_scene_to_display + the unorm store's clamp: what blit.wgsl fs_main_monitor writes into a
// target whose format is not *Srgb (tonemapping.rs:44)
R3N_DEV float srgb_scene_to_display(float x) {
    const float e = x > 0.0031308f ? 1.055f * powf(x, 0.4166f) - 0.055f : x * 12.92f;
    if (!(e > 0.0f)) return 0.0f;
    return e >= 1.0f ? 1.0f : e;
}

R3N_DEV ushort4 pack_half4(const float v[4]) {
    ushort4 o;
    // float -> half conversion rounds to nearest even (v_cvt_f16_f32)
    o.x = __builtin_bit_cast(unsigned short, (_Float16)v[0]);
    o.y = __builtin_bit_cast(unsigned short, (_Float16)v[1]);
    o.z = __builtin_bit_cast(unsigned short, (_Float16)v[2]);
    o.w = __builtin_bit_cast(unsigned short, (_Float16)v[3]);
    return o;
}

// opaque.wgsl VS (:91-135) + FS (:203-551) for triangle slot `id - 1` at the centre of pixel (x, y).
// What the vertex stage (opaque.wgsl:91-135) and the triangle setup produce for one triangle: everything the fragment
// stage needs that does not depend on the pixel.  64 floats = 256 B.  With one sample per pixel the resolve does not
// recompute this per pixel: k_mark_visible flags the triangles that own a pixel, k_vertex_stage evaluates the record once
// per flagged triangle, the per-pixel kernel loads it (neighbouring pixels share it).  Same arithmetic either way.
struct TriRecord {
    float e[3][3];      // oriented edge functions of the triangle setup
    float vp[3][4];     // view-space positions
    float vn[3][3];     // view-space normals (normalised per vertex)
    float vt[3][3];     // view-space tangents (only when the material has a normal map, else 0)
    float vc[3][4];     // vertex colours
    float uv[3][2];     // texture coordinates 0
    uint32_t object, material;
    uint32_t _pad[5];
};
static_assert(sizeof(TriRecord) == 256, "triangle record is 64 dwords");

template <bool TEX>
R3N_DEV void vertex_stage(const ShadeArgs &a, uint32_t id, TriRecord &r) {
    const uint32_t slot = id - 1u;
    // object = last o with tri_base[o] <= slot; the coarse table narrows the binary search to the objects that
    // start inside one 256-slot bucket (usually zero or one step instead of log2(capacity))
    const uint32_t bucket = slot >> R3N_SLOT_TABLE_SHIFT;
    uint32_t lo = a.slot_table[bucket];
    uint32_t hi = bucket + 1u < a.slot_table_size ? a.slot_table[bucket + 1u] + 1u : a.hdr->object_count;
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (a.tri_base[mid] <= slot) lo = mid; else hi = mid;
    }
    const uint32_t obj = lo, tri = slot - a.tri_base[obj];
    const r3n_object128 &ob = a.objects[obj];
    const uint32_t mat_index = ob.material_index < a.n_materials ? ob.material_index : 0u;
    const r3n_material208 &mat = a.materials[mat_index];
    const float *mv = a.baked[obj].model_view;
    r.object = obj;
    r.material = mat_index;

    // vertex stage for the 3 vertices (opaque.wgsl:114-134)
    uint32_t idx[3];
    float p[3][4];
    const float inv_s2[3] = {1.0f / dot3(mv, mv), 1.0f / dot3(mv + 4, mv + 4), 1.0f / dot3(mv + 8, mv + 8)};
    const uint32_t first = ob.first_index + tri * 3u;
    const uint32_t pos_off = ob.vertex_attribute_start_offsets[0];
    const uint32_t nrm_off = ob.vertex_attribute_start_offsets[1];
    const uint32_t col_off = ob.vertex_attribute_start_offsets[5];
    bool any_tex = false;
    if (TEX) {
#pragma unroll
        for (int k = 0; k < 10; ++k) any_tex = any_tex || mat.textures[k] != 0u;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        idx[k] = a.mesh[first + (uint32_t)k];
        float v[3];
        fetch_vec3(a.mesh, pos_off, idx[k], v);
        mul_point(a.baked[obj].model_view_proj, v, p[k]);
        mul_point(mv, v, r.vp[k]);
        float nm[3] = {0.0f, 0.0f, 0.0f};
        if (nrm_off != R3N_INVALID) fetch_vec3(a.mesh, nrm_off, idx[k], nm);
        const float sn[3] = {inv_s2[0] * nm[0], inv_s2[1] * nm[1], inv_s2[2] * nm[2]};
        mat3_mul_vec3(mv, mv + 4, mv + 8, sn, r.vn[k]);
        normalize3(r.vn[k]);
        if (TEX && mat.textures[1] != 0u) {  // vs_out.tangent (opaque.wgsl:129); only the normal map reads it
            float tg[3] = {0.0f, 0.0f, 0.0f};
            const uint32_t tan_off = ob.vertex_attribute_start_offsets[2];
            if (tan_off != R3N_INVALID) fetch_vec3(a.mesh, tan_off, idx[k], tg);
            const float st[3] = {inv_s2[0] * tg[0], inv_s2[1] * tg[1], inv_s2[2] * tg[2]};
            mat3_mul_vec3(mv, mv + 4, mv + 8, st, r.vt[k]);
            normalize3(r.vt[k]);
        } else {
            r.vt[k][0] = r.vt[k][1] = r.vt[k][2] = 0.0f;
        }
        if (col_off != R3N_INVALID) {
            const uint32_t cw = a.mesh[col_off / 4u + idx[k]];
#pragma unroll
            for (int c = 0; c < 4; ++c) r.vc[k][c] = (float)((cw >> (8 * c)) & 0xFFu) / 255.0f;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) r.vc[k][c] = 1.0f;
        }
        if (TEX && any_tex) fetch_uv0(a.mesh, ob.vertex_attribute_start_offsets[3], idx[k], r.uv[k]);
        else r.uv[k][0] = r.uv[k][1] = 0.0f;
    }
    TriSetup ts;
    setup_triangle(p, (float)a.width / 2.0f, (float)a.height / 2.0f,
                   (a.hdr->flags & R3N_PCU_POSITIVE_AREA_VISIBLE) != 0u, ts);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) r.e[i][c] = ts.e[i][c];
}

// opaque.wgsl FS (:203-551) for the triangle record `r` at the centre of pixel (x, y).
template <bool TEX>
R3N_DEV void fragment_stage(const ShadeArgs &a, const LdsDirLight *s_dir, const LdsPointLight *s_point, uint32_t n_dir,
                            uint32_t n_point, const TriRecord &r, uint32_t x, uint32_t y, float out[4]) {
    const r3n_material208 &mat = a.materials[r.material];
    TriSetup ts;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) ts.e[i][c] = r.e[i][c];
    ts.z[0] = ts.z[1] = ts.z[2] = 0.0f; ts.det = 1.0f; ts.valid = true;  // not used by the fragment stage
    float E[3];
    (void)edge_eval(ts, (float)x + 0.5f, (float)y + 0.5f, E);
    const float rs = 1.0f / ((E[0] + E[1]) + E[2]);
    const float lam[3] = {E[0] * rs, E[1] * rs, E[2] * rs};
    float vpos[4], nrm[3], col[4] = {1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
    for (int c = 0; c < 4; ++c) vpos[c] = (lam[0] * r.vp[0][c] + lam[1] * r.vp[1][c]) + lam[2] * r.vp[2][c];
#pragma unroll
    for (int c = 0; c < 3; ++c) nrm[c] = (lam[0] * r.vn[0][c] + lam[1] * r.vn[1][c]) + lam[2] * r.vn[2][c];
    if ((mat.flags & R3N_FLAGS_ALBEDO_ACTIVE) && (mat.flags & R3N_FLAGS_ALBEDO_BLEND)) {  // the only reader of vs_out.color
#pragma unroll
        for (int c = 0; c < 4; ++c) col[c] = (lam[0] * r.vc[0][c] + lam[1] * r.vc[1][c]) + lam[2] * r.vc[2][c];
    }

    // fragment stage (opaque.wgsl:203-424).  Texture slots (managers/material.rs:25-29 order): 0 albedo, 1 normal,
    // 2 roughness, 3 metallic, 4 reflectance, 5 clear coat, 6 clear coat roughness, 7 emissive, 8 anisotropy, 9 AO
    PixelData px;
    const uint32_t mflags = mat.flags;
    bool any_tex = false;
    if (TEX) {
#pragma unroll
        for (int k = 0; k < 10; ++k) any_tex = any_tex || mat.textures[k] != 0u;
    }
    float coords[2] = {0.0f, 0.0f}, ddx[2] = {0.0f, 0.0f}, ddy[2] = {0.0f, 0.0f};
    const bool nearest = (mflags & R3N_FLAGS_NEAREST) != 0u;
    if (TEX && any_tex) {  // opaque.wgsl:207-209
        const float self_raw[2] = {(lam[0] * r.uv[0][0] + lam[1] * r.uv[1][0]) + lam[2] * r.uv[2][0],
                                   (lam[0] * r.uv[0][1] + lam[1] * r.uv[1][1]) + lam[2] * r.uv[2][1]};
        frag_coords(ts, r.uv, mat.uv_transform0, (int)x, (int)y, coords, ddx, ddy, self_raw);
    }
    auto tex = [&](int slot, float dst[4]) { tex_sample_grad(a.tex, mat.textures[slot], nearest, coords[0], coords[1], ddx, ddy, dst); };
    auto has = [&](int slot) { return TEX && mat.textures[slot] != 0u; };
    if (mflags & R3N_FLAGS_ALBEDO_ACTIVE) {
#pragma unroll
        for (int c = 0; c < 4; ++c) px.albedo[c] = 1.0f;
        if (has(0)) tex(0, px.albedo);
        if (mflags & R3N_FLAGS_ALBEDO_BLEND) {
            if (mflags & R3N_FLAGS_ALBEDO_VERTEX_SRGB) {
#pragma unroll
                for (int c = 0; c < 3; ++c) px.albedo[c] *= srgb_to_linear(col[c]);
                px.albedo[3] *= col[3];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) px.albedo[c] *= col[c];
            }
        }
    } else {
        px.albedo[0] = px.albedo[1] = px.albedo[2] = 0.0f;
        px.albedo[3] = 1.0f;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) px.albedo[c] *= mat.albedo[c];

    if (mflags & R3N_FLAGS_UNLIT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = px.albedo[c];
    } else {
        // --- normal (opaque.wgsl:246-273)
        if (has(1)) {
            float t[4], n[3];
            tex(1, t);
            if (mflags & R3N_FLAGS_BICOMPONENT_NORMAL) {
                float b0 = (mflags & R3N_FLAGS_SWIZZLED_NORMAL) ? t[3] : t[0], b1 = t[1];  // texture_read.ag : .rg
                b0 = b0 * 2.0f - 1.0f;
                b1 = b1 * 2.0f - 1.0f;
                n[0] = b0; n[1] = b1;
                n[2] = sqrtf((1.0f - b0 * b0) - b1 * b1);
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) n[c] = t[c] * 2.0f - 1.0f;
                normalize3(n);
            }
            if (mflags & R3N_FLAGS_YDOWN_NORMAL) n[1] = -n[1];
            float tng[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) tng[c] = (lam[0] * r.vt[0][c] + lam[1] * r.vt[1][c]) + lam[2] * r.vt[2][c];
            float nn[3] = {nrm[0], nrm[1], nrm[2]};
            normalize3(nn);
            normalize3(tng);
            const float bt[3] = {nn[1] * tng[2] - tng[1] * nn[2], nn[2] * tng[0] - tng[2] * nn[0], nn[0] * tng[1] - tng[0] * nn[1]};
            mat3_mul_vec3(tng, bt, nn, n, px.normal);  // tbn * normal
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) px.normal[c] = nrm[c];
        }
        normalize3(px.normal);
        // --- AO, metallic, roughness (opaque.wgsl:277-351)
        float ao = mat.ambient_occlusion, pr = mat.roughness, metallic = mat.metallic;
        if (mflags & R3N_FLAGS_AOMR_COMBINED) {
            if (has(2)) {
                float t[4];
                tex(2, t);
                ao = mat.ambient_occlusion * t[0];
                pr = mat.roughness * t[1];
                metallic = mat.metallic * t[2];
            }
        } else if (mflags & R3N_FLAGS_AOMR_BW_SPLIT) {
            float t[4];
            if (has(2)) { tex(2, t); pr = mat.roughness * t[0]; }
            if (has(3)) { tex(3, t); metallic = mat.metallic * t[0]; }
            if (has(9)) { tex(9, t); ao = mat.ambient_occlusion * t[0]; }
        } else {
            float t[4];
            if (has(2)) {
                tex(2, t);
                const bool sw = (mflags & R3N_FLAGS_AOMR_SWIZZLED_SPLIT) != 0u;
                pr = mat.roughness * (sw ? t[1] : t[0]);
                metallic = mat.metallic * (sw ? t[2] : t[1]);
            }
            if (has(9)) { tex(9, t); ao = mat.ambient_occlusion * t[0]; }
        }
        // --- reflectance (opaque.wgsl:355-359)
        float reflectance = mat.reflectance;
        if (has(4)) { float t[4]; tex(4, t); reflectance = mat.reflectance * t[0]; }
        // --- clear coat (opaque.wgsl:363-391)
        float cc = mat.clear_coat, ccpr = mat.clear_coat_roughness;
        if (mflags & R3N_FLAGS_CC_GLTF_COMBINED) {
            if (has(5)) {
                float t[4];
                tex(5, t);
                cc = mat.clear_coat * t[0];
                ccpr = mat.clear_coat_roughness * t[1];
            }
        } else {
            float t[4];
            if (has(5)) { tex(5, t); cc = mat.clear_coat * t[0]; }
            if (has(6)) {
                tex(6, t);
                ccpr = mat.clear_coat_roughness * ((mflags & R3N_FLAGS_CC_GLTF_SPLIT) ? t[1] : t[0]);
            }
        }
        // --- emissive (opaque.wgsl:395-399); the anisotropy texture (:403-407) feeds nothing downstream
#pragma unroll
        for (int c = 0; c < 3; ++c) px.emissive[c] = mat.emissive[c];
        if (has(7)) {
            float t[4];
            tex(7, t);
#pragma unroll
            for (int c = 0; c < 3; ++c) px.emissive[c] = mat.emissive[c] * t[c];
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) px.diffuse[c] = px.albedo[c] * (1.0f - metallic);
        const float refl = (0.16f * reflectance) * reflectance;
#pragma unroll
        for (int c = 0; c < 3; ++c) px.f0[c] = px.albedo[c] * metallic + (refl * (1.0f - metallic));
        if (cc != 0.0f) {
            const float base_pr = fmaxf(pr, ccpr);
            pr = pr * (1.0f - cc) + base_pr * cc;
        }
        px.roughness = pr * pr;
        px.ao = ao;

        float vv[3] = {vpos[0], vpos[1], vpos[2]};
        normalize3(vv);
#pragma unroll
        for (int c = 0; c < 3; ++c) vv[c] = -vv[c];
        float color[3] = {px.emissive[0], px.emissive[1], px.emissive[2]};
#if R3N_SKIP_OCCLUDED
        // A fully occluded light (shadow * ao == 0) adds (finite) * 0 = +-0 when every factor of surface_shading is
        // finite: skip its BRDF.  Finite is guaranteed by: all pixel inputs finite (the sum of magnitudes is finite;
        // NaN fails the comparison), roughness^2 >= 1e-9 (D <= 1/(pi a^2) <= 3.2e17 without underflow of f^2,
        // V <= 0.5/(1e-5 a) <= 5e13, Fresnel <= 2) and |light colour| <= 1e6 (checked when the lights are staged).
        const float mag = (((fabsf(px.normal[0]) + fabsf(px.normal[1])) + (fabsf(px.normal[2]) + fabsf(vv[0]))) +
                           ((fabsf(vv[1]) + fabsf(vv[2])) + (fabsf(px.f0[0]) + fabsf(px.f0[1])))) +
                          (((fabsf(px.f0[2]) + fabsf(px.diffuse[0])) + (fabsf(px.diffuse[1]) + fabsf(px.diffuse[2]))) + fabsf(px.ao));
        const bool skip_ok = px.roughness >= 1e-9f && px.roughness <= 1e9f && mag < 1e30f;
#endif
        for (uint32_t i = 0; i < n_dir; ++i) {
            const LdsDirLight &L = s_dir[i];
            // surface_shading scales by k = nol * occlusion.  With nol == 0 and roughness > 0 every factor is finite
            // (D <= 1/(pi a^2), V <= 0.5/(nov a), nov >= 1e-5), so the light adds exactly +0: skip the shadow lookup
            // and the BRDF.  `+= 0.0f` keeps the -0 -> +0 behaviour of the full expression.
            const float nl_raw = dot3(px.normal, L.l);
            if (px.roughness > 0.0f && nl_raw == nl_raw && sat(nl_raw) == 0.0f) {  // (a NaN normal must stay NaN)
#pragma unroll
                for (int c = 0; c < 3; ++c) color[c] += 0.0f;
                continue;
            }
            float sn[4];
            mul_vec4(L.m, vpos[0], vpos[1], vpos[2], vpos[3], sn);
            const float fl[2] = {sn[0] * 0.5f + 0.5f, sn[1] * 0.5f + 0.5f};
            const float local[2] = {fl[0], 1.0f - fl[1]};
            float tl[2] = {L.offset[0], L.offset[1]};
            float tr[2] = {tl[0] + L.size[0], tl[1] + L.size[1]};
            const float coords[2] = {tl[0] * (1.0f - local[0]) + tr[0] * local[0],
                                     tl[1] * (1.0f - local[1]) + tr[1] * local[1]};
            const float border[2] = {L.inv_res[0] * 1.5f, L.inv_res[1] * 1.5f};
            tl[0] += border[0]; tl[1] += border[1];
            tr[0] -= border[0]; tr[1] -= border[1];
            float shadow = 1.0f;
            // opaque.wgsl:509-514 (quirk: `any`, un-atlased coords vs atlas-space bounds -- reproduced)
            if ((fl[0] >= tl[0] || fl[1] >= tl[1]) && (fl[0] <= tr[0] || fl[1] <= tr[1]) && sn[2] >= 0.0f && sn[2] <= 1.0f)
                shadow = shadow_pcf5(a.atlas, a.atlas_w, a.atlas_h, coords[0], coords[1], sn[2]);
#if R3N_SKIP_OCCLUDED
            if (skip_ok && L.sane != 0.0f && shadow * px.ao == 0.0f) {
#pragma unroll
                for (int c = 0; c < 3; ++c) color[c] += 0.0f;
                continue;
            }
#endif
            float res[3];
            surface_shading(L.l, L.color, px, vv, shadow * px.ao, res);
#pragma unroll
            for (int c = 0; c < 3; ++c) color[c] += res[c];
        }
        for (uint32_t i = 0; i < n_point; ++i) {
            const LdsPointLight &P = s_point[i];
            const float delta[3] = {P.vpos[0] - vpos[0], P.vpos[1] - vpos[1], P.vpos[2] - vpos[2]};
            const float d = sqrtf(dot3(delta, delta));
            const float s = sat(d / P.radius);
            const float s2 = s * s, is2 = 1.0f - s2;
            const float att = is2 * is2 / (1.0f + s2);
            const float inten[3] = {P.color[0] * att, P.color[1] * att, P.color[2] * att};
            const float l[3] = {delta[0] / d, delta[1] / d, delta[2] / d};
            float res[3];
            surface_shading(l, inten, px, vv, px.ao, res);
#pragma unroll
            for (int c = 0; c < 3; ++c) color[c] += (res[c] > 0.0f ? res[c] : 0.0f);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = fmaxf(a.fu->ambient[c] * px.albedo[c], color[c]);
        out[3] = fmaxf(a.fu->ambient[3] * px.albedo[3], px.albedo[3]);
    }
}

template <bool TEX>
R3N_DEV void shade_fragment(const ShadeArgs &a, const LdsDirLight *s_dir, const LdsPointLight *s_point, uint32_t n_dir,
                            uint32_t n_point, uint32_t id, uint32_t x, uint32_t y, float out[4]) {
    TriRecord r;
    vertex_stage<TEX>(a, id, r);
    fragment_stage<TEX>(a, s_dir, s_point, n_dir, n_point, r, x, y, out);
}

// Flags the triangles that own at least one pixel (plain byte stores: every writer writes 1).
__global__ __launch_bounds__(256) void k_mark_visible(const unsigned long long *__restrict__ vis, unsigned char *__restrict__ seen,
                                                      size_t first_pixel, size_t n_pixels) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pixels) return;
    const uint32_t id = (uint32_t)(vis[first_pixel + i] & 0xFFFFFFFFull);
    if (id != 0u) seen[id - 1u] = 1;
}
// One thread per canonical triangle slot: the vertex stage + setup of the flagged ones, once per frame.
template <bool TEX>
__global__ __launch_bounds__(256) void k_vertex_stage(ShadeArgs a) {
    const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
    if (slot >= a.total_tris || !a.seen[slot]) return;
    TriRecord r;
    vertex_stage<TEX>(a, slot + 1u, r);
    r._pad[0] = r._pad[1] = r._pad[2] = r._pad[3] = r._pad[4] = 0u;
    a.tri_rec[slot] = r;
}

// The light list in view space, once per workgroup (LDS): matrices light.view_proj * uniforms.inv_view, directions,
// point-light positions.  Ends with a barrier.
R3N_DEV void stage_lights(const ShadeArgs &a, LdsDirLight *s_dir, LdsPointLight *s_point, uint32_t &n_dir, uint32_t &n_point) {
    n_dir = min(*reinterpret_cast<const uint32_t *>(a.dir_buf), (uint32_t)R3N_MAX_DIR_LIGHTS);
    n_point = min(*reinterpret_cast<const uint32_t *>(a.point_buf), (uint32_t)R3N_MAX_POINT_LIGHTS);

    const r3n_dir_light128 *dirs = reinterpret_cast<const r3n_dir_light128 *>(a.dir_buf + 16);
    const r3n_point_light32 *points = reinterpret_cast<const r3n_point_light32 *>(a.point_buf + 16);
    for (uint32_t i = threadIdx.x; i < n_dir * 4u; i += 256u) {
        const uint32_t li = i >> 2, c = i & 3u;
        const float *col = a.fu->inv_view + 4 * c;  // column c of (view_proj * inv_view)
        float o[4];
        mul_vec4(dirs[li].view_proj, col[0], col[1], col[2], col[3], o);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_dir[li].m[4 * c + r] = o[r];
        if (c == 0) {
            const float nd[3] = {-dirs[li].direction[0], -dirs[li].direction[1], -dirs[li].direction[2]};
            float l[3];
            mat3_mul_vec3(a.fu->view, a.fu->view + 4, a.fu->view + 8, nd, l);
            normalize3(l);
#pragma unroll
            for (int r = 0; r < 3; ++r) { s_dir[li].l[r] = l[r]; s_dir[li].color[r] = dirs[li].color[r]; }
            s_dir[li].sane = (fabsf(dirs[li].color[0]) <= 1e6f && fabsf(dirs[li].color[1]) <= 1e6f && fabsf(dirs[li].color[2]) <= 1e6f &&
                              fabsf(l[0]) <= 2.0f && fabsf(l[1]) <= 2.0f && fabsf(l[2]) <= 2.0f) ? 1.0f : 0.0f;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                s_dir[li].inv_res[r] = dirs[li].inv_resolution[r];
                s_dir[li].offset[r] = dirs[li].atlas_offset[r];
                s_dir[li].size[r] = dirs[li].atlas_size[r];
            }
        }
    }
    for (uint32_t i = threadIdx.x; i < n_point; i += 256u) {
        float o[4];
        mul_vec4(a.fu->view, points[i].position[0], points[i].position[1], points[i].position[2], points[i].position[3], o);
#pragma unroll
        for (int r = 0; r < 3; ++r) { s_point[i].vpos[r] = o[r]; s_point[i].color[r] = points[i].color[r]; }
        s_point[i].radius = points[i].radius;
    }
    __syncthreads();
}

// One thread per pixel, 16x16 pixel tiles; the light list is transformed once per workgroup and staged in LDS.
// S = samples per pixel.  S == 4: every sample of the multisampled Rgba16Float target holds the half-rounded colour
// of its nearest fragment (shaded once per distinct triangle, at the pixel centre) or the clear colour; the render
// pass resolve (base.rs:245-258) is their box average ((s0 + s1) + (s2 + s3)) * 0.25.
// Register budget: the untextured single-sample variant is VALU-bound and measurably faster at 5 waves per SIMD
// (<= 96 VGPRs: 347 vs 375 us on the bench scene) -- the second launch-bound asks for that.
// REC: the per-triangle records exist: no vertex-stage code in the kernel at all.
template <int S, bool TEX, bool REC = false, bool SPLIT = false>
__global__ __launch_bounds__(256, (S == 1 && !TEX) ? 5 : (REC ? (S == 1 ? R3N_TEX_OCC : R3N_MS_OCC) : 1)) void k_resolve_opaque(ShadeArgs a) {
    __shared__ LdsDirLight s_dir[R3N_MAX_DIR_LIGHTS];
    __shared__ LdsPointLight s_point[R3N_MAX_POINT_LIGHTS];
    __shared__ float s_decode[512];
    if (TEX) {  // texel decode tables into LDS (texture.h)
        s_decode[threadIdx.x] = a.tex.decode[threadIdx.x];
        s_decode[256u + threadIdx.x] = a.tex.decode[256u + threadIdx.x];
        a.tex.decode = s_decode;
    }
    uint32_t n_dir, n_point;
    stage_lights(a, s_dir, s_point, n_dir, n_point);

    // each wavefront shades an 8x8 pixel quad of the 16x16 tile (fewer distinct triangles / atlas texels per wave
    // than a 16x4 strip; measured 3 % faster)
    const uint32_t wv = threadIdx.x >> 6, ln = threadIdx.x & 63u;
#if R3N_XCD_REMAP
    // Workgroups are dealt to the 8 XCDs round-robin by linear id; remap so that every XCD shades a contiguous band of
    // tiles (its L2 then holds one band's triangle records, texels and shadow texels instead of a slice of all of them).
    const uint32_t nb = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t per = (nb + 7u) / 8u;
    uint32_t tile = (lin & 7u) * per + (lin >> 3);
    if (tile >= nb) tile = lin;  // (only when nb is not a multiple of 8: the tail keeps its place)
    const uint32_t bx = tile % gridDim.x, by = tile / gridDim.x;
#else
    const uint32_t bx = blockIdx.x, by = blockIdx.y;
#endif
    const uint32_t x = bx * 16u + (ln & 7u) + 8u * (wv & 1u);
    const uint32_t y = a.row_begin + by * 16u + (ln >> 3) + 8u * (wv >> 1);
    const bool inside = x < a.width && y < a.row_end;
    if (!inside && !SPLIT) return;  // SPLIT: every thread of the workgroup takes part in the queue reservation below
    const size_t pix = inside ? (size_t)y * a.width + x : 0u;
    float out[4];
    if (S == 1) {
        const uint32_t id = (uint32_t)(a.vis[pix] & 0xFFFFFFFFull);
        if (id == 0u) {
            const ushort4 hc = pack_half4(a.clear);
            a.hdr_out[pix] = hc;
            a.ldr_out[pix] = tonemap_half4(a.srgb_lut, hc, a.out_bgr);
            return;
        }
        if (REC) fragment_stage<TEX>(a, s_dir, s_point, n_dir, n_point, a.tri_rec[id - 1u], x, y, out);
        else shade_fragment<TEX>(a, s_dir, s_point, n_dir, n_point, id, x, y, out);
    } else {
        uint32_t ids[S];
        float col[S][4];
#pragma unroll
        for (int sm = 0; sm < S; ++sm) ids[sm] = inside ? (uint32_t)(a.vis[pix * (size_t)S + (size_t)sm] & 0xFFFFFFFFull) : 0u;
        // The distinct triangles among the pixel's samples, each shaded ONCE (same triangle, same pixel centre: same
        // value) by one copy of the fragment stage in a rolled loop: unrolling it per sample made the kernel four
        // fragment stages long (instruction cache, registers) although interior pixels hold one triangle.
        uint32_t first_of[S];  // index of the first sample with the same id
        uint32_t n_unique = 0;
#pragma unroll
        for (int sm = 0; sm < S; ++sm) {
            uint32_t f = (uint32_t)sm;
#pragma unroll
            for (int p = sm - 1; p >= 0; --p)
                if (ids[p] == ids[sm]) f = (uint32_t)p;
            first_of[sm] = f;
            n_unique += f == (uint32_t)sm ? 1u : 0u;
        }
#pragma unroll
        for (int sm = 0; sm < S; ++sm)
#pragma unroll
            for (int c = 0; c < 4; ++c) col[sm][c] = 0.0f;
        // SPLIT: this kernel shades only the triangle of sample 0 -- every lane busy once -- and queues the pixel's
        // other triangles (edge pixels, a minority) for k_resolve_edges, which runs them densely; k_resolve_edge_pixels
        // then averages.  Unsplit, a wavefront pays a whole fragment stage for every extra triangle of its worst pixel.
        uint32_t edge_base = 0;
        bool edge_fits = true;
        if (SPLIT) {
            __shared__ uint32_t s_extra, s_base;
            if (threadIdx.x == 0u) s_extra = 0u;
            __syncthreads();
            const uint32_t extra = n_unique - 1u;
            uint32_t my_off = 0;
            if (extra) my_off = atomicAdd(&s_extra, extra);
            __syncthreads();
            const uint32_t q = (blockIdx.y * gridDim.x + blockIdx.x) % R3N_EDGEQ;
            if (threadIdx.x == 0u && s_extra) s_base = atomicAdd(&a.edge_count[q], s_extra);
            __syncthreads();
            edge_base = s_base + my_off;
            edge_fits = extra == 0u || edge_base + extra <= a.edge_capacity;  // else: shade everything here (never drop work)
            if (extra && !edge_fits) {  // the slots this pixel reserved inside the list stay empty
                for (uint32_t k = edge_base; k < min(edge_base + extra, a.edge_capacity); ++k)
                    a.edge_list[(size_t)q * a.edge_capacity + k] = 0xFFFFFFFFu;
            }
            if (extra && edge_fits) {
                uint32_t *dst = a.edge_list + (size_t)q * a.edge_capacity + edge_base;
                uint32_t last = 0;
#pragma unroll
                for (int sm = 1; sm < S; ++sm)
                    if (first_of[sm] == (uint32_t)sm) last = (uint32_t)sm;
                uint32_t w = 0;
#pragma unroll
                for (int sm = 1; sm < S; ++sm)
                    if (first_of[sm] == (uint32_t)sm) dst[w++] = ((uint32_t)pix << 3) | ((uint32_t)sm << 1) | (last == (uint32_t)sm ? 1u : 0u);
            }
        }
        if (!inside) return;  // (after the workgroup barriers)
        if (SPLIT && edge_fits) {
            // the common case, kept lean: only the first triangle is shaded here, so nothing per sample has to stay in
            // registers across the fragment stage except which samples it owns
            uint32_t mask0 = 0;
#pragma unroll
            for (int sm = 0; sm < S; ++sm) mask0 |= first_of[sm] == 0u ? 1u << sm : 0u;
            const uint32_t id0 = ids[0];
            const bool single = n_unique == 1u;
            float v[4];
            if (id0 == 0u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = a.clear[c];
            } else if (REC) {
                fragment_stage<TEX>(a, s_dir, s_point, n_dir, n_point, a.tri_rec[id0 - 1u], x, y, v);
            } else {
                shade_fragment<TEX>(a, s_dir, s_point, n_dir, n_point, id0, x, y, v);
            }
            const ushort4 h = pack_half4(v);
            if (!single) {  // edge pixel: park the samples of the first triangle; the other passes finish the pixel
#pragma unroll
                for (int sm = 0; sm < S; ++sm)
                    if ((mask0 >> sm) & 1u) a.samples_out[pix * (size_t)S + (size_t)sm] = h;
                return;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {  // box resolve of four equal samples, same expression as everywhere
                const float cf = (float)(_Float16)v[c];
                out[c] = ((cf + cf) + (cf + cf)) * 0.25f;
            }
            const ushort4 ho = pack_half4(out);
            a.hdr_out[pix] = ho;
            a.ldr_out[pix] = tonemap_half4(a.srgb_lut, ho, a.out_bgr);
            return;
        }
        const uint32_t n_here = n_unique;
#pragma unroll 1
        for (uint32_t k = 0, sm_at = 0; k < n_here; ++k, ++sm_at) {
            while (first_of[sm_at == 0u ? 0 : (sm_at == 1u ? 1 : (sm_at == 2u ? 2 : 3))] != sm_at) ++sm_at;  // next leader sample
            const uint32_t id = sm_at == 0u ? ids[0] : (sm_at == 1u ? ids[1] : (sm_at == 2u ? ids[2] : ids[3]));
            float v[4];
            if (id == 0u) {
#pragma unroll
                for (int c = 0; c < 4; ++c) v[c] = a.clear[c];
            } else if (REC) {
                fragment_stage<TEX>(a, s_dir, s_point, n_dir, n_point, a.tri_rec[id - 1u], x, y, v);
            } else {
                shade_fragment<TEX>(a, s_dir, s_point, n_dir, n_point, id, x, y, v);
            }
#pragma unroll
            for (int sm = 0; sm < S; ++sm)
                if (first_of[sm] == sm_at) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) col[sm][c] = (float)(_Float16)v[c];
                }
        }
        if (!SPLIT && a.samples_out != nullptr) {
#pragma unroll
            for (int sm = 0; sm < S; ++sm) a.samples_out[pix * (size_t)S + (size_t)sm] = pack_half4(col[sm]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = ((col[0][c] + col[1][c]) + (col[2][c] + col[3][c])) * 0.25f;
    }
    const ushort4 ho = pack_half4(out);
    a.hdr_out[pix] = ho;
    a.ldr_out[pix] = tonemap_half4(a.srgb_lut, ho, a.out_bgr);
}

// Split MSAA resolve, pass B: one thread per queued (pixel, leader sample): shade that triangle at the pixel centre and
// park the half-rounded colour in every sample it owns.  Pass C (k_resolve_edge_pixels): the entry flagged as its
// pixel's last one averages the four parked samples -- the same box resolve expression as everywhere else.
template <bool TEX, bool REC>
__global__ __launch_bounds__(256, REC ? R3N_TEX_OCC : 1) void k_resolve_edges(ShadeArgs a) {
    __shared__ LdsDirLight s_dir[R3N_MAX_DIR_LIGHTS];
    __shared__ LdsPointLight s_point[R3N_MAX_POINT_LIGHTS];
    __shared__ float s_decode[512];
    if (TEX) {
        s_decode[threadIdx.x] = a.tex.decode[threadIdx.x];
        s_decode[256u + threadIdx.x] = a.tex.decode[256u + threadIdx.x];
        a.tex.decode = s_decode;
    }
    uint32_t n_dir, n_point;
    stage_lights(a, s_dir, s_point, n_dir, n_point);
    const uint32_t q = blockIdx.x % R3N_EDGEQ;
    const uint32_t n = min(a.edge_count[q], a.edge_capacity);
    const uint32_t *list = a.edge_list + (size_t)q * a.edge_capacity;
    const uint32_t stride = (gridDim.x / R3N_EDGEQ) * 256u;
    for (uint32_t i = (blockIdx.x / R3N_EDGEQ) * 256u + threadIdx.x; i < n; i += stride) {
        const uint32_t e = list[i];
        if (e == 0xFFFFFFFFu) continue;  // reserved by a pixel that did not fit and shaded itself
        const size_t pix = e >> 3;
        const uint32_t leader = (e >> 1) & 3u;
        uint32_t ids[4];
#pragma unroll
        for (int sm = 0; sm < 4; ++sm) ids[sm] = (uint32_t)(a.vis[pix * 4u + (size_t)sm] & 0xFFFFFFFFull);
        const uint32_t id = leader == 1u ? ids[1] : (leader == 2u ? ids[2] : ids[3]);
        const uint32_t x = (uint32_t)(pix % a.width), y = (uint32_t)(pix / a.width);
        float v[4];
        if (id == 0u) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = a.clear[c];
        } else if (REC) {
            fragment_stage<TEX>(a, s_dir, s_point, n_dir, n_point, a.tri_rec[id - 1u], x, y, v);
        } else {
            shade_fragment<TEX>(a, s_dir, s_point, n_dir, n_point, id, x, y, v);
        }
        const ushort4 h = pack_half4(v);
#pragma unroll
        for (int sm = 1; sm < 4; ++sm)
            if ((uint32_t)sm >= leader && ids[sm] == id) a.samples_out[pix * 4u + (size_t)sm] = h;  // samples led by `leader`
    }
}
__global__ __launch_bounds__(256) void k_resolve_edge_pixels(ShadeArgs a) {
    const uint32_t q = blockIdx.x % R3N_EDGEQ;
    const uint32_t n = min(a.edge_count[q], a.edge_capacity);
    const uint32_t *list = a.edge_list + (size_t)q * a.edge_capacity;
    const uint32_t stride = (gridDim.x / R3N_EDGEQ) * 256u;
    for (uint32_t i = (blockIdx.x / R3N_EDGEQ) * 256u + threadIdx.x; i < n; i += stride) {
        const uint32_t e = list[i];
        if (e == 0xFFFFFFFFu || !(e & 1u)) continue;
        const size_t pix = e >> 3;
        float col[4][4], out[4];
#pragma unroll
        for (int sm = 0; sm < 4; ++sm) {
            const ushort4 h = a.samples_out[pix * 4u + (size_t)sm];
            col[sm][0] = (float)__builtin_bit_cast(_Float16, h.x); col[sm][1] = (float)__builtin_bit_cast(_Float16, h.y);
            col[sm][2] = (float)__builtin_bit_cast(_Float16, h.z); col[sm][3] = (float)__builtin_bit_cast(_Float16, h.w);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = ((col[0][c] + col[1][c]) + (col[2][c] + col[3][c])) * 0.25f;
        const ushort4 ho = pack_half4(out);
        a.hdr_out[pix] = ho;
        a.ldr_out[pix] = tonemap_half4(a.srgb_lut, ho, a.out_bgr);
    }
}

// ------------------------------------------------------------------------------------------------ transparent pass
// Stage 3 (row N3): the collected fragments, sorted by (pixel sample, draw order).  The thread that owns the first
// fragment of a sample walks that sample's run in order: evaluate the fragment (once per triangle and pixel centre,
// like the forward pass), BlendState::ALPHA_BLENDING on the half-rounded destination -- rgb = src * a + dst * (1 - a),
// alpha = src.a + dst.a * (1 - a) in f32, result rounded to half -- exactly the oracle's sequence.
struct BlendApplyArgs {
    const unsigned long long *keys;
    const uint32_t *vals;
    uint32_t n;
    ushort4 *samples;  // S == 1: the HDR target itself; S == 4: the per-sample colours
};
template <int S, bool TEX>
__global__ __launch_bounds__(256) void k_blend_apply(ShadeArgs a, BlendApplyArgs b) {
    __shared__ LdsDirLight s_dir[R3N_MAX_DIR_LIGHTS];
    __shared__ LdsPointLight s_point[R3N_MAX_POINT_LIGHTS];
    __shared__ float s_decode[512];
    if (TEX) {
        s_decode[threadIdx.x] = a.tex.decode[threadIdx.x];
        s_decode[256u + threadIdx.x] = a.tex.decode[256u + threadIdx.x];
        a.tex.decode = s_decode;
    }
    uint32_t n_dir, n_point;
    stage_lights(a, s_dir, s_point, n_dir, n_point);
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= b.n) return;
    const uint32_t ps = (uint32_t)(b.keys[i] >> 32);
    if (i > 0u && (uint32_t)(b.keys[i - 1u] >> 32) == ps) return;  // not the head of its run
    const uint32_t pix = ps / (uint32_t)S;
    const uint32_t x = pix % a.width, y = pix / a.width;
    const ushort4 d16 = b.samples[ps];
    float d[4] = {(float)__builtin_bit_cast(_Float16, d16.x), (float)__builtin_bit_cast(_Float16, d16.y),
                  (float)__builtin_bit_cast(_Float16, d16.z), (float)__builtin_bit_cast(_Float16, d16.w)};
    for (uint32_t j = i; j < b.n && (uint32_t)(b.keys[j] >> 32) == ps; ++j) {
        float src[4];
        shade_fragment<TEX>(a, s_dir, s_point, n_dir, n_point, b.vals[j], x, y, src);
        const float al = src[3];
        float r[4];
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = src[c] * al + d[c] * (1.0f - al);
        r[3] = src[3] * 1.0f + d[3] * (1.0f - al);
#pragma unroll
        for (int c = 0; c < 4; ++c) d[c] = (float)(_Float16)r[c];
    }
    b.samples[ps] = pack_half4(d);
}

// Render-pass resolve of the blended samples (S == 4): box average, same expression as in k_resolve_opaque.
__global__ __launch_bounds__(256) void k_resolve_samples(const ushort4 *__restrict__ samples, ushort4 *__restrict__ hdr_out,
                                                         size_t first_pixel, size_t n_pixels) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pixels) return;
    const size_t pix = first_pixel + i;
    float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float col[4][4];
#pragma unroll
    for (int sm = 0; sm < 4; ++sm) {
        const ushort4 h = samples[pix * 4u + (size_t)sm];
        col[sm][0] = (float)__builtin_bit_cast(_Float16, h.x); col[sm][1] = (float)__builtin_bit_cast(_Float16, h.y);
        col[sm][2] = (float)__builtin_bit_cast(_Float16, h.z); col[sm][3] = (float)__builtin_bit_cast(_Float16, h.w);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = ((col[0][c] + col[1][c]) + (col[2][c] + col[3][c])) * 0.25f;
    hdr_out[pix] = pack_half4(out);
}

// ------------------------------------------------------------------------------------------------ K7 tonemap
// 2 pixels per thread: one 16-byte load, one 8-byte store.
__global__ __launch_bounds__(256) void k_tonemap(const ushort4 *__restrict__ hdr, uchar4 *__restrict__ out,
                                                 float4 *__restrict__ out_f32, size_t first_pixel, size_t n_pixels,
                                                 const unsigned char *__restrict__ srgb_lut, uint32_t output_format) {
    const bool bgr = (output_format & 1u) != 0u, manual = (output_format & 2u) != 0u;
    const size_t pair = (size_t)blockIdx.x * 256u + threadIdx.x;
    const size_t i0 = first_pixel + pair * 2u;
    if (pair * 2u >= n_pixels) return;
    const bool two = pair * 2u + 1u < n_pixels;
    ushort4 h[2];
    if (two && (i0 & 1u) == 0u) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(hdr + i0);
        h[0] = make_ushort4(raw.x & 0xFFFFu, raw.x >> 16, raw.y & 0xFFFFu, raw.y >> 16);
        h[1] = make_ushort4(raw.z & 0xFFFFu, raw.z >> 16, raw.w & 0xFFFFu, raw.w >> 16);
    } else {
        h[0] = hdr[i0];
        h[1] = two ? hdr[i0 + 1u] : make_ushort4(0, 0, 0, 0);
    }
    uchar4 o8[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        o8[k] = tonemap_half4(srgb_lut, h[k], bgr);
        if (out_f32 != nullptr && (k == 0 || two)) {  // float view of the same target (readback tap only)
            const float r = (float)__builtin_bit_cast(_Float16, h[k].x), g = (float)__builtin_bit_cast(_Float16, h[k].y);
            const float b = (float)__builtin_bit_cast(_Float16, h[k].z), al = (float)__builtin_bit_cast(_Float16, h[k].w);
            out_f32[i0 + (size_t)k] = make_float4(manual ? srgb_scene_to_display(r) : srgb_oetf(r), manual ? srgb_scene_to_display(g) : srgb_oetf(g),
                                                  manual ? srgb_scene_to_display(b) : srgb_oetf(b),
                                                  (!(al > 0.0f)) ? 0.0f : (al >= 1.0f ? 1.0f : al));
        }
    }
    out[i0] = o8[0];
    if (two) out[i0 + 1u] = o8[1];
}
