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
    // ... as nodes of one linked list per pixel sample (no sort, no count on the host):
    unsigned long long *frag_keys;         // node: next node of the sample's list (R3N_INVALID ends it) << 32 | draw order of the triangle
    uint32_t *frag_vals;                   // node: canonical slot + 1
    uint32_t *frag_count;                  // nodes allocated
    uint32_t frag_capacity;
    uint32_t *frag_head;                   // per pixel sample: first node of its list, R3N_INVALID = none
    uint32_t *status;                      // host-visible: bit 0 set when nodes / work items ran out (reported by a later call)
    uint32_t row_begin, row_end;           // rows of the viewport this launch may touch: everything ([0, 0xFFFFFFFF)) unless the rank is sharded by rows
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
    // ... then, in the WORK-ITEM kernel (HOIST): what that sample needs of the material and of the texture array, fetched once
    // per item through scalar loads instead of per fragment behind the material record: texture id, sampler choice, coordinate
    // transform, descriptor.  A fragment's chain is then level offset -> texels instead of material -> descriptor -> level offset
    // -> texels.  (The per-triangle pass keeps the per-fragment loads: its triangles are per THREAD, and twenty more vector
    // registers there cost an occupancy step.)
    uint32_t tex0;
    bool nearest;
    float uvt[12];                      // material.uv_transform0 (forward only)
    r3n_texture_desc32 tdesc;           // descriptor of tex0 (zero when the id is outside the array)
    bool cutout;
    float thr[3];    // device_math.h::edge_threshold of the three edges
    int x0, y0, x1, y1;
};


// TEX: the launch may meet cutout materials whose alpha comes from the albedo texture (row N2).  The lean variant
// (no texture code, fewer registers) is launched whenever the world has no textures or the key is not cutout.
template <bool DEPTH_ONLY, bool TEX, bool NOCUT = false>
R3N_DEV bool prepare_triangle(const RasterArgs &a, uint32_t obj, uint32_t tri, bool positive_visible, TriWork &tw) {
    const r3n_object128 &ob = a.objects[obj];
    // The record's fields this function needs in TWO loads issued together -- bytes 80..95 (first_index, index_count, material_index,
    // the position attribute's offset) and `enabled` -- and waited for once: read field by field behind the `enabled` test they were
    // three dependent round trips of the per-triangle chain (list entry -> record -> indices -> positions).
    const uint4 of = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(&ob) + offsetof(r3n_object128, first_index));
    const uint32_t enabled = ob.enabled;
    asm volatile("" : : "v"(of.x), "v"(of.w), "v"(enabled));  // both loads in flight before the test below can split them
    if (enabled == 0u) return false;  // opaque.wgsl:104-112 / depth.wgsl:64-72
    const uint32_t first = of.x + tri * 3u;
    const uint32_t pos_off = of.w;
    uint32_t idx[3];
    float p[3][4];
    fetch_indices3(a.mesh, first, idx);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float v[3];
        fetch_vec3(a.mesh, pos_off, idx[k], v);
        mul_point(a.baked[obj].model_view_proj, v, p[k]);
    }
    const float half_w = (float)a.vp_w / 2.0f, half_h = (float)a.vp_h / 2.0f;
    setup_triangle(p, half_w, half_h, positive_visible, tw.ts);
    if (!tw.ts.valid) return false;
    if (!tri_bounds(p, half_w, half_h, (int)a.vp_w, (int)a.vp_h, tw.x0, tw.y0, tw.x1, tw.y1)) return false;
    // sort-first sharding (R3N_SHARD_ROWS): this rank scans rows [row_begin, row_end) only.  The box only limits the scan
    // (coverage and depth are decided per pixel), so the rows inside come out exactly as in the unsharded frame.
    tw.y0 = max(tw.y0, (int)a.row_begin);
    tw.y1 = min(tw.y1, (int)min(a.row_end, a.vp_h) - 1);
    if (tw.y0 > tw.y1) return false;
#pragma unroll
    for (int i = 0; i < 3; ++i) tw.thr[i] = edge_threshold(tw.ts.e[i][0], tw.ts.e[i][1]);
    // NOCUT: the launch draws the opaque key (r3n_forward picks the instantiation): what the cutout test needs -- vertex alphas,
    // material alpha / threshold, texture coordinates -- is then not even allocated: 102 -> 80 vector registers = six waves per
    // SIMD instead of four.  (Stand-alone the kernel runs the same, 43.7 vs 44.0 us per shadow launch; the FRAME gains 2.2 %,
    // 0.988 -> 0.966 ms: kernels of other streams find room beside it.)
    tw.cutout = !NOCUT && a.key == R3N_KEY_CUTOUT;
    tw.material = of.z < a.n_materials ? of.z : 0u;
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

// (constants, not build options: the measurements that fixed them are in profiles/r0N_summary.md; an experiment is a patch)
#define R3N_PREREAD_SMALL 0
#define R3N_PREREAD_MS 1        // work-item kernel on a multisampled viewport: read the pixel's keys before the atomics
#define R3N_PREREAD_VIEWPORT 0  // the same at one sample per pixel: the kernel alone gains (168 -> 152 us) but the frame with
                                // frames in flight loses (1.18 -> 1.20 ms): off
#define R3N_PREREAD_BIG 0
// PREREAD: plain load + compare before the atomic.  It filters occluded fragments cheaply (the load may be
// stale, which is only conservative because keys grow monotonically) but puts a dependent load in front of every
// atomic; without it the atomic is fire-and-forget.
// Alpha of the albedo texture at pixel (x, y) for the cutout test.  Forward (opaque.wgsl:207-215): coordinates through
// uv_transform0, sampler chosen by FLAGS_NEAREST.  Depth-only (depth.wgsl:108-118, quirks reproduced): raw coords0,
// always the primary sampler, and uvdy = dpdx(coords).
// SHORTA: every albedo map a cutout material of the world binds is on the sampler's short path (the host's census, r3n.hip
// refresh_material_classes): the general sampler is not instantiated -- it is most of these kernels' vector registers.
template <bool DEPTH_ONLY, bool TEX, bool HOIST = false, bool SHORTA = false>
R3N_DEV float cutout_texture_alpha(const RasterArgs &a, const TriWork &tw, int x, int y) {
    if (!TEX || !tw.alpha_tex) return 1.0f;
    float coords[2], ddx[2], ddy[2];
    if (HOIST) {
        if (DEPTH_ONLY) {
            frag_coords(tw.ts, tw.uv, nullptr, x, y, coords, ddx, ddy);
            return tex_sample_alpha<MathExact, SHORTA>(a.tex, tw.tex0, tw.tdesc, false, coords[0], coords[1], ddx, ddx);
        }
        frag_coords(tw.ts, tw.uv, tw.uvt, x, y, coords, ddx, ddy);
        return tex_sample_alpha<MathExact, SHORTA>(a.tex, tw.tex0, tw.tdesc, tw.nearest, coords[0], coords[1], ddx, ddy);
    }
    const r3n_material208 &m = a.materials[tw.material];
    const uint32_t id = m.textures[0];
    r3n_texture_desc32 d{};
    if (id - 1u < a.tex.count) d = a.tex.descs[id - 1u];
    if (DEPTH_ONLY) {
        frag_coords(tw.ts, tw.uv, nullptr, x, y, coords, ddx, ddy);
        return tex_sample_alpha<MathExact, SHORTA>(a.tex, id, d, false, coords[0], coords[1], ddx, ddx);
    }
    frag_coords(tw.ts, tw.uv, m.uv_transform0, x, y, coords, ddx, ddy);
    return tex_sample_alpha<MathExact, SHORTA>(a.tex, id, d, (m.flags & R3N_FLAGS_NEAREST) != 0u, coords[0], coords[1], ddx, ddy);
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
R3N_DEV uint32_t global_add_u32(uint32_t *p, uint32_t v) {  // returning; global address space whatever the pointer's provenance
    typedef __attribute__((address_space(1))) uint32_t *gp_t;
    return __hip_atomic_fetch_add((gp_t)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
R3N_DEV void global_max_u64(unsigned long long *p, unsigned long long v) {
    typedef __attribute__((address_space(1))) unsigned long long *gp_t;
    (void)__hip_atomic_fetch_max((gp_t)(unsigned long long)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The same with the target named as (wave-uniform base, 32-bit BYTE offset): the atomic then takes the base from scalar registers
// and the offset from one vector register (global_atomic_* v_off, v_data, s[base]) -- no 64-bit address arithmetic per pixel
// (v_mad_u64_u32 + two v_lshl_add_u64 per fragment before).  Targets stay below 4 GiB (r3n_frame_begin checks).
R3N_DEV void global_max_u32_at(uint32_t *base, uint32_t byte_off, uint32_t v) {
    typedef __attribute__((address_space(1))) uint32_t *gp_t;
    typedef __attribute__((address_space(1))) char *gc_t;
    (void)__hip_atomic_fetch_max((gp_t)((gc_t)(unsigned long long)base + byte_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
R3N_DEV void global_max_u64_at(unsigned long long *base, uint32_t byte_off, unsigned long long v) {
    typedef __attribute__((address_space(1))) unsigned long long *gp_t;
    typedef __attribute__((address_space(1))) char *gc_t;
    (void)__hip_atomic_fetch_max((gp_t)((gc_t)(unsigned long long)base + byte_off), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Element index of pixel (x, y) of the viewport inside the target.  Rows and pitch are below 2^16 (r3n_frame_begin), the target
// below 2^29 samples: 24-bit multiply, 32-bit offsets.
// (a 4 x 4-texel tiled atlas was timed in round 4 -- shadow work items -22 % -- and not built: the PCF's footprints would straddle
// tiles in the kernel the frame waits for; profiles/r04_summary.md section 6c)
template <bool DEPTH_ONLY>
R3N_DEV uint32_t target_pixel(const RasterArgs &a, uint32_t x, uint32_t y) {
    return __umul24(a.vp_y + y, a.target_pitch) + (a.vp_x + x);
}

template <bool DEPTH_ONLY, bool PREREAD, int S = 1, bool TEX = false, bool BLEND = false, bool HOIST = false, bool SHORTA = false>
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
            if (!edge_eval_thr(tw.ts, tw.thr, (float)x + sx, (float)y + sy, E)) continue;
            const float z = frag_depth(tw.ts, (float)x + sx, (float)y + sy);
            if (!(z >= 0.0f && z <= 1.0f)) continue;
            const size_t ps = pix * (size_t)S + (size_t)sm;
            const float dz = __uint_as_float((uint32_t)(a.vis[ps] >> 32));
            if (!(z >= dz)) continue;
            const uint32_t at = atomicAdd(a.frag_count, 1u);
            if (at < a.frag_capacity) {
                const uint32_t next = atomicExch(&a.frag_head[ps], at);  // push: the list's order is irrelevant (k_blend_apply orders by key)
                a.frag_keys[at] = ((unsigned long long)next << 32) | (unsigned long long)tw.material;
                a.frag_vals[at] = tw.slot1;
            } else {
                __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }
    if (S == 1) {
        float E[3];
        if (!edge_eval_thr(tw.ts, tw.thr, (float)x + 0.5f, (float)y + 0.5f, E)) return;
        float z = frag_depth(tw.ts, (float)x + 0.5f, (float)y + 0.5f);
        if (!(z >= 0.0f && z <= 1.0f)) return;  // depth clip (unclipped_depth: false, forward.rs:343)
        if (z == 0.0f) z = 0.0f;                // canonicalise -0
        // rows and pitch are below 2^16 (r3n_frame_begin), the target below 2^29 samples: 24-bit multiply, 32-bit byte offsets
        const uint32_t pix = target_pixel<DEPTH_ONLY>(a, (uint32_t)x, (uint32_t)y);
        const uint32_t zb = __float_as_uint(z);
        if (tw.cutout) {
            // The cutout test is the expensive part of this fragment (two more attribute interpolations for the derivatives and
            // a trilinear sample of the albedo map: the Bistro-like scene's foliage cards, profiles/r05_summary.md) and decides
            // nothing for a fragment the target already beats: the target only grows (MAX), so a key that loses against the value
            // read here loses against the final one -- skip the test and the atomic.  The opaque key of the same pass was drawn by
            // the launches in front of this one, so foliage behind walls ends here.
            if (DEPTH_ONLY) { if (!(zb > a.depth[pix])) return; }
            else if (!((((unsigned long long)zb << 32) | (unsigned long long)tw.slot1) > a.vis[pix])) return;
            // (the interpolated vertex alpha enters the test only under ALBEDO_ACTIVE + ALBEDO_BLEND, cutout_alpha: skipped otherwise --
            // a reciprocal and eight more operations per fragment of a kernel that is bound by vector issue on a foliage scene)
            float al = 1.0f;
            if ((tw.mat_flags & (R3N_FLAGS_ALBEDO_ACTIVE | R3N_FLAGS_ALBEDO_BLEND)) == (R3N_FLAGS_ALBEDO_ACTIVE | R3N_FLAGS_ALBEDO_BLEND)) {
                const float rs = 1.0f / ((E[0] + E[1]) + E[2]);
                al = ((E[0] * rs) * tw.va[0] + (E[1] * rs) * tw.va[1]) + (E[2] * rs) * tw.va[2];
            }
            if (cutout_alpha(tw.mat_flags, tw.mat_alpha, cutout_texture_alpha<DEPTH_ONLY, TEX, HOIST, SHORTA>(a, tw, x, y), al) < tw.mat_cutoff) return;  // opaque.wgsl:231-235 / depth.wgsl:123-125
        }
        if (DEPTH_ONLY) {
            if (!PREREAD || zb > a.depth[pix]) global_max_u32_at(a.depth, pix << 2, zb);
        } else {
            const unsigned long long key = ((unsigned long long)zb << 32) | (unsigned long long)tw.slot1;
            if (!PREREAD || key > a.vis[pix]) global_max_u64_at(a.vis, pix << 3, key);
        }
    } else {
        uint32_t mask = 0u;
        float zs[S];
#pragma unroll
        for (int sm = 0; sm < S; ++sm) {
            float E[3];
            zs[sm] = 0.0f;
            if (!edge_eval_thr(tw.ts, tw.thr, (float)x + k_sample_pos4[sm][0], (float)y + k_sample_pos4[sm][1], E)) continue;
            float z = frag_depth(tw.ts, (float)x + k_sample_pos4[sm][0], (float)y + k_sample_pos4[sm][1]);
            if (!(z >= 0.0f && z <= 1.0f)) continue;
            if (z == 0.0f) z = 0.0f;
            zs[sm] = z;
            mask |= 1u << sm;
        }
        if (!mask) return;
        const size_t pix = ((size_t)(a.vp_y + (uint32_t)y) * a.target_pitch + a.vp_x + (uint32_t)x) * (size_t)S;
        // PREREAD (the work-item kernel under MSAA): the pixel's four keys are 32 contiguous bytes; read them once and
        // skip the atomics that cannot win.  Memory-side atomics are 65 % of that kernel at 4 samples (ablation,
        // bench scene: 1.43 ms -> 0.50 ms without them); the read removes the overdrawn ones: 1.43 -> 1.04 ms.  Keys only
        // grow, so a key that loses against a stale read loses against the current value too.
        // The cutout key reads them whatever PREREAD says, IN FRONT of its alpha test (see the single-sample branch): a pixel
        // none of whose covered samples can still win skips the texture sample.
        const bool preread = PREREAD || tw.cutout;
        unsigned long long cur[S];
        if (preread) {
#pragma unroll
            for (int sm = 0; sm < S; ++sm) cur[sm] = a.vis[pix + (size_t)sm];
#pragma unroll
            for (int sm = 0; sm < S; ++sm)
                if ((mask & (1u << sm)) && !((((unsigned long long)__float_as_uint(zs[sm]) << 32) | (unsigned long long)tw.slot1) > cur[sm])) mask &= ~(1u << sm);
            if (!mask) return;
        }
        if (tw.cutout) {
            float al = 1.0f;
            if ((tw.mat_flags & (R3N_FLAGS_ALBEDO_ACTIVE | R3N_FLAGS_ALBEDO_BLEND)) == (R3N_FLAGS_ALBEDO_ACTIVE | R3N_FLAGS_ALBEDO_BLEND)) {
                float E[3];
                (void)edge_eval_thr(tw.ts, tw.thr, (float)x + 0.5f, (float)y + 0.5f, E);
                const float rs = 1.0f / ((E[0] + E[1]) + E[2]);
                al = ((E[0] * rs) * tw.va[0] + (E[1] * rs) * tw.va[1]) + (E[2] * rs) * tw.va[2];
            }
            if (cutout_alpha(tw.mat_flags, tw.mat_alpha, cutout_texture_alpha<DEPTH_ONLY, TEX, HOIST, SHORTA>(a, tw, x, y), al) < tw.mat_cutoff) return;
        }
#pragma unroll
        for (int sm = 0; sm < S; ++sm)
            if (mask & (1u << sm))
                global_max_u64(&a.vis[pix + (size_t)sm], ((unsigned long long)__float_as_uint(zs[sm]) << 32) | (unsigned long long)tw.slot1);
    }
}

// One scan step of the work-item kernel at one sample per pixel, opaque key: the same coverage / depth-clip / target arithmetic as
// shade_pixel, as STRAIGHT-LINE code -- every lane evaluates everything, the tests are combined into one predicate and only the
// atomic is predicated.  Why: the kernel is bound by instruction ISSUE, and more by the scalar unit than by the vector units
// (profiles/r04_summary.md: 31 M scalar against 27 M vector instructions per shadow launch, 78 % of a SIMD's scalar issue slots):
// every early-out `if` costs a mask AND, an exec update and a branch on the scalar unit and saves vector work only when ALL 64
// lanes fail, which the block rejection test in front of the step has already made rare.
// The predicate is a chain of v_cmpx (each narrows EXEC on the vector unit: no mask ANDs, no exec update and no branch on the
// scalar unit), the atomic under the narrowed mask, EXEC restored -- one asm statement.  (Stand-alone the early-out form of rounds
// 1-3, a one-predicate C++ form and this one run the same, 85.3 / 85.3 / 85.1 us per shadow launch: the kernel waits for its
// atomics; in the frame, beside the resolve, fewer scalar instructions are worth 1.2 %, 1.0504 -> 1.0378 ms.)
// fine: the step's lanes hold block index `b` (0xFFFFFFFF: none); coarse: b is not tested.
template <bool DEPTH_ONLY, bool FINE>
R3N_DEV void shade_pixel_cmpx(const RasterArgs &a, const TriWork &tw, int x, int y, uint32_t b, int rx0, int rx1, int ry1) {
    float E[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) E[i] = (tw.ts.e[i][0] * ((float)x + 0.5f) + tw.ts.e[i][1] * ((float)y + 0.5f)) + tw.ts.e[i][2];
    const float z = frag_depth(tw.ts, (float)x + 0.5f, (float)y + 0.5f);
    const uint32_t zb = __float_as_uint(z) & 0x7FFFFFFFu;  // an accepted z is in [0, 1] or -0: clearing the sign bit canonicalises -0 and changes nothing else
    const uint32_t pix = target_pixel<DEPTH_ONLY>(a, (uint32_t)x, (uint32_t)y);
    unsigned long long save;
    typedef __attribute__((address_space(1))) void *gv_t;
#define R3N_CMPX_CHAIN                                                                                             \
    "s_mov_b64 %[save], exec\n\t"                                                                                   \
    "v_cmpx_le_i32 vcc, %[rx0], %[x]\n\tv_cmpx_ge_i32 vcc, %[rx1], %[x]\n\tv_cmpx_ge_i32 vcc, %[ry1], %[y]\n\t"                                        \
    "v_cmpx_le_f32 vcc, %[t0], %[e0]\n\tv_cmpx_le_f32 vcc, %[t1], %[e1]\n\tv_cmpx_le_f32 vcc, %[t2], %[e2]\n\t"     \
    "v_cmpx_le_f32 vcc, 0, %[z]\n\tv_cmpx_ge_f32 vcc, 1.0, %[z]\n\t"
    if (DEPTH_ONLY) {
        if (FINE)
            asm volatile("s_mov_b64 %[save], exec\n\tv_cmpx_gt_u32 vcc, 64, %[b]\n\t"
                         "v_cmpx_le_i32 vcc, %[rx0], %[x]\n\tv_cmpx_ge_i32 vcc, %[rx1], %[x]\n\tv_cmpx_ge_i32 vcc, %[ry1], %[y]\n\t"
                         "v_cmpx_le_f32 vcc, %[t0], %[e0]\n\tv_cmpx_le_f32 vcc, %[t1], %[e1]\n\tv_cmpx_le_f32 vcc, %[t2], %[e2]\n\t"
                         "v_cmpx_le_f32 vcc, 0, %[z]\n\tv_cmpx_ge_f32 vcc, 1.0, %[z]\n\t"
                         "global_atomic_umax %[off], %[data], %[base]\n\t"
                         "s_mov_b64 exec, %[save]"
                         : [save] "=&s"(save)
                         : [b] "v"(b), [rx0] "s"(rx0), [rx1] "s"(rx1), [x] "v"(x), [ry1] "s"(ry1), [y] "v"(y), [t0] "s"(tw.thr[0]), [e0] "v"(E[0]),
                           [t1] "s"(tw.thr[1]), [e1] "v"(E[1]), [t2] "s"(tw.thr[2]), [e2] "v"(E[2]), [z] "v"(z), [off] "v"(pix << 2),
                           [data] "v"(zb), [base] "s"((gv_t)(unsigned long long)a.depth)
                         : "vcc", "memory");
        else
            asm volatile(R3N_CMPX_CHAIN
                         "global_atomic_umax %[off], %[data], %[base]\n\t"
                         "s_mov_b64 exec, %[save]"
                         : [save] "=&s"(save)
                         : [rx0] "s"(rx0), [rx1] "s"(rx1), [x] "v"(x), [ry1] "s"(ry1), [y] "v"(y), [t0] "s"(tw.thr[0]), [e0] "v"(E[0]),
                           [t1] "s"(tw.thr[1]), [e1] "v"(E[1]), [t2] "s"(tw.thr[2]), [e2] "v"(E[2]), [z] "v"(z), [off] "v"(pix << 2),
                           [data] "v"(zb), [base] "s"((gv_t)(unsigned long long)a.depth)
                         : "vcc", "memory");
    } else {
        const unsigned long long key = ((unsigned long long)zb << 32) | (unsigned long long)tw.slot1;
        if (FINE)
            asm volatile("s_mov_b64 %[save], exec\n\tv_cmpx_gt_u32 vcc, 64, %[b]\n\t"
                         "v_cmpx_le_i32 vcc, %[rx0], %[x]\n\tv_cmpx_ge_i32 vcc, %[rx1], %[x]\n\tv_cmpx_ge_i32 vcc, %[ry1], %[y]\n\t"
                         "v_cmpx_le_f32 vcc, %[t0], %[e0]\n\tv_cmpx_le_f32 vcc, %[t1], %[e1]\n\tv_cmpx_le_f32 vcc, %[t2], %[e2]\n\t"
                         "v_cmpx_le_f32 vcc, 0, %[z]\n\tv_cmpx_ge_f32 vcc, 1.0, %[z]\n\t"
                         "global_atomic_umax_x2 %[off], %[data], %[base]\n\t"
                         "s_mov_b64 exec, %[save]"
                         : [save] "=&s"(save)
                         : [b] "v"(b), [rx0] "s"(rx0), [rx1] "s"(rx1), [x] "v"(x), [ry1] "s"(ry1), [y] "v"(y), [t0] "s"(tw.thr[0]), [e0] "v"(E[0]),
                           [t1] "s"(tw.thr[1]), [e1] "v"(E[1]), [t2] "s"(tw.thr[2]), [e2] "v"(E[2]), [z] "v"(z), [off] "v"(pix << 3),
                           [data] "v"(key), [base] "s"((gv_t)(unsigned long long)a.vis)
                         : "vcc", "memory");
        else
            asm volatile(R3N_CMPX_CHAIN
                         "global_atomic_umax_x2 %[off], %[data], %[base]\n\t"
                         "s_mov_b64 exec, %[save]"
                         : [save] "=&s"(save)
                         : [rx0] "s"(rx0), [rx1] "s"(rx1), [x] "v"(x), [ry1] "s"(ry1), [y] "v"(y), [t0] "s"(tw.thr[0]), [e0] "v"(E[0]),
                           [t1] "s"(tw.thr[1]), [e1] "v"(E[1]), [t2] "s"(tw.thr[2]), [e2] "v"(E[2]), [z] "v"(z), [off] "v"(pix << 3),
                           [data] "v"(key), [base] "s"((gv_t)(unsigned long long)a.vis)
                         : "vcc", "memory");
    }
#undef R3N_CMPX_CHAIN
}

// The three edge thresholds of a work item travel in the top bits of its `material` word (bit 29 + i set: edge i is NOT a
// top / left edge, threshold = the smallest subnormal): the producer has them in registers, the consumer would spend two dozen
// scalar instructions per item deriving them from the coefficients' bits.
#define R3N_BIG_THR_SHIFT 29u
#define R3N_BIG_MATERIAL_MASK 0x1FFFFFFFu
R3N_DEV uint32_t pack_thresholds(const float thr[3]) {
    return ((thr[0] != 0.0f ? 1u : 0u) | (thr[1] != 0.0f ? 2u : 0u) | (thr[2] != 0.0f ? 4u : 0u)) << R3N_BIG_THR_SHIFT;
}

// The rasterisers' constants.  They are NOT build options: each was fixed by a measurement (quoted beside it, details in
// profiles/r0N_summary.md); an experiment with another value is a patch against this file (profiles/patches/).
#define R3N_SMALL_MAX 8      // triangles up to 8 x 8 px are scanned in place by their thread
// Work items cover at most R3N_TILE x R3N_TILE px.  Measured on the bench scene (us per frame, shadow big / viewport
// big): tile 64 coarse-only 478 / 218, tile 32 coarse-only 475 / 204, tile 32 + fine 423 / 185, tile 16 + fine 482 / 191.
#define R3N_TILE 32
#define R3N_SMALL_OCC 1      // min waves per SIMD asked of k_raster_small (launch bound)
#define R3N_ITEM_ALIGN 16    // (a power of two <= R3N_TILE; measured: viewport work items 84.7 -> 73.2 us, shadow 85.0 -> 80.8 us per launch)
#define R3N_FINE_LW_DEPTH 3  // log2 of the fine block's width on a depth target (2: 4x4, 3: 8x2, 4: 16x1 texels).  Measured (shadow work items
                             // per launch stand-alone / frame): 4x4 81.5 us / 1.025 ms, 8x2 74.0 / 1.003, 16x1 76.4 / 1.033 (fewest lines per
                             // step, but more steps: partly covered blocks along every edge)
#define R3N_FINE_LW_VIS 3    // the same on the key target (a line is 8 keys): 4x4 72.6 us / 1.025 ms, 8x2 66.4 / 1.001; both 8x2: 1.003
// Measured and NOT kept (round 4, profiles/r04_summary.md sections 2 and 6c; the code is in the history, commit 79bedf3): the
// per-triangle pass with the object record + baked matrix through scalar loads in a waterfall over the wave's distinct objects
// (viewport 22.9 -> 37.6 us, shadow 43.8 -> 45.9 us per launch); an XCD-affine walk of the work queue (shadow 81.7 -> 87.1 us);
// the record after the next one pulled into the local L2 by an unwaited vector load (81.6 -> 81.3 us, frame 1.017 -> 1.024 ms).

// Stage 1: one thread per list entry.  Small triangles are scanned in place; larger ones are split into
// <=64x64 px items for stage 2.
template <bool DEPTH_ONLY, int S = 1, bool TEX = false, bool NOCUT = false, bool SHORTA = false>
R3N_DEV void raster_small_body(const RasterArgs &a) {
    // block b walks sub-list (b % R3N_SUBQ) of the region, and appends to work sub-queue (b % R3N_BIGQ)
    const uint32_t q = blockIdx.x % R3N_SUBQ;
    const uint32_t n = a.sub_counts[a.key * R3N_SUBQ + q];
    const r3n_tri_ref *list = a.list + (size_t)(a.key * R3N_SUBQ + q) * a.subcap;
    // producer side of the work queue: the sub-queue is chosen per WAVE.  readfirstlane makes the counter address
    // provably wave-uniform, which is what lets the compiler fold a wave's appends into ONE atomic (with a
    // per-lane address every lane issues its own returning atomic -- measured 4x slower kernels)
    const uint32_t bq = __builtin_amdgcn_readfirstlane((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % R3N_BIGQ);
    const bool positive_visible = (a.hdr->flags & R3N_PCU_POSITIVE_AREA_VISIBLE) != 0u;
    const uint32_t stride = (gridDim.x / R3N_SUBQ) * blockDim.x;
    for (uint32_t i = (blockIdx.x / R3N_SUBQ) * blockDim.x + threadIdx.x; i < n; i += stride) {
        const r3n_tri_ref ref = list[i];
        TriWork tw;
        if (!prepare_triangle<DEPTH_ONLY, TEX, NOCUT>(a, ref.object, ref.triangle, positive_visible, tw)) continue;
        const int bw = tw.x1 - tw.x0 + 1, bh = tw.y1 - tw.y0 + 1;
        if (bw <= R3N_SMALL_MAX && bh <= R3N_SMALL_MAX) {
            for (int y = tw.y0; y <= tw.y1; ++y)
                for (int x = tw.x0; x <= tw.x1; ++x) shade_pixel<DEPTH_ONLY, R3N_PREREAD_SMALL != 0, S, TEX, false, false, SHORTA>(a, tw, x, y);
        } else {
            // Work items start on a multiple of R3N_ITEM_ALIGN pixels in x: the scan's blocks then sit on the target's
            // 64-byte lines (16 depth texels, 8 keys) instead of straddling them, and one atomic instruction touches fewer lines --
            // the work-item kernel is bound by the number of line-sized atomic requests a CU can have in flight to the memory
            // side (profiles/r04_summary.md).  The box only limits the scan: results are unchanged.
            const int ax0 = tw.x0 & ~(R3N_ITEM_ALIGN - 1);
            const uint32_t tx = (uint32_t)(tw.x1 - ax0 + R3N_TILE) / R3N_TILE, ty = (uint32_t)(bh + (R3N_TILE - 1)) / R3N_TILE;
            const uint32_t cnt = tx * ty;
            const uint32_t start = global_add_u32(&a.big_count[bq], cnt);
            r3n_big_item *big = a.big_items + (size_t)bq * a.big_capacity;
            for (uint32_t t = 0; t < cnt; ++t) {
                const uint32_t ix = t % tx, iy = t / tx;
                // the item's box stays inside the triangle's (the scan never leaves the box the oracle scans: the boxes of
                // triangles that cross the depth-clip planes are not supersets of their coverage); the consumer lays its block
                // grid out from rx0 rounded down to R3N_ITEM_ALIGN, which is this column's aligned start
                const int rx0 = max(ax0 + (int)ix * R3N_TILE, tw.x0), ry0 = tw.y0 + (int)iy * R3N_TILE;
                const int rx1 = min(ax0 + (int)ix * R3N_TILE + (R3N_TILE - 1), tw.x1), ry1 = min(ry0 + (R3N_TILE - 1), tw.y1);
                if (start + t < a.big_capacity) {
                    r3n_big_item it;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) it.e[i][c] = tw.ts.e[i][c];
                        it.z[i] = tw.ts.z[i];
                    }
                    it.slot1 = DEPTH_ONLY ? 0u : tw.slot1;
                    it.material = tw.material | pack_thresholds(tw.thr);  // (material indices stay below 2^29: r3n_materials_write)
                    it.xy0 = (uint32_t)rx0 | ((uint32_t)ry0 << 16);
                    it.xy1 = (uint32_t)rx1 | ((uint32_t)ry1 << 16);
                    big[start + t] = it;
                    if (tw.cutout) {  // (launch-uniform) vertex alpha, and the uvs of a textured alpha
                        r3n_big_uv bu;
#pragma unroll
                        for (int k = 0; k < 3; ++k) { bu.uv[k][0] = tw.uv[k][0]; bu.uv[k][1] = tw.uv[k][1]; bu.va[k] = tw.va[k]; }
                        bu._pad[0] = bu._pad[1] = bu._pad[2] = 0u;
                        a.big_uv[(size_t)bq * a.big_capacity + start + t] = bu;
                    }
                } else {
                    // queue full: never drop work -- scan the region here (slow path)
                    for (int y = ry0; y <= ry1; ++y)
                        for (int x = rx0; x <= rx1; ++x) shade_pixel<DEPTH_ONLY, R3N_PREREAD_SMALL != 0, S, TEX, false, false, SHORTA>(a, tw, x, y);
                }
            }
        }
    }
}

template <bool DEPTH_ONLY, int S = 1, bool TEX = false, bool NOCUT = false, bool SHORTA = false>
__global__ __launch_bounds__(256, R3N_SMALL_OCC) void k_raster_small(RasterArgs a) {
    raster_small_body<DEPTH_ONLY, S, TEX, NOCUT, SHORTA>(a);
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
        if (start + t >= a.big_capacity) {  // reported through the status word (a later call fails with R3N_ERR_CAPACITY)
            __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            continue;
        }
        r3n_big_item it;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) it.e[i][c] = tw.ts.e[i][c];
            it.z[i] = tw.ts.z[i];
        }
        it.slot1 = tw.slot1;
        it.material = g | pack_thresholds(tw.thr);  // draw order (< 2^29: r3n_blend_order_write)
        it.xy0 = (uint32_t)rx0 | ((uint32_t)ry0 << 16);
        it.xy1 = (uint32_t)rx1 | ((uint32_t)ry1 << 16);
        big[start + t] = it;
    }
}

// Upper bound of edge function i over the pixel centres of an SxS block whose first pixel is (bx,by).  Each
// f32 operation is monotone, so evaluating the same expression at the extreme corner gives the exact maximum of
// the per-pixel values: a block with a negative maximum holds no covered pixel.
// MS: the evaluation points are the 4x sample positions, which span [0.125, 0.875] of a pixel in x and y.
template <int S = 8, bool MS = false, int SH = S>  // S x SH pixels
R3N_DEV bool block_may_cover(const TriSetup &ts, int bx, int by, int rx1, int ry1) {
    const float lo = MS ? 0.125f : 0.5f, hi = MS ? 0.875f : 0.5f;
    const float x_lo = (float)bx + lo, x_hi = (float)min(bx + (S - 1), rx1) + hi;
    const float y_lo = (float)by + lo, y_hi = (float)min(by + (SH - 1), ry1) + hi;
    bool may = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float A = ts.e[i][0], B = ts.e[i][1];
        // sign tests on the bits (scalar unit for the work items' uniform coefficients, device_math.h::f32_positive); with a
        // NaN coefficient v is NaN whichever corner is taken
        const float px = f32_positive(A) ? x_hi : x_lo, py = f32_positive(B) ? y_hi : y_lo;
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
template <bool DEPTH_ONLY, int S = 1, bool TEX = false, bool BLEND = false, bool NOCUT = false, bool SHORTA = false>
R3N_DEV void raster_big_body(RasterArgs a) {
    // Pin the kernel arguments the scan uses into SGPRs here: hipcc otherwise sinks the wait for their s_load
    // into the scan loop, and an `s_waitcnt lgkmcnt(0)` there would also wait for the record prefetch below.
    asm volatile("" : "+s"(a.target_pitch), "+s"(a.vp_x), "+s"(a.vp_y), "+s"(a.depth), "+s"(a.vis), "+s"(a.key),
                 "+s"(a.materials), "+s"(a.big_items), "+s"(a.big_count), "+s"(a.big_capacity));
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t qgroup = 0u;
    const uint32_t wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const int lx = (int)(lane & 7u), ly = (int)(lane >> 3);
    const uint32_t cap = a.big_capacity;
    // Sub-queue bounds live in registers: lane q < R3N_BIGQ holds [excl, incl), the flat indices of sub-queue q in the
    // concatenation (one vector load + a wave scan at kernel start, before any atomic is in flight).  Locating an item
    // is then a ballot + readlane, with no memory access in the item loop.
    const uint32_t qcnt_l = lane < R3N_BIGQ ? min(a.big_count[qgroup + lane], cap) : 0u;
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
        return reinterpret_cast<const uint32_t *>(a.big_items + (size_t)((qgroup + q) * cap + (flat - qb)));  // (queues hold < 2^32 items: r3n_create)
    };
    typedef __attribute__((address_space(4))) const uint32_t *sptr_t;
    // The record is read with SCALAR loads (constant address space + wave-uniform address => s_load into SGPRs).
    // A vector load would share the vmcnt counter with the scan's fire-and-forget atomics, and waiting for the
    // record would then wait for every outstanding atomic of the previous item (measured: 2.4 us per item).
    // Scalar-cache coherence is not an issue: the queue was written by the previous kernel on this stream.
    typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
    uint32_t flat = wave_global;
    const uint32_t *rec = locate(flat);
    u32x16 da = {}, na = {};
    if (rec) {
        sptr_t sp = (sptr_t)(unsigned long long)rec;
        da = *reinterpret_cast<__attribute__((address_space(4))) const u32x16 *>(sp);
    }
    while (rec) {
        // The next record's scalar loads stay in flight while this item is scanned (the records come from HBM /
        // Infinity Cache: without the overlap every item costs a full memory latency per wave).  hipcc sinks a
        // plain load to its first use, so the prefetch is an asm load it does not track; the matching wait
        // statement at the end of the iteration names both destinations (cdna_hip_programming.md section 5.7 (ii)).
        flat += nwaves;
        const uint32_t *nrec = locate(flat);
        if (nrec) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(na) : "s"(nrec) : "memory");
        uint32_t d[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) d[j] = da[j];
        auto bu = [&](int j) { return d[j]; };
        auto bf = [&](int j) { return __uint_as_float(d[j]); };
        TriWork w;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) w.ts.e[i][c] = bf(3 * i + c);
            w.ts.z[i] = bf(9 + i);
            w.va[i] = 1.0f;
        }
        w.ts.det = 1.0f;  // (the scan does not use it)
        w.ts.valid = true;
        // edge thresholds on the SCALAR unit: the coefficients are wave-uniform, the scalar unit compares integers only, and
        // behind the (empty) asm the compiler can no longer turn the bit tests back into vector float compares.  Same outcome
        // as edge_threshold for every non-NaN pair; with a NaN coefficient the edge value is NaN and fails any threshold.
        const uint32_t mt = bu(13);
#pragma unroll
        for (int i = 0; i < 3; ++i) w.thr[i] = __uint_as_float((mt >> (R3N_BIG_THR_SHIFT + (uint32_t)i)) & 1u);  // 0 or the smallest subnormal
        w.slot1 = bu(12);
        w.cutout = !BLEND && !NOCUT && a.key == R3N_KEY_CUTOUT;  // launch-uniform
        w.material = mt & R3N_BIG_MATERIAL_MASK;
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
            {  // rec - big_items = item index; the cutout record has the same index
                const size_t item = (size_t)(reinterpret_cast<const r3n_big_item *>(rec) - a.big_items);
                sptr_t up = (sptr_t)(unsigned long long)(a.big_uv + item);
#pragma unroll
                for (int k = 0; k < 3; ++k) w.va[k] = __uint_as_float(up[6 + k]);
                if (TEX && w.alpha_tex) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { w.uv[k][0] = __uint_as_float(up[2 * k]); w.uv[k][1] = __uint_as_float(up[2 * k + 1]); }
                    // the material's sampling state and the texture's descriptor: wave-uniform, through scalar loads
                    w.tex0 = mp[0];
                    w.nearest = (w.mat_flags & R3N_FLAGS_NEAREST) != 0u;
#pragma unroll
                    for (int k = 0; k < 12; ++k) w.uvt[k] = __uint_as_float(mp[offsetof(r3n_material208, uv_transform0) / 4 + k]);
                    w.tdesc = r3n_texture_desc32{};
                    if (w.tex0 - 1u < a.tex.count) {
                        sptr_t dp = (sptr_t)(unsigned long long)(a.tex.descs + (w.tex0 - 1u));
                        w.tdesc.offset = dp[0]; w.tdesc.width = dp[1]; w.tdesc.height = dp[2]; w.tdesc.mips = dp[3]; w.tdesc.format = dp[4];
                    }
                }
            }
        }
        const uint32_t kxy0 = bu(14), kxy1 = bu(15);
        const int rx0 = (int)(kxy0 & 0xFFFFu), ry0 = (int)(kxy0 >> 16);
        const int rx1 = (int)(kxy1 & 0xFFFFu), ry1 = (int)(kxy1 >> 16);
        const int gx0 = rx0 & ~(R3N_ITEM_ALIGN - 1);  // origin of the block grid: blocks sit on the target's cache lines
        if (rx1 - gx0 < 32 && ry1 - ry0 < 32) {
            // fine mode (regions up to 32x32 px): lane = 16-pixel block for the rejection test; every step then scans
            // FOUR surviving blocks, 16 lanes each -- small triangles fill the wave far better than with 8x8 blocks.
            // The block is FW x FH pixels (FW * FH = 16): the shape decides how many of the target's 64-byte lines one
            // atomic instruction touches (a line is 16 depth texels / 8 keys of ONE row), which is what bounds this kernel.
            constexpr int LW = DEPTH_ONLY ? R3N_FINE_LW_DEPTH : R3N_FINE_LW_VIS, FW = 1 << LW, FH = 16 >> LW, LC = 5 - LW;  // LC: log2 of the block columns
            static_assert(LW >= 2 && LW <= 4, "blocks of 4x4, 8x2 or 16x1 pixels");
            const int cbx = gx0 + (int)(lane & ((1u << LC) - 1u)) * FW, cby = ry0 + (int)(lane >> LC) * FH;
            const bool cand = cbx <= rx1 && cbx + (FW - 1) >= rx0 && cby <= ry1 && block_may_cover<FW, (S > 1), FH>(w.ts, cbx, cby, rx1, ry1);
            unsigned long long blocks = __ballot(cand);
            const uint32_t grp = lane >> 4;
            const int px = (int)(lane & (uint32_t)(FW - 1)), py = (int)((lane & 15u) >> LW);
            if (S == 1 && !BLEND && !w.cutout) {
                while (blocks) {
                    // four find-first / clear-bit pairs (an empty mask yields -1: no block, and clearing bit 63 of zero is harmless)
                    // (one statement: between separate asm statements the compiler pads every scalar write with a hazard s_nop)
                    uint32_t bsel[4];
                    asm("s_ff1_i32_b64 %0, %4\n\ts_bitset0_b64 %4, %0\n\t"
                        "s_ff1_i32_b64 %1, %4\n\ts_bitset0_b64 %4, %1\n\t"
                        "s_ff1_i32_b64 %2, %4\n\ts_bitset0_b64 %4, %2\n\t"
                        "s_ff1_i32_b64 %3, %4\n\ts_bitset0_b64 %4, %3"
                        : "=&s"(bsel[0]), "=&s"(bsel[1]), "=&s"(bsel[2]), "=&s"(bsel[3]), "+s"(blocks));
                    const uint32_t b = grp == 0u ? bsel[0] : (grp == 1u ? bsel[1] : (grp == 2u ? bsel[2] : bsel[3]));
                    int x, y;
                    asm("v_lshl_add_u32 %0, %1, %3, %2" : "=v"(x) : "v"(b & ((1u << LC) - 1u)), "v"(gx0 + px), "n"(LW));
                    asm("v_lshl_add_u32 %0, %1, %3, %2" : "=v"(y) : "v"(b >> LC), "v"(ry0 + py), "n"(4 - LW));
                    shade_pixel_cmpx<DEPTH_ONLY, true>(a, w, x, y, b, rx0, rx1, ry1);
                }
            } else
            while (blocks) {
                int bsel[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bsel[g] = blocks ? __builtin_ctzll(blocks) : 64;
                    blocks &= blocks - 1ull;
                }
                const int b = grp == 0u ? bsel[0] : (grp == 1u ? bsel[1] : (grp == 2u ? bsel[2] : bsel[3]));
                // (field << 2) + base as ONE shift-add each (the compiler's canonical (b << 2) & 28 form costs an instruction
                // more per coordinate and cannot be talked out of it)
                int x, y;
                asm("v_lshl_add_u32 %0, %1, %3, %2" : "=v"(x) : "v"(b & ((1 << LC) - 1)), "v"(gx0 + px), "n"(LW));
                asm("v_lshl_add_u32 %0, %1, %3, %2" : "=v"(y) : "v"(b >> LC), "v"(ry0 + py), "n"(4 - LW));
                if (b < 64 && x >= rx0 && x <= rx1 && y <= ry1) shade_pixel<DEPTH_ONLY, (R3N_PREREAD_BIG != 0) || (!DEPTH_ONLY && (S > 1 ? R3N_PREREAD_MS != 0 : R3N_PREREAD_VIEWPORT != 0)), S, TEX, BLEND, true, SHORTA>(a, w, x, y);
            }
        } else {
            const int cbx = gx0 + lx * 8, cby = ry0 + ly * 8;
            const bool cand = cbx <= rx1 && cbx + 7 >= rx0 && cby <= ry1 && block_may_cover<8, (S > 1)>(w.ts, cbx, cby, rx1, ry1);
            unsigned long long blocks = __ballot(cand);
            if (S == 1 && !BLEND && !w.cutout) {
                while (blocks) {
                    int b;
                    asm("s_ff1_i32_b64 %0, %1\n\ts_bitset0_b64 %1, %0" : "=&s"(b), "+s"(blocks));
                    const int x = gx0 + (b & 7) * 8 + lx, y = ry0 + (b >> 3) * 8 + ly;
                    shade_pixel_cmpx<DEPTH_ONLY, false>(a, w, x, y, 0u, rx0, rx1, ry1);
                }
            } else
            while (blocks) {
                const int b = __builtin_ctzll(blocks);
                blocks &= blocks - 1ull;
                const int x = gx0 + (b & 7) * 8 + lx, y = ry0 + (b >> 3) * 8 + ly;
                if (x >= rx0 && x <= rx1 && y <= ry1) shade_pixel<DEPTH_ONLY, (R3N_PREREAD_BIG != 0) || (!DEPTH_ONLY && (S > 1 ? R3N_PREREAD_MS != 0 : R3N_PREREAD_VIEWPORT != 0)), S, TEX, BLEND, true, SHORTA>(a, w, x, y);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(na) : : "memory");
        rec = nrec;
        da = na;
    }
}

template <bool DEPTH_ONLY, int S = 1, bool TEX = false, bool BLEND = false, bool NOCUT = false, bool SHORTA = false>
__global__ __launch_bounds__(256) void k_raster_big(RasterArgs a) {
    raster_big_body<DEPTH_ONLY, S, TEX, BLEND, NOCUT, SHORTA>(a);
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
R3N_DEV void hiz_head_body(const unsigned long long *__restrict__ vis, float *__restrict__ pyr, const r3n_hiz_desc &d, uint32_t levels,
                           uint32_t samples, float (*t)[17]) {

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
                if (samples == 0u) {  // mip 0 was filled in before this launch (the cross-rank depth merge, r3n_exchange_depth)
                    v = pyr[pix];
                } else if (samples == 1u) {
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
R3N_DEV void hiz_tail_body(float *__restrict__ pyr, const r3n_hiz_desc &d, uint32_t first, uint32_t nthreads, float *lds_a, uint32_t cap_a,
                           float *lds_b, uint32_t cap_b) {
    const float *src_lds = nullptr;  // previous level, when it was kept in LDS
    bool to_a = true;
    for (uint32_t l = first; l < d.mips; ++l) {
        const uint32_t sw = mip_dim(d.width, l - 1u), sh = mip_dim(d.height, l - 1u);
        const uint32_t dw = mip_dim(d.width, l), dh = mip_dim(d.height, l);
        const float *src = pyr + d.offset[l - 1u];
        float *dst = pyr + d.offset[l];
        float *dst_lds = dw * dh <= (to_a ? cap_a : cap_b) ? (to_a ? lds_a : lds_b) : nullptr;
        const uint32_t nx = 2u + (sw & 1u), ny = 2u + (sh & 1u);
        for (uint32_t i = threadIdx.x; i < dw * dh; i += nthreads) {
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
// (Measured and removed: building the tail inside this launch by its last block to finish -- every block fences its stores and
// takes a ticket -- costs 1.0 ms instead of 0.044: an agent-scope release fence is a write-back of the XCD's whole L2 on this
// part, and 8 160 blocks issue one each.  The tail stays a second, single-block launch.)
__global__ __launch_bounds__(256) void k_hiz_head(const unsigned long long *__restrict__ vis, float *__restrict__ pyr,
                                                  r3n_hiz_desc d, uint32_t levels, uint32_t samples) {
    __shared__ float t[16][17];
    hiz_head_body(vis, pyr, d, levels, samples, t);
}

#define R3N_HIZ_PRIO 3  // wave priority of the single-workgroup tail: it sits on the frame's serial chain while the CU it lands on is
                        // shared with the shadow lanes' and the previous frame's resolve waves (85 us average in flight against 21 alone)
__global__ __launch_bounds__(1024) void k_hiz_tail(float *__restrict__ pyr, r3n_hiz_desc d, uint32_t first) {
    __shared__ float lds_a[R3N_HIZ_LDS_A];
    __shared__ float lds_b[R3N_HIZ_LDS_B];
    __builtin_amdgcn_s_setprio(R3N_HIZ_PRIO);
    hiz_tail_body(pyr, d, first, blockDim.x, lds_a, R3N_HIZ_LDS_A, lds_b, R3N_HIZ_LDS_B);
}
