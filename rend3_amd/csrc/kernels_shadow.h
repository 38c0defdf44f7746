// kernels_shadow.h -- the directional-shadow views (base.rs:148-153: shadow_object_uniform_upload, pbr_shadow_culling,
// pbr_shadow_rendering), all views of a frame in ONE launch per stage.
//
// Reference behaviour restated (file:line):
//   uniform_prep.wgsl:9-27, batching.rs:134-236, cull.wgsl:264-303 (a shadow view stops after the sub-pixel test, :300-303)
//   depth.wgsl:51-127, forward.rs:318-371 (depth-only pipeline: cull Front, GreaterEqual + write), base.rs:366-396
//
// Shape.  The reference records, per shadow view, a uniform bake, a cull dispatch per 256 objects and an
// indirect depth draw.  Here the views of a frame are batched -- blockIdx.y = view -- so a frame has one bake, one object pass,
// one cull and one raster launch for all of them, and the depth target is never touched with memory-side atomics:
//   k_shadow_cull_bin  the triangle cull of K2 for every view; a triangle that passes gets its raster setup right there (its
//                      clip positions are in registers: no list -> object -> indices -> positions chain in a later kernel) as
//                      a 64-byte record, and a reference to it is appended to the list of every 64 x 64 texel tile its
//                      bounding box touches (appends of one wavefront to one tile fold into ONE returning atomic);
//   k_shadow_tiles     one workgroup per tile: the tile's depth lives in LDS (ds_max_u32), small triangles are scanned one
//                      per lane, larger ones cooperatively by a wavefront, and the finished tile is written with plain,
//                      fully coalesced stores.
// What does not fit -- cutout materials (they need the alpha test), triangles whose box spans more than 2 x 2 tiles, tile
// lists that are full -- goes to a per-view fallback list drawn by the general rasteriser (k_raster_small / k_raster_big in
// depth mode, atomicMax on the atlas) AFTER the tile kernel; max is idempotent, so a triangle drawn by both is harmless.
// Results (L1 / L2 sets, draw-call counts, atlas) are bit-identical to the per-view path they replace.
#pragma once
#include "kernels_cull.h"
#include "kernels_raster.h"

#define R3N_STILE 64u            // shadow tile edge in texels
#define R3N_STILE_CAP 8192u      // triangle references per tile list
#define R3N_STILE_THREADS 256u  // threads of the workgroup that owns a tile: busy tiles (thousands of triangles) set the duration of
                                 // the kernel, and a tile's work is only as parallel as its workgroup is wide

// 64 bytes: everything the scan of one triangle needs.
struct r3n_shadow_tri {
    float e[3][3];   // oriented edge functions
    float z[3];      // clip-space z per vertex
    float det;
    uint32_t xy0;    // x0 | y0 << 16 (viewport-local texel bounds, inclusive)
    uint32_t xy1;
    uint32_t _pad;
};
static_assert(sizeof(r3n_shadow_tri) == 64, "shadow triangle record is 16 dwords");

// Device-resident descriptor of one shadow view of the batch.
struct ShadowView {
    const r3n_camera_header240 *hdr;
    r3n_baked128 *baked;
    uint8_t *vis_flags;
    ObjBlockSums *block_sums;
    ObjBlockOffsets *block_off;
    r3n_vis_entry *vis_list;
    uint32_t *slot_base;
    r3n_cull_counts *counts;
    r3n_sub_counts *sub_counts;        // n[0][key][q] = passing triangles (the IndirectCall counts of cull.wgsl:63-73)
    unsigned long long *mask;          // result bits, one u64 per wave slot
    r3n_shadow_tri *recs;              // one record slot per triangle slot (wave slot * 64 + lane); written for binned triangles
    uint32_t *tile_count;              // tiles_x * tiles_x append counters
    uint32_t *tile_list;               // R3N_STILE_CAP record indices per tile
    r3n_tri_ref *fallback;             // [3][R3N_SUBQ][subcap] triangle references for the general rasteriser
    uint32_t *fb_counts;               // [3][R3N_SUBQ] lengths of those lists
    uint32_t subcap;
    uint32_t vp_x, vp_y, vp_size;      // the view's square viewport in the atlas
    uint32_t tiles_x;                  // ceil(vp_size / R3N_STILE)
    ObjOwn own;                        // object slots this view draws on this rank (multi-GPU sharding)
    uint32_t _pad;
};

struct ShadowBatchArgs {
    const ShadowView *views;
    const r3n_object128 *objects;
    ObjSoA soa;                        // spheres + (triangle count | material key) per object slot: what the object pass reads
    const uint32_t *mesh;
    const uint8_t *material_keys;
    uint32_t n_materials;
    uint32_t *atlas;                   // f32 bits
    uint32_t atlas_pitch;
    uint32_t bin_tiles;                // 1: opaque triangles are set up and binned for k_shadow_tiles; 0: every passing triangle goes
                                       //    to the per-view lists of the general rasteriser (batched per-view path, no tile pass)
};

// ------------------------------------------------------------------------------------------------ bake + object pass, batched
__global__ __launch_bounds__(256) void k_shadow_bake(ShadowBatchArgs a) {
    const ShadowView &V = a.views[blockIdx.y];
    uniform_bake_body(V.hdr, a.objects, V.baked);
}
__global__ __launch_bounds__(256) void k_shadow_object_count(ShadowBatchArgs a) {
    const ShadowView &V = a.views[blockIdx.y];
    object_count_body(V.hdr, a.soa, V.own, V.vis_flags, V.block_sums);
}
__global__ __launch_bounds__(1024) void k_shadow_object_scan(ShadowBatchArgs a, uint32_t nblocks) {
    const ShadowView &V = a.views[blockIdx.y];
    object_scan_body(V.block_sums, nblocks, V.block_off, V.counts, V.vis_list, V.sub_counts);
    // the view's other append counters start the frame at zero too
    const uint32_t ntiles = V.tiles_x * V.tiles_x;
    for (uint32_t i = threadIdx.x; i < ntiles; i += blockDim.x) V.tile_count[i] = 0u;
    for (uint32_t i = threadIdx.x; i < 3u * R3N_SUBQ; i += blockDim.x) V.fb_counts[i] = 0u;
}
__global__ __launch_bounds__(256) void k_shadow_object_scatter(ShadowBatchArgs a) {
    const ShadowView &V = a.views[blockIdx.y];
    object_scatter_body(V.hdr, a.soa, V.vis_flags, V.block_off, V.vis_list, V.slot_base, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------ cull + setup + bin
// Persistent blocks walk chunks of R3N_CHUNK_WAVES wave slots like k_triangle_cull (kernels_cull.h); the LDS-staged compaction
// remains for the fallback references, the pass counts go to the draw-call counters with one atomic per (chunk, key).
__global__ __launch_bounds__(256) void k_shadow_cull_bin(ShadowBatchArgs a) {
    __shared__ unsigned long long s_fb[4][R3N_CHUNK_ITERS];
    __shared__ uint32_t s_obj[4][R3N_CHUNK_ITERS];
    __shared__ uint32_t s_tri0[4][R3N_CHUNK_ITERS];
    __shared__ uint32_t s_key[4][R3N_CHUNK_ITERS];
    __shared__ uint32_t cnt[4][6];    // per wave: [0..3) passing triangles per key, [3..6) fallback references per key
    __shared__ uint32_t base[4][3];   // per wave: first fallback entry per key
    const ShadowView &V = a.views[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total_waves = V.counts->total_waves;
    const uint32_t nvis = V.counts->visible_objects;
    const uint32_t nchunks = (total_waves + R3N_CHUNK_WAVES - 1u) / R3N_CHUNK_WAVES;
    const uint32_t flags = V.hdr->flags;
    const bool positive_visible = (flags & R3N_PCU_POSITIVE_AREA_VISIBLE) != 0u;
    const float res_x = V.hdr->resolution[0], res_y = V.hdr->resolution[1];
    const float half = (float)V.vp_size / 2.0f;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    HizView no_hiz{};

    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint32_t w0 = chunk * R3N_CHUNK_WAVES + wave * R3N_CHUNK_ITERS;
        uint32_t e = 0, next_start = 0;
        if (w0 < total_waves) {
            uint32_t lo = 0, hi = nvis;
            while (hi - lo > 1u) {
                const uint32_t mid = lo + (hi - lo) / 2u;
                if (V.vis_list[mid].wave_start <= w0) lo = mid; else hi = mid;
            }
            e = lo;
            next_start = V.vis_list[e + 1u].wave_start;
        }
        uint32_t c_p0 = 0, c_p1 = 0, c_p2 = 0, c_f0 = 0, c_f1 = 0, c_f2 = 0;

#pragma unroll 1
        for (uint32_t it = 0; it < R3N_CHUNK_ITERS; ++it) {
            const uint32_t w = w0 + it;
            unsigned long long ballot = 0, fbm = 0;
            uint32_t obj = 0, wrel = 0, key = 0;
            if (w < total_waves) {
                while (w >= next_start) { ++e; next_start = V.vis_list[e + 1u].wave_start; }
                obj = __builtin_amdgcn_readfirstlane(V.vis_list[e].object);
                wrel = w - __builtin_amdgcn_readfirstlane(V.vis_list[e].wave_start);
                const r3n_object128 *ob = &a.objects[obj];
                const uint32_t ntri = ob->index_count / 3u;
                const uint32_t tri = wrel * 64u + lane;
                const uint32_t mi = ob->material_index;
                key = mi < a.n_materials ? a.material_keys[mi] : 0u;
                key = key > 2u ? 2u : key;
                bool pass = false, binned = false, fb = false;
                TriSetup ts;
                int x0 = 0, y0 = 0, x1 = -1, y1 = -1;
                if (tri < ntri) {
                    const uint32_t first = ob->first_index + tri * 3u;
                    const uint32_t pos_off = ob->vertex_attribute_start_offsets[0];
                    float v[3][3], p[3][4];
                    uint32_t idx[3];
                    fetch_indices3(a.mesh, first, idx);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        fetch_vec3(a.mesh, pos_off, idx[k], v[k]);
                        mul_point(V.baked[obj].model_view_proj, v[k], p[k]);
                    }
                    pass = execute_culling_clip(p, flags, true, res_x, res_y, no_hiz);
                    if (pass && key == R3N_KEY_OPAQUE && a.bin_tiles == 0u) {
                        fb = true;
                    } else if (pass && key == R3N_KEY_OPAQUE) {
                        // what prepare_triangle<DEPTH_ONLY> (kernels_raster.h) derives from the same clip positions
                        setup_triangle(p, half, half, positive_visible, ts);
                        if (ts.valid && tri_bounds(p, half, half, (int)V.vp_size, (int)V.vp_size, x0, y0, x1, y1)) {
                            const int tw = (x1 >> 6) - (x0 >> 6), th = (y1 >> 6) - (y0 >> 6);
                            if (tw <= 1 && th <= 1) binned = true; else fb = true;
                        }
                    } else if (pass && key == R3N_KEY_CUTOUT) {
                        fb = true;  // needs the alpha test: general rasteriser
                    }
                }
                ballot = __ballot(pass);
                if (lane == 0) V.mask[w] = ballot;  // cull.wgsl:229-240: result bits, 64 per wave slot
                // ---- binned triangles: record + one list entry per touched tile
                if (binned) {
                    r3n_shadow_tri rec;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) rec.e[i][c] = ts.e[i][c];
                        rec.z[i] = ts.z[i];
                    }
                    rec.det = ts.det;
                    rec.xy0 = (uint32_t)x0 | ((uint32_t)y0 << 16);
                    rec.xy1 = (uint32_t)x1 | ((uint32_t)y1 << 16);
                    rec._pad = 0u;
                    V.recs[(size_t)w * 64u + lane] = rec;
                }
                const uint32_t tx0 = (uint32_t)x0 >> 6, ty0 = (uint32_t)y0 >> 6;
                const uint32_t ntx = binned ? (uint32_t)(x1 >> 6) - tx0 + 1u : 0u, nty = binned ? (uint32_t)(y1 >> 6) - ty0 + 1u : 0u;
                bool overflow = false;
#pragma unroll 1
                for (uint32_t j = 0; j < 4u; ++j) {
                    const uint32_t jx = j & 1u, jy = j >> 1;
                    const uint32_t tile = (jx < ntx && jy < nty) ? (ty0 + jy) * V.tiles_x + (tx0 + jx) : R3N_INVALID;
                    unsigned long long rem = __ballot(tile != R3N_INVALID);
                    while (rem) {  // one returning atomic per distinct tile among the wave's triangles
                        const uint32_t leader = (uint32_t)__builtin_ctzll(rem);
                        const uint32_t t = __builtin_amdgcn_readlane(tile, leader);
                        const unsigned long long m = __ballot(tile == t);
                        uint32_t at = 0;
                        if (lane == leader) at = atomicAdd(&V.tile_count[t], (uint32_t)__popcll(m));
                        at = __builtin_amdgcn_readlane(at, leader);
                        if (tile == t) {
                            const uint32_t slot = at + (uint32_t)__popcll(m & lane_lt);
                            if (slot < R3N_STILE_CAP) V.tile_list[(size_t)t * R3N_STILE_CAP + slot] = w * 64u + lane;
                            else overflow = true;
                        }
                        rem &= ~m;
                    }
                }
                fb = fb || overflow;  // a full tile list: the general rasteriser draws the whole triangle (max is idempotent)
                fbm = __ballot(fb);
                const uint32_t np = (uint32_t)__popcll(ballot), nf = (uint32_t)__popcll(fbm);
                c_p0 += key == 0u ? np : 0u; c_p1 += key == 1u ? np : 0u; c_p2 += key == 2u ? np : 0u;
                c_f0 += key == 0u ? nf : 0u; c_f1 += key == 1u ? nf : 0u; c_f2 += key == 2u ? nf : 0u;
            }
            if (lane == 0) {
                s_fb[wave][it] = fbm;
                s_obj[wave][it] = obj; s_tri0[wave][it] = wrel * 64u; s_key[wave][it] = key;
            }
        }

        if (lane == 0) {
            cnt[wave][0] = c_p0; cnt[wave][1] = c_p1; cnt[wave][2] = c_p2;
            cnt[wave][3] = c_f0; cnt[wave][4] = c_f1; cnt[wave][5] = c_f2;
        }
        __syncthreads();
        if (threadIdx.x < 6u) {
            const uint32_t k = threadIdx.x;
            const uint32_t c0 = cnt[0][k], c1 = cnt[1][k], c2 = cnt[2][k], c3 = cnt[3][k];
            const uint32_t tot = c0 + c1 + c2 + c3;
            if (k < 3u) {  // cull.wgsl:63-73: the region's vertex_count / 3
                if (tot) atomicAdd(&V.sub_counts->n[0][k][chunk % R3N_SUBQ], tot);
            } else {
                uint32_t start = 0;
                if (tot) start = atomicAdd(&V.fb_counts[(k - 3u) * R3N_SUBQ + (chunk % R3N_SUBQ)], tot);
                base[0][k - 3u] = start; base[1][k - 3u] = start + c0; base[2][k - 3u] = start + c0 + c1; base[3][k - 3u] = start + c0 + c1 + c2;
            }
        }
        __syncthreads();
        uint32_t run[3] = {base[wave][0], base[wave][1], base[wave][2]};
#pragma unroll 1
        for (uint32_t it = 0; it < R3N_CHUNK_ITERS; ++it) {
            const unsigned long long fb = s_fb[wave][it];
            if (fb == 0ull) continue;
            const uint32_t key = s_key[wave][it];
            const r3n_tri_ref ref = {s_obj[wave][it], s_tri0[wave][it] + lane};
            const uint32_t nf = (uint32_t)__popcll(fb);
            const uint32_t r = key == 0u ? run[0] : (key == 1u ? run[1] : run[2]);
            const uint32_t region = (key * R3N_SUBQ + (chunk % R3N_SUBQ)) * V.subcap;
            if ((fb >> lane) & 1ull) V.fallback[region + r + (uint32_t)__popcll(fb & lane_lt)] = ref;
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k) run[k] += key == k ? nf : 0u;
        }
        __syncthreads();  // LDS staging reused by the next chunk
    }
}

// ------------------------------------------------------------------------------------------------ tile raster
// One fragment of a binned triangle into the LDS tile: coverage (top-left rule), depth clip, GreaterEqual + write == max.
R3N_DEV void shadow_tile_pixel(const TriSetup &ts, int x, int y, int tx0, int ty0, uint32_t *depth) {
    float E[3];
    if (!edge_eval(ts, (float)x + 0.5f, (float)y + 0.5f, E)) return;
    float z = frag_depth(ts, (float)x + 0.5f, (float)y + 0.5f);
    if (!(z >= 0.0f && z <= 1.0f)) return;  // depth clip (unclipped_depth: false, forward.rs:343)
    if (z == 0.0f) z = 0.0f;                // canonicalise -0
    atomicMax(&depth[(uint32_t)(y - ty0) * R3N_STILE + (uint32_t)(x - tx0)], __float_as_uint(z));
}
R3N_DEV void shadow_load_tri(const r3n_shadow_tri *__restrict__ recs, uint32_t idx, TriSetup &ts, int &x0, int &y0, int &x1, int &y1) {
    const float4 *p = reinterpret_cast<const float4 *>(recs + idx);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    ts.e[0][0] = a.x; ts.e[0][1] = a.y; ts.e[0][2] = a.z; ts.e[1][0] = a.w;
    ts.e[1][1] = b.x; ts.e[1][2] = b.y; ts.e[2][0] = b.z; ts.e[2][1] = b.w;
    ts.e[2][2] = c.x; ts.z[0] = c.y; ts.z[1] = c.z; ts.z[2] = c.w;
    ts.det = d.x; ts.valid = true;
    const uint32_t xy0 = __float_as_uint(d.y), xy1 = __float_as_uint(d.z);
    x0 = (int)(xy0 & 0xFFFFu); y0 = (int)(xy0 >> 16); x1 = (int)(xy1 & 0xFFFFu); y1 = (int)(xy1 >> 16);
}

// The list is walked in rounds of R3N_STILE_THREADS entries (one per thread).  A thread whose triangle's box inside the tile exceeds 8 x 8
// texels parks the whole 64-byte record in an LDS queue, which the workgroup's wavefronts drain cooperatively at the end of the
// round: the wave-per-triangle pass then reads LDS, not memory (with the records left in memory that pass was a chain of
// dependent global loads, one per triangle and wavefront, and set the duration of the kernel on busy tiles).
__global__ __launch_bounds__(R3N_STILE_THREADS) void k_shadow_tiles(ShadowBatchArgs a) {
    __shared__ uint32_t depth[R3N_STILE * R3N_STILE];
    __shared__ float4 bigq[R3N_STILE_THREADS][4];
    __shared__ uint32_t nbig;
    const ShadowView &V = a.views[blockIdx.y];
    const uint32_t tile = blockIdx.x;
    if (tile >= V.tiles_x * V.tiles_x) return;
    const int tx0 = (int)((tile % V.tiles_x) * R3N_STILE), ty0 = (int)((tile / V.tiles_x) * R3N_STILE);
    const int tx1 = min(tx0 + (int)R3N_STILE - 1, (int)V.vp_size - 1), ty1 = min(ty0 + (int)R3N_STILE - 1, (int)V.vp_size - 1);
    for (uint32_t i = threadIdx.x; i < R3N_STILE * R3N_STILE; i += R3N_STILE_THREADS) depth[i] = 0u;  // depth clear 0.0 (clear.rs:4-20)
    const uint32_t n = min(V.tile_count[tile], R3N_STILE_CAP);
    const uint32_t *list = V.tile_list + (size_t)tile * R3N_STILE_CAP;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const int lx = (int)(lane & 7u), ly = (int)(lane >> 3);
    for (uint32_t first = 0; first < n; first += R3N_STILE_THREADS) {
        if (threadIdx.x == 0u) nbig = 0u;
        __syncthreads();  // (also: the clear above / the previous round's queue reads)
        // pass 1: one triangle per lane; boxes up to 8 x 8 texels inside the tile are scanned in place
        const uint32_t i = first + threadIdx.x;
        if (i < n) {
            const float4 *p = reinterpret_cast<const float4 *>(V.recs + list[i]);
            const float4 r0 = p[0], r1 = p[1], r2 = p[2], r3 = p[3];
            const uint32_t xy0 = __float_as_uint(r3.y), xy1 = __float_as_uint(r3.z);
            const int x0 = max((int)(xy0 & 0xFFFFu), tx0), y0 = max((int)(xy0 >> 16), ty0);
            const int x1 = min((int)(xy1 & 0xFFFFu), tx1), y1 = min((int)(xy1 >> 16), ty1);
            if (x1 >= x0 && y1 >= y0) {
                if (x1 - x0 < 8 && y1 - y0 < 8) {
                    TriSetup ts;
                    ts.e[0][0] = r0.x; ts.e[0][1] = r0.y; ts.e[0][2] = r0.z; ts.e[1][0] = r0.w;
                    ts.e[1][1] = r1.x; ts.e[1][2] = r1.y; ts.e[2][0] = r1.z; ts.e[2][1] = r1.w;
                    ts.e[2][2] = r2.x; ts.z[0] = r2.y; ts.z[1] = r2.z; ts.z[2] = r2.w;
                    ts.det = r3.x; ts.valid = true;
                    for (int y = y0; y <= y1; ++y)
                        for (int x = x0; x <= x1; ++x) shadow_tile_pixel(ts, x, y, tx0, ty0, depth);
                } else {
                    const uint32_t q = atomicAdd(&nbig, 1u);  // < R3N_STILE_THREADS: one entry per thread and round
                    bigq[q][0] = r0; bigq[q][1] = r1; bigq[q][2] = r2; bigq[q][3] = r3;
                }
            }
        }
        __syncthreads();
        // pass 2: the larger ones, one wavefront per triangle: lane = 8 x 8 block for the exact rejection test, then lane = texel
        const uint32_t nb = nbig;
        for (uint32_t b = wave; b < nb; b += R3N_STILE_THREADS / 64u) {
            const float4 r0 = bigq[b][0], r1 = bigq[b][1], r2 = bigq[b][2], r3 = bigq[b][3];  // LDS broadcast reads
            TriSetup ts;
            ts.e[0][0] = r0.x; ts.e[0][1] = r0.y; ts.e[0][2] = r0.z; ts.e[1][0] = r0.w;
            ts.e[1][1] = r1.x; ts.e[1][2] = r1.y; ts.e[2][0] = r1.z; ts.e[2][1] = r1.w;
            ts.e[2][2] = r2.x; ts.z[0] = r2.y; ts.z[1] = r2.z; ts.z[2] = r2.w;
            ts.det = r3.x; ts.valid = true;
            const uint32_t xy0 = __float_as_uint(r3.y), xy1 = __float_as_uint(r3.z);
            const int x0 = max((int)(xy0 & 0xFFFFu), tx0), y0 = max((int)(xy0 >> 16), ty0);
            const int x1 = min((int)(xy1 & 0xFFFFu), tx1), y1 = min((int)(xy1 >> 16), ty1);
            if (x1 - x0 < 32 && y1 - y0 < 32) {
                // boxes up to 32 x 32: lane = 4 x 4 block for the rejection test; every step then scans FOUR surviving blocks,
                // 16 lanes each -- mid-sized triangles fill the wave far better than with 8 x 8 blocks
                const int cbx = x0 + lx * 4, cby = y0 + ly * 4;
                const bool cand = cbx <= x1 && cby <= y1 && block_may_cover<4, false>(ts, cbx, cby, x1, y1);
                unsigned long long blocks = __ballot(cand);
                const uint32_t grp = lane >> 4;
                const int px = (int)(lane & 3u), py = (int)((lane >> 2) & 3u);
                while (blocks) {
                    int bsel[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bsel[g] = blocks ? __builtin_ctzll(blocks) : 64;
                        blocks &= blocks - 1ull;
                    }
                    const int b = grp == 0u ? bsel[0] : (grp == 1u ? bsel[1] : (grp == 2u ? bsel[2] : bsel[3]));
                    const int x = x0 + (b & 7) * 4 + px, y = y0 + (b >> 3) * 4 + py;
                    if (b < 64 && x <= x1 && y <= y1) shadow_tile_pixel(ts, x, y, tx0, ty0, depth);
                }
            } else {
                const int cbx = x0 + lx * 8, cby = y0 + ly * 8;
                const bool cand = cbx <= x1 && cby <= y1 && block_may_cover<8, false>(ts, cbx, cby, x1, y1);
                unsigned long long blocks = __ballot(cand);
                while (blocks) {
                    const int bsel = __builtin_ctzll(blocks);
                    blocks &= blocks - 1ull;
                    const int x = x0 + (bsel & 7) * 8 + lx, y = y0 + (bsel >> 3) * 8 + ly;
                    if (x <= x1 && y <= y1) shadow_tile_pixel(ts, x, y, tx0, ty0, depth);
                }
            }
        }
    }
    __syncthreads();
    // the finished tile, a 256-byte row segment per wavefront: MERGED with what the atlas already holds (max == the depth test) --
    // the stages of a view can reach the atlas in more than one flush (a read-back or an exchange between a view's opaque and
    // cutout draws), and a plain store would drop what an earlier flush rasterised there.  The tile is this workgroup's alone
    // during the launch, so a read + max + store is enough.
    for (uint32_t i = threadIdx.x; i < R3N_STILE * R3N_STILE; i += R3N_STILE_THREADS) {
        const uint32_t x = (uint32_t)tx0 + (i & (R3N_STILE - 1u)), y = (uint32_t)ty0 + (i / R3N_STILE);
        if (x < V.vp_size && y < V.vp_size) {
            uint32_t *t = &a.atlas[(size_t)(V.vp_y + y) * a.atlas_pitch + V.vp_x + x];
            const uint32_t have = *t;
            if (depth[i] > have) *t = depth[i];
        }
    }
}
