// kernels_cull.h -- uniform bake (K1), object-level frustum cull + device-side slot assignment (B1 moved
// to the GPU), per-triangle cull + LDS-staged compaction into per-region indirect calls (K2).
//
// Reference behaviour restated (file:line):
//   uniform_prep.wgsl:9-27, culler.rs:427-529          k_uniform_bake
//   batching.rs:134-170,191-236, frustum.rs:148-161    k_object_count / k_object_scan / k_object_scatter
//   cull.wgsl:264-324 (execute_culling), :326-390      k_triangle_cull
//
// Shape changes vs. the reference (results identical): no 256-object batches, no CPU sort,
// one launch per camera; objects own whole wavefronts (64 triangle slots) so matrices are wave-uniform;
// compaction = wave ballot + popcount, staged through LDS per 4096-slot chunk, one global atomic per
// (chunk, list, region) instead of one per triangle, spread over R3N_SUBQ sub-lists per region.
#pragma once
#include "device_math.h"

// ------------------------------------------------------------------------------------------------ K1
// 4 threads per object: thread c bakes column c of model_view and model_view_proj.  Consecutive threads
// read consecutive 16-byte columns (fully coalesced) and write two 16-byte columns.
R3N_DEV void uniform_bake_body(const r3n_camera_header240 *__restrict__ hdr, const r3n_object128 *__restrict__ objects,
                                r3n_baked128 *__restrict__ baked) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t obj = gid >> 2, c = gid & 3u;
    if (obj >= hdr->object_count) return;            // uniform_prep.wgsl:15-17
    if (objects[obj].enabled == 0u) return;          // :18-20 (stale matrices stay, App. D.2)
    const float4 col = reinterpret_cast<const float4 *>(objects[obj].transform)[c];
    float mv[4], mvp[4];
    mul_vec4(hdr->view, col.x, col.y, col.z, col.w, mv);
    mul_vec4(hdr->view_proj, col.x, col.y, col.z, col.w, mvp);
    reinterpret_cast<float4 *>(baked[obj].model_view)[c] = make_float4(mv[0], mv[1], mv[2], mv[3]);
    reinterpret_cast<float4 *>(baked[obj].model_view_proj)[c] = make_float4(mvp[0], mvp[1], mvp[2], mvp[3]);
}
__global__ __launch_bounds__(256) void k_uniform_bake(const r3n_camera_header240 *__restrict__ hdr,
                                                      const r3n_object128 *__restrict__ objects,
                                                      r3n_baked128 *__restrict__ baked) {
    uniform_bake_body(hdr, objects, baked);
}

// ------------------------------------------------------------------------------------------------ object pass
struct ObjBlockSums {
    uint32_t visible, waves, tris_all, key_tris[3];
};
// Which opaque / cutout object slots this context culls and draws for a camera (multi-GPU sharding; everything on one GPU):
// a contiguous slot range, or -- `owners` non-null -- the slots whose owner byte equals `rank` (spatial partitions: the slots of a
// compact region of the world are not contiguous).
struct ObjOwn {
    uint32_t begin, end;
    const uint8_t *owners;
    uint32_t rank;
};
R3N_DEV bool obj_owned(const ObjOwn &o, uint32_t i) {
    return o.owners != nullptr ? (uint32_t)o.owners[i] == o.rank : (i >= o.begin && i < o.end);
}

R3N_DEV uint32_t wave_reduce_add(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

R3N_DEV uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t n = __shfl_up(v, o, 64);
        if (lane >= (uint32_t)o) v += n;
    }
    return v;
}

// Structure-of-arrays view of what the object pass reads of an object (BASELINE.json north_star: "object transforms, bounding
// spheres, material data ... live in HBM as structure-of-arrays").  rend3's 128-byte record (object.rs:23-36) stays the INPUT
// contract of r3n_objects_write and the layout of the kernels that want a whole record; the object pass -- one thread per object
// slot of the whole world, per camera -- reads 20 coalesced bytes per slot instead of four fields out of its own 128-byte line:
//   spheres[i] = world-space bounding sphere (centre.xyz, radius)                      object.rs:30-31
//   meta[i]    = (enabled ? index_count / 3 : 0) | Material::key() of its material << 30   object.rs:33,36; pbr/material.rs:497-499
// r3n_objects_write scatters both beside the record; r3n_materials_write refreshes the key bits (r3n.hip refresh_obj_meta).
struct ObjSoA {
    const float4 *__restrict__ spheres;
    const uint32_t *__restrict__ meta;
};
#define R3N_META_NTRI_MASK 0x3FFFFFFFu
#define R3N_META_KEY_SHIFT 30u

// batching.rs:146 + frustum.rs:148-161 on one object of the SoA view.  Two bits, the byte a camera's vis_flags holds per slot:
//   R3N_VIS_INSIDE  the object has triangles and its sphere passes the frustum test -- whoever owns it.  This frame's and last
//                   frame's bit decide which slots the chained pass bakes (k_object_pass_chained): a rank also RESOLVES pixels of
//                   objects other ranks drew (their keys arrive through the exchange), and the resolve reads model_view;
//   R3N_VIS_DRAWN   ... and this rank culls and draws it: the reference's visible set (L1).  Multi-rank sharding: a rank owns the
//                   opaque / cutout objects of its slot range; blend-key objects are culled and drawn by every rank (ordered
//                   blending cannot be merged by the MAX reduce of the depth keys).
#define R3N_VIS_DRAWN 1u
#define R3N_VIS_INSIDE 2u
R3N_DEV uint32_t object_visible(const r3n_camera_header240 *__restrict__ hdr, const ObjOwn &own, uint32_t i, uint32_t meta, const float4 sph) {
    const uint32_t ntri = meta & R3N_META_NTRI_MASK, key0 = meta >> R3N_META_KEY_SHIFT;
    if (ntri == 0u) return 0u;
    const float c[3] = {sph.x, sph.y, sph.z};
    const float neg_radius = -sph.w;
    bool inside = true;
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const float d = dot3(hdr->frustum + 4 * p, c) + hdr->frustum[4 * p + 3];
        inside = inside && (d >= neg_radius);
    }
    if (!inside) return 0u;
    return R3N_VIS_INSIDE | ((obj_owned(own, i) || key0 == 2u) ? R3N_VIS_DRAWN : 0u);
}

// first_entry[k] = the work-list entry that owns wave slot k * R3N_CHUNK_ITERS: where a wavefront of k_triangle_cull starts
// (it replaces the reference's per-invocation binary search, cull.wgsl:181-207, and this implementation's former 12-step scalar
// search per chunk).  Entry e owns wave slots [ws, ws + nw); it writes every k with ws <= k * R3N_CHUNK_ITERS < ws + nw: up to
// eight stores by its own thread, larger objects by the whole wavefront.  Called convergently (ballot + lane reads inside).
#define R3N_CHUNK_ITERS 4u                      // wave slots per wavefront per chunk
R3N_DEV void write_first_entries(uint32_t *__restrict__ first_entry, uint32_t flag, uint32_t e, uint32_t ws, uint32_t nw, uint32_t lane) {
    if (first_entry == nullptr) return;
    uint32_t k_lo = 0u, cnt = 0u;
    if (flag && nw) {
        k_lo = (ws + R3N_CHUNK_ITERS - 1u) / R3N_CHUNK_ITERS;
        const uint32_t k_end = (ws + nw - 1u) / R3N_CHUNK_ITERS + 1u;
        cnt = k_end > k_lo ? k_end - k_lo : 0u;
    }
    if (cnt <= 8u)  // (objects of up to 2 048 triangles: a short divergent loop of stores)
        for (uint32_t k = 0; k < cnt; ++k) first_entry[k_lo + k] = e;
    unsigned long long big = __ballot(cnt > 8u);
    while (big) {
        const int l = __ffsll((long long)big) - 1;
        big &= big - 1ull;
        // (l is wave-uniform: v_readlane, not a ds_bpermute round trip per value)
        const uint32_t bk = (uint32_t)__builtin_amdgcn_readlane((int)k_lo, l), bc = (uint32_t)__builtin_amdgcn_readlane((int)cnt, l),
                       be = (uint32_t)__builtin_amdgcn_readlane((int)e, l);
        for (uint32_t k = lane; k < bc; k += 64u) first_entry[bk + k] = be;
    }
}

// Pass A: frustum test (batching.rs:146, frustum.rs:148-161) + per-block totals.
R3N_DEV void object_count_body(const r3n_camera_header240 *__restrict__ hdr, ObjSoA soa, ObjOwn own,
                                uint8_t *__restrict__ vis_flags, ObjBlockSums *__restrict__ block_sums) {
    __shared__ uint32_t red[4][6];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t cap = hdr->object_count;
    uint32_t flag = 0, ntri_all = 0, ntri_vis = 0, key = 0;
    if (i < cap) {
        const uint32_t meta = soa.meta[i];
        const float4 sph = soa.spheres[i];
        ntri_all = meta & R3N_META_NTRI_MASK;
        const uint32_t bits = object_visible(hdr, own, i, meta, sph);
        flag = bits & R3N_VIS_DRAWN;
        if (flag) { ntri_vis = ntri_all; key = meta >> R3N_META_KEY_SHIFT; }
        vis_flags[i] = (uint8_t)bits;
    }
    uint32_t vals[6] = {flag, flag ? (ntri_vis + 63u) / 64u : 0u, ntri_all, key == 0u ? ntri_vis : 0u,
                        key == 1u ? ntri_vis : 0u, key == 2u ? ntri_vis : 0u};
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t s = wave_reduce_add(vals[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const uint32_t k = threadIdx.x;
        const uint32_t s = red[0][k] + red[1][k] + red[2][k] + red[3][k];
        uint32_t *dst = reinterpret_cast<uint32_t *>(&block_sums[blockIdx.x]);
        dst[k] = s;
    }
}
__global__ __launch_bounds__(256) void k_object_count(const r3n_camera_header240 *__restrict__ hdr, ObjSoA soa,
                                                      ObjOwn own, uint8_t *__restrict__ vis_flags,
                                                      ObjBlockSums *__restrict__ block_sums) {
    object_count_body(hdr, soa, own, vis_flags, block_sums);
}

struct ObjBlockOffsets {
    uint32_t visible, waves, tris_all;
};

// Pass B: one block scans the per-block totals, derives region bases and resets the append counters
// (culler.rs:642 clear_buffer + cull.wgsl:47-61 init_draw_calls).
R3N_DEV void object_scan_body(const ObjBlockSums *__restrict__ block_sums, uint32_t nblocks, ObjBlockOffsets *__restrict__ block_off,
                               r3n_cull_counts *__restrict__ counts, r3n_vis_entry *__restrict__ vis_list,
                               r3n_sub_counts *__restrict__ sub_counts) {
    __shared__ uint32_t sh[3][1024];
    __shared__ uint32_t carry[3];
    __shared__ uint32_t ktot[3];
    const uint32_t t = threadIdx.x;
    if (blockDim.x == 64u) {
        // up to 64 block totals (16 384 object slots): ONE wavefront, scans by lane shuffles, no LDS and no barrier.  A
        // 1024-thread workgroup needs half a CU's wave slots at once and waits for them while the resolve of the previous
        // frame fills the chip (5 us alone, 20+ us in flight); one wavefront starts anywhere.
        for (uint32_t k = t; k < 2u * 3u * R3N_SUBQ; k += 64u) (&sub_counts->n[0][0][0])[k] = 0u;
        uint32_t v[3] = {0, 0, 0}, key[3] = {0, 0, 0};
        if (t < nblocks) {
            v[0] = block_sums[t].visible; v[1] = block_sums[t].waves; v[2] = block_sums[t].tris_all;
            key[0] = block_sums[t].key_tris[0]; key[1] = block_sums[t].key_tris[1]; key[2] = block_sums[t].key_tris[2];
        }
        uint32_t inc[3], tot[3], ktotal[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            inc[k] = wave_inclusive_scan(v[k], t);
            tot[k] = __shfl(inc[k], 63, 64);
            ktotal[k] = __shfl(wave_inclusive_scan(key[k], t), 63, 64);
        }
        if (t < nblocks) {
            block_off[t].visible = inc[0] - v[0];
            block_off[t].waves = inc[1] - v[1];
            block_off[t].tris_all = inc[2] - v[2];
        }
        if (t == 0u) {
            counts->visible_objects = tot[0];
            counts->total_waves = tot[1];
            counts->total_triangles = tot[2];
            uint32_t base = 0;
            for (int k = 0; k < 3; ++k) {
                counts->key_triangles[k] = ktotal[k];
                counts->region_base[k] = base;
                base += ktotal[k];
            }
            vis_list[tot[0]].object = R3N_INVALID;  // sentinel
            vis_list[tot[0]].wave_start = tot[1];
        }
        return;
    }
    if (t < 3) { carry[t] = 0; ktot[t] = 0; }
    if (t < 2u * 3u * R3N_SUBQ) (&sub_counts->n[0][0][0])[t] = 0u;
    __syncthreads();
    uint32_t kacc[3] = {0, 0, 0};
    for (uint32_t base = 0; base < nblocks; base += 1024u) {
        const uint32_t b = base + t;
        uint32_t v[3] = {0, 0, 0};
        if (b < nblocks) {
            v[0] = block_sums[b].visible; v[1] = block_sums[b].waves; v[2] = block_sums[b].tris_all;
            kacc[0] += block_sums[b].key_tris[0]; kacc[1] += block_sums[b].key_tris[1]; kacc[2] += block_sums[b].key_tris[2];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) sh[k][t] = v[k];
        __syncthreads();
        // Hillis-Steele inclusive scan over 1024 entries (tiny: one block, runs once per camera)
        for (uint32_t off = 1; off < 1024u; off <<= 1) {
            uint32_t add[3] = {0, 0, 0};
            if (t >= off) {
#pragma unroll
                for (int k = 0; k < 3; ++k) add[k] = sh[k][t - off];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 3; ++k) sh[k][t] += add[k];
            __syncthreads();
        }
        if (b < nblocks) {
            block_off[b].visible = carry[0] + sh[0][t] - v[0];
            block_off[b].waves = carry[1] + sh[1][t] - v[1];
            block_off[b].tris_all = carry[2] + sh[2][t] - v[2];
        }
        __syncthreads();
        if (t < 3) carry[t] += sh[t][1023];
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (kacc[k]) atomicAdd(&ktot[k], kacc[k]);
    __syncthreads();
    if (t == 0) {
        counts->visible_objects = carry[0];
        counts->total_waves = carry[1];
        counts->total_triangles = carry[2];
        uint32_t base = 0;
        for (int k = 0; k < 3; ++k) {
            counts->key_triangles[k] = ktot[k];
            counts->region_base[k] = base;
            base += ktot[k];
        }
        vis_list[carry[0]].object = R3N_INVALID;  // sentinel
        vis_list[carry[0]].wave_start = carry[1];
    }
}
__global__ __launch_bounds__(1024) void k_object_scan(const ObjBlockSums *__restrict__ block_sums, uint32_t nblocks,
                                                      ObjBlockOffsets *__restrict__ block_off,
                                                      r3n_cull_counts *__restrict__ counts,
                                                      r3n_vis_entry *__restrict__ vis_list,
                                                      r3n_sub_counts *__restrict__ sub_counts) {
    object_scan_body(block_sums, nblocks, block_off, counts, vis_list, sub_counts);
}


// Pass C: scatter visible objects into the work list (object-slot order => deterministic layout).
// slot_base[i] = first triangle slot of object i in this frame's result bitmask, or INVALID when the object
// was not batched (== "not in current_invocation_map", batching.rs:226,230).  tri_base[i] = canonical base.
R3N_DEV void object_scatter_body(const r3n_camera_header240 *__restrict__ hdr, ObjSoA soa,
                                  const uint8_t *__restrict__ vis_flags, const ObjBlockOffsets *__restrict__ block_off,
                                  r3n_vis_entry *__restrict__ vis_list, uint32_t *__restrict__ slot_base,
                                  uint32_t *__restrict__ tri_base, uint32_t *__restrict__ first_entry) {
    __shared__ uint32_t wtot[4][3];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t cap = hdr->object_count;
    uint32_t flag = 0, nw = 0, ntri = 0;
    if (i < cap) {
        flag = vis_flags[i] & R3N_VIS_DRAWN;
        ntri = soa.meta[i] & R3N_META_NTRI_MASK;
        nw = flag ? (ntri + 63u) / 64u : 0u;
    }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t s0 = wave_inclusive_scan(flag, lane);
    const uint32_t s1 = wave_inclusive_scan(nw, lane);
    const uint32_t s2 = wave_inclusive_scan(ntri, lane);
    if (lane == 63u) { wtot[wave][0] = s0; wtot[wave][1] = s1; wtot[wave][2] = s2; }
    __syncthreads();
    uint32_t p0 = block_off[blockIdx.x].visible, p1 = block_off[blockIdx.x].waves, p2 = block_off[blockIdx.x].tris_all;
    for (uint32_t w = 0; w < wave; ++w) { p0 += wtot[w][0]; p1 += wtot[w][1]; p2 += wtot[w][2]; }
    const uint32_t e = p0 + s0 - flag, ws = p1 + s1 - nw;
    if (i < cap) {
        if (flag) {
            vis_list[e].object = i;
            vis_list[e].wave_start = ws;
        }
        slot_base[i] = flag ? ws * 64u : R3N_INVALID;
        if (tri_base != nullptr) tri_base[i] = p2 + s2 - ntri;
    }
    write_first_entries(first_entry, flag, e, ws, nw, lane);
}
__global__ __launch_bounds__(256) void k_object_scatter(const r3n_camera_header240 *__restrict__ hdr, ObjSoA soa,
                                                        const uint8_t *__restrict__ vis_flags,
                                                        const ObjBlockOffsets *__restrict__ block_off,
                                                        r3n_vis_entry *__restrict__ vis_list,
                                                        uint32_t *__restrict__ slot_base,
                                                        uint32_t *__restrict__ tri_base, uint32_t *__restrict__ first_entry) {
    object_scatter_body(hdr, soa, vis_flags, block_off, vis_list, slot_base, tri_base, first_entry);
}

// The three passes above in ONE single-block launch, for worlds of up to R3N_FUSED_OBJECT_PASS_MAX object slots: the block
// walks the slots 1024 at a time -- frustum test, block-wide exclusive scan (wave scans + LDS) with a running carry, scatter
// -- and finishes with the totals.  Same outputs, bit for bit (the layout is slot order either way).  Measured on the bench
// scene (4096 slots, four rounds of dependent record loads in one block): 23 us per camera against 14 us for the three
// launches, so the limit is one round; the small scenes of the tests and examples run through it.
#define R3N_FUSED_OBJECT_PASS_MAX 1024u
R3N_DEV void object_pass_fused_body(const r3n_camera_header240 *__restrict__ hdr, ObjSoA soa, ObjOwn own,
                                    uint8_t *__restrict__ vis_flags, r3n_cull_counts *__restrict__ counts,
                                    r3n_vis_entry *__restrict__ vis_list, r3n_sub_counts *__restrict__ sub_counts,
                                    uint32_t *__restrict__ slot_base, uint32_t *__restrict__ tri_base,
                                    uint32_t *__restrict__ first_entry) {
    __shared__ uint32_t wtot[16][3];
    __shared__ uint32_t carry[3];
    __shared__ uint32_t ktot[3];
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
    const uint32_t cap = hdr->object_count;
    if (t < 3u) { carry[t] = 0u; ktot[t] = 0u; }
    if (t < 2u * 3u * R3N_SUBQ) (&sub_counts->n[0][0][0])[t] = 0u;
    __syncthreads();
    uint32_t kacc[3] = {0, 0, 0};
    for (uint32_t base = 0; base < cap; base += 1024u) {
        const uint32_t i = base + t;
        uint32_t flag = 0, ntri = 0, nw = 0;
        if (i < cap) {
            const uint32_t meta = soa.meta[i];
            const float4 sph = soa.spheres[i];
            ntri = meta & R3N_META_NTRI_MASK;
            const uint32_t bits = object_visible(hdr, own, i, meta, sph);
            flag = bits & R3N_VIS_DRAWN;
            if (flag) {
                nw = (ntri + 63u) / 64u;
                kacc[meta >> R3N_META_KEY_SHIFT] += ntri;
            }
            vis_flags[i] = (uint8_t)bits;
        }
        const uint32_t s0 = wave_inclusive_scan(flag, lane), s1 = wave_inclusive_scan(nw, lane), s2 = wave_inclusive_scan(ntri, lane);
        if (lane == 63u) { wtot[wave][0] = s0; wtot[wave][1] = s1; wtot[wave][2] = s2; }
        __syncthreads();
        uint32_t p0 = carry[0], p1 = carry[1], p2 = carry[2];
        for (uint32_t w = 0; w < wave; ++w) { p0 += wtot[w][0]; p1 += wtot[w][1]; p2 += wtot[w][2]; }
        const uint32_t e = p0 + s0 - flag, ws = p1 + s1 - nw;
        if (i < cap) {
            if (flag) {
                vis_list[e].object = i;
                vis_list[e].wave_start = ws;
            }
            slot_base[i] = flag ? ws * 64u : R3N_INVALID;
            if (tri_base != nullptr) tri_base[i] = p2 + s2 - ntri;
        }
        write_first_entries(first_entry, flag, e, ws, nw, lane);
        __syncthreads();  // everyone has read carry / wtot
        if (t == 1023u) { carry[0] = p0 + s0; carry[1] = p1 + s1; carry[2] = p2 + s2; }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (kacc[k]) atomicAdd(&ktot[k], kacc[k]);
    __syncthreads();
    if (t == 0u) {
        counts->visible_objects = carry[0];
        counts->total_waves = carry[1];
        counts->total_triangles = carry[2];
        uint32_t rb = 0;
        for (int k = 0; k < 3; ++k) {
            counts->key_triangles[k] = ktot[k];
            counts->region_base[k] = rb;
            rb += ktot[k];
        }
        vis_list[carry[0]].object = R3N_INVALID;  // sentinel
        vis_list[carry[0]].wave_start = carry[1];
    }
}
__global__ __launch_bounds__(1024) void k_object_pass_fused(const r3n_camera_header240 *__restrict__ hdr, ObjSoA soa,
                                                            ObjOwn own, uint8_t *__restrict__ vis_flags,
                                                            r3n_cull_counts *__restrict__ counts, r3n_vis_entry *__restrict__ vis_list,
                                                            r3n_sub_counts *__restrict__ sub_counts, uint32_t *__restrict__ slot_base,
                                                            uint32_t *__restrict__ tri_base, uint32_t *__restrict__ first_entry) {
    object_pass_fused_body(hdr, soa, own, vis_flags, counts, vis_list, sub_counts, slot_base, tri_base, first_entry);
}

// Uniform bake + the three object passes in ONE multi-block launch (r3n_render_frame's path).  A block owns `rounds` x 256
// consecutive object slots:
//   phase 1  per round: 20 B of the SoA view per slot (four rounds' loads in flight together), frustum test, the wave's visible
//            bits / bake bits / triangle counts into LDS, running totals;
//   publish  the block's six totals; read the totals of every block in front (the exclusive prefix it needs; agent-scope atomics:
//            the L2 of another XCD is not coherent for plain loads; no fences: every published word carries the launch's epoch);
//   phase 3  per round and wave, without a barrier: list entry + slot base + first-entry table, and the UNIFORM BAKE
//            (uniform_prep.wgsl:9-27) of the slots that need one.
// Same lists bit for bit as k_object_count / k_object_scan / k_object_scatter (slot order either way).  A block waits only for
// blocks with LOWER indices; the host keeps the grid at R3N_CHAINED_OBJECT_PASS_MAX_BLOCKS blocks at most (more rounds per block
// instead: every block reads every record in front of it) and falls back to the three launches beyond
// MAX_BLOCKS x MAX_ROUNDS x 256 slots.  Per camera and frame this is 1 launch instead of 4.
//
// Which slots are baked: the reference bakes every enabled slot of the buffer (uniform_prep.wgsl:15-20) -- 196 B per slot and
// camera, the largest stream of the front end on a million-object world -- but a baked matrix is read only through a triangle
// list: this frame's lists hold objects that pass the frustum test now, last frame's predicted list (drawn in this frame's first
// pass WITH THIS FRAME'S MATRICES, forward.rs:224-232) objects that passed it then.  So: R3N_VIS_INSIDE now, or
// (viewport camera) in the byte the camera's previous object pass left in vis_flags -- the frustum bit, not the drawn bit: under
// a multi-rank split the resolve also reads the matrices of objects other ranks drew.  Every matrix any kernel reads is the
// reference's, the rest keep whatever they held (the `baked` parity tap compares the frustum-visible slots).
#define R3N_CHAINED_OBJECT_PASS_MAX_BLOCKS 512u
#define R3N_CHAINED_OBJECT_PASS_MAX_ROUNDS 16u
struct ObjChainRec {
    unsigned long long v[6];  // epoch of the launch that wrote it << 32 | visible, waves, tris_all, key_tris[3]: every word validates
                              // itself, so publishing needs no release fence (an agent-scope release is a write-back of the XCD's
                              // whole L2 on this part: measured 1 ms for 8 160 of them in a Hi-Z experiment)
    unsigned long long _pad[2];
};
struct ObjChainArgs {
    const r3n_camera_header240 *__restrict__ hdr;
    ObjSoA soa;
    const r3n_object128 *__restrict__ objects;   // transforms of the slots that are baked
    ObjOwn own;
    uint8_t *__restrict__ vis_flags;
    ObjChainRec *chain;
    uint32_t epoch, rounds;
    r3n_cull_counts *__restrict__ counts;
    r3n_vis_entry *__restrict__ vis_list;
    r3n_sub_counts *__restrict__ sub_counts;
    uint32_t *__restrict__ slot_base;
    uint32_t use_prev;                            // viewport camera with history: vis_flags still holds last frame's bytes
    r3n_baked128 *__restrict__ baked;
    uint32_t *__restrict__ first_entry;
};
template <bool BAKE>
__global__ __launch_bounds__(256) void k_object_pass_chained(ObjChainArgs a) {
    __shared__ uint32_t red[4][6];
    __shared__ uint32_t pre[6];
    __shared__ unsigned long long s_vis[R3N_CHAINED_OBJECT_PASS_MAX_ROUNDS][4], s_need[R3N_CHAINED_OBJECT_PASS_MAX_ROUNDS][4];
    __shared__ uint32_t s_ntri[R3N_CHAINED_OBJECT_PASS_MAX_ROUNDS][256];
    __shared__ uint32_t s_wt[R3N_CHAINED_OBJECT_PASS_MAX_ROUNDS * 4u][2];  // per (round, wave): visible objects, wave slots -> their exclusive prefix
    const uint32_t t = threadIdx.x, wave = t >> 6, lane = t & 63u;
    const uint32_t cap = a.hdr->object_count, rounds = a.rounds;
    const uint32_t first = blockIdx.x * rounds * 256u;
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    if (blockIdx.x == 0u && t < 2u * 3u * R3N_SUBQ) (&a.sub_counts->n[0][0][0])[t] = 0u;  // culler.rs:642 + cull.wgsl:47-61
    // ---- phase 1: count (object_count_body)
    uint32_t vals[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t r0 = 0; r0 < rounds; r0 += 4u) {
        uint32_t meta[4], prev[4];
        float4 sph[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t i = first + (r0 + k) * 256u + t;
            const bool in = r0 + k < rounds && i < cap;
            meta[k] = in ? a.soa.meta[i] : 0u;
            sph[k] = in ? a.soa.spheres[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            prev[k] = (BAKE && in && a.use_prev != 0u) ? (uint32_t)a.vis_flags[i] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            if (r0 + k >= rounds) break;
            const uint32_t i = first + (r0 + k) * 256u + t;
            const uint32_t ntri = meta[k] & R3N_META_NTRI_MASK, key = meta[k] >> R3N_META_KEY_SHIFT;
            const uint32_t bits = i < cap ? object_visible(a.hdr, a.own, i, meta[k], sph[k]) : 0u;
            const uint32_t flag = bits & R3N_VIS_DRAWN;
            if (i < cap) a.vis_flags[i] = (uint8_t)bits;
            const uint32_t nw = flag ? (ntri + 63u) / 64u : 0u, ntri_vis = flag ? ntri : 0u;
            const unsigned long long vb = __ballot(flag != 0u);
            const unsigned long long nb = __ballot(ntri != 0u && ((bits | prev[k]) & R3N_VIS_INSIDE) != 0u);
            const uint32_t wnw = wave_reduce_add(nw);
            s_ntri[r0 + k][t] = ntri;
            if (lane == 0u) {
                s_vis[r0 + k][wave] = vb; s_need[r0 + k][wave] = nb;
                s_wt[(r0 + k) * 4u + wave][0] = (uint32_t)__popcll(vb); s_wt[(r0 + k) * 4u + wave][1] = wnw;
            }
            vals[0] += flag; vals[1] += nw; vals[2] += ntri;
            vals[3] += key == 0u ? ntri_vis : 0u; vals[4] += key == 1u ? ntri_vis : 0u; vals[5] += key >= 2u ? ntri_vis : 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t sum = wave_reduce_add(vals[k]);
        if (lane == 0u) red[wave][k] = sum;
    }
    __syncthreads();
    uint32_t mine = 0;
    if (t < 6u) {
        mine = red[0][t] + red[1][t] + red[2][t] + red[3][t];
        __hip_atomic_store(&a.chain[blockIdx.x].v[t], ((unsigned long long)a.epoch << 32) | (unsigned long long)mine, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- exclusive prefix over the blocks in front (object_scan_body's result for this block)
    uint32_t acc[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t j = t; j < blockIdx.x; j += 256u) {
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            unsigned long long w = __hip_atomic_load(&a.chain[j].v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while ((uint32_t)(w >> 32) != a.epoch) {
                __builtin_amdgcn_s_sleep(1);
                w = __hip_atomic_load(&a.chain[j].v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            acc[k] += (uint32_t)w;
        }
    }
    __syncthreads();  // red is reused
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint32_t sum = wave_reduce_add(acc[k]);
        if (lane == 0u) red[wave][k] = sum;
    }
    __syncthreads();
    if (t < 6u) pre[t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    __syncthreads();
    // the block's (round, wave) totals -> exclusive prefixes, in slot order: one wavefront, lane = round * 4 + wave
    if (wave == 0u) {
        const bool in = lane < rounds * 4u;
        const uint32_t v0 = in ? s_wt[lane][0] : 0u, v1 = in ? s_wt[lane][1] : 0u;
        const uint32_t i0 = wave_inclusive_scan(v0, lane), i1 = wave_inclusive_scan(v1, lane);
        if (in) { s_wt[lane][0] = pre[0] + i0 - v0; s_wt[lane][1] = pre[1] + i1 - v1; }
    }
    __syncthreads();
    // ---- phase 3: scatter (object_scatter_body) + bake; no barrier between the rounds, a wavefront works on its own 64 slots
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t i = first + r * 256u + t;
        const unsigned long long vb = s_vis[r][wave];
        const uint32_t flag = (uint32_t)(vb >> lane) & 1u;
        const uint32_t ntri = s_ntri[r][t];
        const uint32_t nw = flag ? (ntri + 63u) / 64u : 0u;
        const uint32_t e = s_wt[r * 4u + wave][0] + (uint32_t)__popcll(vb & lane_lt);
        const uint32_t ws = s_wt[r * 4u + wave][1] + wave_inclusive_scan(nw, lane) - nw;
        if (i < cap) {
            if (flag) {
                a.vis_list[e].object = i;
                a.vis_list[e].wave_start = ws;
            }
            a.slot_base[i] = flag ? ws * 64u : R3N_INVALID;
        }
        write_first_entries(a.first_entry, flag, e, ws, nw, lane);
        if (BAKE) {
            // the wave's slots that need a matrix, compacted to the front of the wavefront by a forward permute (lanes without
            // one fill the tail, so the permutation is total); then 16 slots per step, thread (slot, column): consecutive
            // threads read consecutive 16-byte columns of a transform and write two 16-byte columns
            const unsigned long long nb = s_need[r][wave];
            const bool need = ((nb >> lane) & 1ull) != 0ull;
            const uint32_t n = (uint32_t)__popcll(nb);
            const uint32_t rank = need ? (uint32_t)__popcll(nb & lane_lt) : n + (uint32_t)__popcll(~nb & lane_lt);
            const uint32_t list = (uint32_t)__builtin_amdgcn_ds_permute((int)(rank << 2), (int)i);
            const uint32_t c = lane & 3u;
            uint32_t obj[4];
            float4 col[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {  // the (up to four) steps' transform columns in flight together
                const uint32_t src = j * 16u + (lane >> 2);
                obj[j] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)list);
                col[j] = src < n ? reinterpret_cast<const float4 *>(a.objects[obj[j]].transform)[c] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                if (j * 16u + (lane >> 2) < n) {
                    float mv[4], mvp[4];
                    mul_vec4(a.hdr->view, col[j].x, col[j].y, col[j].z, col[j].w, mv);
                    mul_vec4(a.hdr->view_proj, col[j].x, col[j].y, col[j].z, col[j].w, mvp);
                    reinterpret_cast<float4 *>(a.baked[obj[j]].model_view)[c] = make_float4(mv[0], mv[1], mv[2], mv[3]);
                    reinterpret_cast<float4 *>(a.baked[obj[j]].model_view_proj)[c] = make_float4(mvp[0], mvp[1], mvp[2], mvp[3]);
                }
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1u && t < 6u) red[0][t] = pre[t] + mine;  // totals: everything in front + this block
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1u && t == 0u) {
        a.counts->visible_objects = red[0][0];
        a.counts->total_waves = red[0][1];
        a.counts->total_triangles = red[0][2];
        uint32_t rb = 0;
        for (int k = 0; k < 3; ++k) {
            a.counts->key_triangles[k] = red[0][3 + k];
            a.counts->region_base[k] = rb;
            rb += red[0][3 + k];
        }
        a.vis_list[red[0][0]].object = R3N_INVALID;  // sentinel
        a.vis_list[red[0][0]].wave_start = red[0][1];
    }
}

// ------------------------------------------------------------------------------------------------ K8 skinning
// skinning.wgsl:37-94.  One launch for all skeletons: wave slot w (64 vertices) belongs to skeleton wave_skeleton[w]
// and covers its vertices [64 * (w - wave_first[skeleton]), +64): the skeleton record and its matrix base are
// wave-uniform, positions / normals / tangents / weights are contiguous per wave (coalesced), outputs likewise.
// HBM-bound: 60 B read + 36 B written per vertex against <= 272 flops -- plain f32 VALU with the
// same operation order as the oracle; MFMA would change the rounding (fma chain) without moving the bound.
__global__ __launch_bounds__(256) void k_skinning(uint32_t *__restrict__ mesh, const r3n_skinning_input40 *__restrict__ inputs,
                                                  const float *__restrict__ joint_matrices,
                                                  const uint32_t *__restrict__ wave_skeleton,
                                                  const uint32_t *__restrict__ wave_first, uint32_t total_waves) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (w >= total_waves) return;
    const uint32_t sk = __builtin_amdgcn_readfirstlane(wave_skeleton[w]);
    const r3n_skinning_input40 in = inputs[sk];
    const uint32_t idx = (w - wave_first[sk]) * 64u + lane;
    if (idx >= in.vertex_count) return;
    const r3n_words2 jj = *reinterpret_cast<const r3n_words2 *>(mesh + in.joint_indices_offset / 4u + idx * 2u);
    const uint32_t j0 = jj.x, j1 = jj.y;
    const uint32_t ji[4] = {j0 & 0xFFFFu, (j0 >> 16) & 0xFFFFu, j1 & 0xFFFFu, (j1 >> 16) & 0xFFFFu};
    const r3n_words4 jww = *reinterpret_cast<const r3n_words4 *>(mesh + in.joint_weight_offset / 4u + idx * 4u);
    const float jw[4] = {__uint_as_float(jww.x), __uint_as_float(jww.y), __uint_as_float(jww.z), __uint_as_float(jww.w)};
    float pos[3] = {0.0f, 0.0f, 0.0f}, nrm[3] = {0.0f, 0.0f, 0.0f}, tan[3] = {0.0f, 0.0f, 0.0f};
    if (in.base_position_offset != R3N_INVALID) fetch_vec3(mesh, in.base_position_offset, idx, pos);
    if (in.base_normal_offset != R3N_INVALID) fetch_vec3(mesh, in.base_normal_offset, idx, nrm);
    if (in.base_tangent_offset != R3N_INVALID) fetch_vec3(mesh, in.base_tangent_offset, idx, tan);
    float pa[3] = {0.0f, 0.0f, 0.0f}, na[3] = {0.0f, 0.0f, 0.0f}, ta[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float wgt = jw[i];
        if (wgt > 0.0f) {
            const float *jm = joint_matrices + 16u * (size_t)(in.joint_matrix_base_offset + ji[i]);
            float m[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) m[k] = jm[k];
            float p4[4];
            mul_vec4(m, pos[0], pos[1], pos[2], 1.0f, p4);
#pragma unroll
            for (int c = 0; c < 3; ++c) pa[c] += p4[c] * wgt;
            const float inv_s2[3] = {1.0f / dot3(m, m), 1.0f / dot3(m + 4, m + 4), 1.0f / dot3(m + 8, m + 8)};
            const float sn[3] = {inv_s2[0] * nrm[0], inv_s2[1] * nrm[1], inv_s2[2] * nrm[2]};
            const float st[3] = {inv_s2[0] * tan[0], inv_s2[1] * tan[1], inv_s2[2] * tan[2]};
            float rn[3], rt[3];
            mat3_mul_vec3(m, m + 4, m + 8, sn, rn);
            mat3_mul_vec3(m, m + 4, m + 8, st, rt);
#pragma unroll
            for (int c = 0; c < 3; ++c) { na[c] += rn[c] * wgt; ta[c] += rt[c] * wgt; }
        }
    }
    normalize3(na);
    normalize3(ta);
    if (in.updated_position_offset != R3N_INVALID) {
        const uint32_t o = in.updated_position_offset / 4u + idx * 3u;
        *reinterpret_cast<r3n_words3 *>(mesh + o) = r3n_words3{__float_as_uint(pa[0]), __float_as_uint(pa[1]), __float_as_uint(pa[2])};
    }
    if (in.updated_normal_offset != R3N_INVALID) {
        const uint32_t o = in.updated_normal_offset / 4u + idx * 3u;
        *reinterpret_cast<r3n_words3 *>(mesh + o) = r3n_words3{__float_as_uint(na[0]), __float_as_uint(na[1]), __float_as_uint(na[2])};
    }
    if (in.updated_tangent_offset != R3N_INVALID) {
        const uint32_t o = in.updated_tangent_offset / 4u + idx * 3u;
        *reinterpret_cast<r3n_words3 *>(mesh + o) = r3n_words3{__float_as_uint(ta[0]), __float_as_uint(ta[1]), __float_as_uint(ta[2])};
    }
}

// slot_table[b] = last object o with tri_base[o] <= (b << R3N_SLOT_TABLE_SHIFT): accelerates the slot -> object
// lookup of the resolve pass.  Rebuilt only when the object set changes.
__global__ __launch_bounds__(256) void k_build_slot_table(const uint32_t *__restrict__ tri_base, uint32_t capacity,
                                                          uint32_t *__restrict__ table, uint32_t table_size) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= table_size) return;
    const uint32_t slot = b << R3N_SLOT_TABLE_SHIFT;
    uint32_t lo = 0, hi = capacity;
    while (hi - lo > 1u) {
        const uint32_t mid = lo + (hi - lo) / 2u;
        if (tri_base[mid] <= slot) lo = mid; else hi = mid;
    }
    table[b] = lo;
}

// ------------------------------------------------------------------------------------------------ K2
struct HizView {
    const float *__restrict__ data;
    r3n_hiz_desc d;
};

// cull.wgsl:243-262 (textureSampleMin) on the linear f32 pyramid
R3N_DEV float hiz_sample_min(const HizView &hz, float u, float v, uint32_t mip) {
    const uint32_t mw = mip_dim(hz.d.width, mip), mh = mip_dim(hz.d.height, mip);
    const float *tex = hz.data + hz.d.offset[mip];
    const float px = u * (float)mw - 0.5f;
    const float py = v * (float)mh - 0.5f;
    const uint32_t lx = clamp_texel(fmaxf(floorf(px), 0.0f), mw);
    const uint32_t ly = clamp_texel(fmaxf(floorf(py), 0.0f), mh);
    const uint32_t hx = clamp_texel(fminf(ceilf(px), (float)mw - 1.0f), mw);
    const uint32_t hy = clamp_texel(fminf(ceilf(py), (float)mh - 1.0f), mh);
    float m = tex[ly * mw + lx];
    m = fminf(m, tex[ly * mw + hx]);
    m = fminf(m, tex[hy * mw + lx]);
    m = fminf(m, tex[hy * mw + hx]);
    return m;
}

// cull.wgsl:264-324, from the clip-space positions p[k] = model_view_proj * vertex k
R3N_DEV bool execute_culling_clip(const float p[3][4], uint32_t flags, bool shadow, float res_x, float res_y, const HizView &hz) {
    const float det = det3_xyw(p[0], p[1], p[2]);
    if (flags & R3N_PCU_POSITIVE_AREA_VISIBLE) {
        if (det <= 0.0f) return false;
    } else {
        if (det >= 0.0f) return false;
    }
    float ndc[3][3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) ndc[k][c] = p[k][c] / p[k][3];
    float mn[2], mx[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        mn[c] = fminf(ndc[0][c], fminf(ndc[1][c], ndc[2][c]));
        mx[c] = fmaxf(ndc[0][c], fmaxf(ndc[1][c], ndc[2][c]));
    }
    const float half_res[2] = {res_x / 2.0f, res_y / 2.0f};
    float smin[2], smax[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        smin[c] = (mn[c] + 1.0f) * half_res[c];
        smax[c] = (mx[c] + 1.0f) * half_res[c];
    }
    if (!(flags & R3N_PCU_MULTISAMPLED)) {
        // WGSL round() is ties-to-even == v_rndne_f32
        if (rintf(smin[0]) == rintf(smax[0]) || rintf(smin[1]) == rintf(smax[1])) return false;
    }
    if (shadow) return true;  // cull.wgsl:300-303

    float mintc[2] = {(mn[0] + 1.0f) / 2.0f, (mn[1] + 1.0f) / 2.0f};
    float maxtc[2] = {(mx[0] + 1.0f) / 2.0f, (mx[1] + 1.0f) / 2.0f};
    mintc[1] = 1.0f - mintc[1];
    maxtc[1] = 1.0f - maxtc[1];
    const float uv[2] = {(maxtc[0] + mintc[0]) / 2.0f, (maxtc[1] + mintc[1]) / 2.0f};
    const float longest = fmaxf(smax[0] - smin[0], smax[1] - smin[1]);
    uint32_t mip = ceil_log2_ge1(longest);
    if (mip > hz.d.mips - 1u) mip = hz.d.mips - 1u;  // App. D.1: clamp to the last level
    const float depth = fmaxf(fmaxf(ndc[0][2], ndc[1][2]), ndc[2][2]);
    const float occ = hiz_sample_min(hz, uv[0], uv[1], mip);
    if (depth < occ) return false;
    return true;
}
R3N_DEV bool execute_culling(const float *__restrict__ mvp, const float v[3][3], uint32_t flags, bool shadow,
                             float res_x, float res_y, const HizView &hz) {
    float p[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) mul_point(mvp, v[k], p[k]);
    return execute_culling_clip(p, flags, shadow, res_x, res_y, hz);
}

// Wave-uniform data through the SCALAR unit.  The address is made provably uniform (readfirstlane) and named in the constant
// address space, which is what lets the compiler emit s_load; only for memory that no kernel of the same launch writes (the
// scalar cache is not coherent with vector stores inside a launch; across launches it is invalidated).  A vector load of a
// uniform value costs a full vector-memory round trip per wave AND blocks on vmcnt behind whatever else is in flight.
#define R3N_CULL_PAIR 1  // the triangle cull fetches two wave slots of an object together
typedef uint32_t r3n_u32x2 __attribute__((ext_vector_type(2)));

#define R3N_CHUNK_WAVES (4u * R3N_CHUNK_ITERS)   // wave slots per 256-thread block per chunk (4096 triangles)

struct TriCullArgs {
    const r3n_camera_header240 *hdr;
    const r3n_object128 *objects;
    const uint32_t *mesh;
    const r3n_baked128 *baked;
    const uint8_t *material_keys;
    uint32_t n_materials;
    const r3n_vis_entry *vis_list;
    const uint32_t *first_entry;          // write_first_entries: the entry owning wave slot k * R3N_CHUNK_ITERS
    const r3n_cull_counts *counts;
    const uint32_t *prev_slot_base;       // per object, INVALID when absent last frame; may be null
    const unsigned long long *prev_mask;  // may be null
    unsigned long long *mask;             // one u64 per wave slot
    r3n_tri_ref *predicted;
    r3n_tri_ref *residual;                // null for shadow cameras
    r3n_sub_counts *sub_counts;           // append counters of this cull
    uint32_t subcap;                      // entries reserved per (material key, sub-list) in each list
    HizView hiz;
};

// Persistent kernel: block b processes chunks b, b+grid, ... ; within a chunk wavefront w owns
// R3N_CHUNK_ITERS consecutive wave slots.  Per-iteration ballots are staged in LDS, then one thread per
// (list, region) reserves output space for the whole 4096-slot chunk with a single global atomic.
__global__ __launch_bounds__(256) void k_triangle_cull(TriCullArgs a) {
    __shared__ unsigned long long s_pass[4][R3N_CHUNK_ITERS];
    __shared__ unsigned long long s_resid[4][R3N_CHUNK_ITERS];
    __shared__ uint32_t s_obj[4][R3N_CHUNK_ITERS];
    __shared__ uint32_t s_tri0[4][R3N_CHUNK_ITERS];
    __shared__ uint32_t s_key[4][R3N_CHUNK_ITERS];
    __shared__ uint32_t cnt[4][6];    // per wave: [list*3 + key] passing triangles in this chunk
    __shared__ uint32_t base[4][6];   // per wave: first output entry for (list,key)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t total_waves = a.counts->total_waves;
    const uint32_t nchunks = (total_waves + R3N_CHUNK_WAVES - 1u) / R3N_CHUNK_WAVES;
    const uint32_t flags = a.hdr->flags;
    const bool shadow = a.hdr->shadow_index != R3N_INVALID;
    const float res_x = a.hdr->resolution[0], res_y = a.hdr->resolution[1];
    const unsigned long long lane_lt = (1ull << lane) - 1ull;

    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint32_t w0 = chunk * R3N_CHUNK_WAVES + wave * R3N_CHUNK_ITERS;
        // the entry owning wave slot w0 (the last e with wave_start[e] <= w0): one scalar load from the table the object pass
        // wrote (write_first_entries) instead of the reference's binary search (cull.wgsl:181-207)
        uint32_t e = 0, next_start = 0;
        if (w0 < total_waves) {
            e = scalar_load<uint32_t>(&a.first_entry[w0 / R3N_CHUNK_ITERS]);
            next_start = scalar_load<uint32_t>(&a.vis_list[e + 1u].wave_start);
        }
        uint32_t c_p0 = 0, c_p1 = 0, c_p2 = 0, c_r0 = 0, c_r1 = 0, c_r2 = 0;
        uint32_t e_held = R3N_INVALID, pb = R3N_INVALID, keyw = 0u;  // the object whose data the scalar registers hold
        r3n_u32x2 ent = {};
        r3n_u32x4 of = {};
        r3n_u32x16 mw = {};
        // a wave slot's result: residual bits against last frame's (cull.wgsl:152-160), the result bits (cull.wgsl:229-240), the
        // per-key counts and the LDS staging of the compaction
        auto finish = [&](uint32_t w, uint32_t it, uint32_t obj, uint32_t wrel, uint32_t key, unsigned long long ballot) {
            unsigned long long resid = 0;
            if (!shadow) {
                unsigned long long prev = 0;
                if (pb != R3N_INVALID) prev = scalar_load<unsigned long long>(&a.prev_mask[pb / 64u + wrel]);
                resid = ballot & ~prev;
            }
            if (lane == 0) a.mask[w] = ballot;
            const uint32_t np = (uint32_t)__popcll(ballot), nr = (uint32_t)__popcll(resid);
            c_p0 += key == 0u ? np : 0u; c_p1 += key == 1u ? np : 0u; c_p2 += key == 2u ? np : 0u;
            c_r0 += key == 0u ? nr : 0u; c_r1 += key == 1u ? nr : 0u; c_r2 += key == 2u ? nr : 0u;
            if (lane == 0) {
                s_pass[wave][it] = ballot; s_resid[wave][it] = resid;
                s_obj[wave][it] = obj; s_tri0[wave][it] = wrel * 64u; s_key[wave][it] = key;
            }
        };

#pragma unroll 1
        for (uint32_t it = 0; it < R3N_CHUNK_ITERS;) {
            const uint32_t w = w0 + it;
            if (w >= total_waves) {
                if (lane == 0) { s_pass[wave][it] = 0ull; s_resid[wave][it] = 0ull; s_obj[wave][it] = 0u; s_tri0[wave][it] = 0u; s_key[wave][it] = 0u; }
                ++it;
                continue;
            }
            // Everything about the wave slot that is wave-uniform -- the list entry, four fields of the object record, the
            // baked matrix, the material key, last frame's result bits -- comes through scalar loads: two scalar round trips
            // (entry; then record + matrix + previous bits together) in front of the two vector ones (indices; positions)
            // instead of seven dependent vector round trips, and the matrix lives in scalar registers.
            while (w >= next_start) { ++e; next_start = scalar_load<uint32_t>(&a.vis_list[e + 1u].wave_start); }
            if (e != e_held) {  // consecutive wave slots mostly stay inside one object: its data is read once per object, not per slot
                e_held = e;
                ent = scalar_load<r3n_u32x2>(&a.vis_list[e]);
                // first_index, index_count, material_index, vertex_attribute_start_offsets[0]: bytes 80..95 of the record
                of = scalar_load<r3n_u32x4>(reinterpret_cast<const char *>(&a.objects[ent.x]) + offsetof(r3n_object128, first_index));
                mw = scalar_load<r3n_u32x16>(a.baked[ent.x].model_view_proj);
                pb = R3N_INVALID;
                if (!shadow && a.prev_slot_base != nullptr) pb = scalar_load<uint32_t>(&a.prev_slot_base[ent.x]);
                keyw = 0u;
                if (of.z < a.n_materials) keyw = scalar_load<uint32_t>(a.material_keys + (of.z & ~3u)) >> ((of.z & 3u) * 8u);
            }
            const uint32_t obj = ent.x, wrel = w - ent.y;
            float mvp[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) mvp[k] = __uint_as_float(mw[k]);
            const uint32_t ntri = of.y / 3u;  // >= 1: the list holds objects with triangles
            uint32_t key = keyw & 0xFFu;
            key = key > 2u ? 2u : key;
            const uint32_t triA = wrel * 64u + lane;
            // The next wave slot belongs to the same object (15 of 16 do on the bench scene): its index and position fetches go out
            // together with this slot's -- two slots per pair of vector round trips.  Out-of-range lanes fetch the object's last
            // triangle instead of branching, so the two slots' loads sit in one block of straight-line code.
            const bool pair = R3N_CULL_PAIR && it + 1u < R3N_CHUNK_ITERS && w + 1u < next_start;
            if (pair) {
                const uint32_t triB = triA + 64u;
                uint32_t idxA[3], idxB[3];
                fetch_indices3(a.mesh, of.x + min(triA, ntri - 1u) * 3u, idxA);
                fetch_indices3(a.mesh, of.x + min(triB, ntri - 1u) * 3u, idxB);
                float vA[3][3], vB[3][3];
#pragma unroll
                for (int k = 0; k < 3; ++k) fetch_vec3(a.mesh, of.w, idxA[k], vA[k]);
#pragma unroll
                for (int k = 0; k < 3; ++k) fetch_vec3(a.mesh, of.w, idxB[k], vB[k]);
                const bool passA = execute_culling(mvp, vA, flags, shadow, res_x, res_y, a.hiz) && triA < ntri;
                const bool passB = execute_culling(mvp, vB, flags, shadow, res_x, res_y, a.hiz) && triB < ntri;
                const unsigned long long bA = __ballot(passA), bB = __ballot(passB);
                finish(w, it, obj, wrel, key, bA);
                finish(w + 1u, it + 1u, obj, wrel + 1u, key, bB);
                it += 2u;
            } else {
                bool pass = false;
                if (triA < ntri) {
                    float v[3][3];
                    uint32_t idx[3];
                    fetch_indices3(a.mesh, of.x + triA * 3u, idx);
#pragma unroll
                    for (int k = 0; k < 3; ++k) fetch_vec3(a.mesh, of.w, idx[k], v[k]);
                    pass = execute_culling(mvp, v, flags, shadow, res_x, res_y, a.hiz);
                }
                finish(w, it, obj, wrel, key, __ballot(pass));
                ++it;
            }
        }

        // ---- LDS-staged compaction: one global atomic per (chunk, list, region)
        if (lane == 0) {
            cnt[wave][0] = c_p0; cnt[wave][1] = c_p1; cnt[wave][2] = c_p2;
            cnt[wave][3] = c_r0; cnt[wave][4] = c_r1; cnt[wave][5] = c_r2;
        }
        __syncthreads();
        if (threadIdx.x < 6u) {
            const uint32_t k = threadIdx.x;
            const uint32_t c0 = cnt[0][k], c1 = cnt[1][k], c2 = cnt[2][k], c3 = cnt[3][k];
            const uint32_t tot = c0 + c1 + c2 + c3;
            uint32_t start = 0;
            // cull.wgsl:63-73 (atomicAdd on the region's vertex_count), one add per chunk and per sub-list
            if (tot) start = atomicAdd(&(&a.sub_counts->n[0][0][0])[k * R3N_SUBQ + (chunk % R3N_SUBQ)], tot);
            base[0][k] = start; base[1][k] = start + c0; base[2][k] = start + c0 + c1; base[3][k] = start + c0 + c1 + c2;
        }
        __syncthreads();
        uint32_t run_p[3] = {base[wave][0], base[wave][1], base[wave][2]};
        uint32_t run_r[3] = {base[wave][3], base[wave][4], base[wave][5]};
#pragma unroll 1
        for (uint32_t it = 0; it < R3N_CHUNK_ITERS; ++it) {
            const unsigned long long pb = s_pass[wave][it];
            if (pb == 0ull) continue;
            const unsigned long long rb = s_resid[wave][it];
            const uint32_t key = s_key[wave][it];
            const r3n_tri_ref ref = {s_obj[wave][it], s_tri0[wave][it] + lane};
            const uint32_t np = (uint32_t)__popcll(pb), nr = (uint32_t)__popcll(rb);
            const uint32_t rp = key == 0u ? run_p[0] : (key == 1u ? run_p[1] : run_p[2]);
            const uint32_t rr = key == 0u ? run_r[0] : (key == 1u ? run_r[1] : run_r[2]);
            const uint32_t region = (key * R3N_SUBQ + (chunk % R3N_SUBQ)) * a.subcap;
            if ((pb >> lane) & 1ull) a.predicted[region + rp + (uint32_t)__popcll(pb & lane_lt)] = ref;
            if ((rb >> lane) & 1ull) a.residual[region + rr + (uint32_t)__popcll(rb & lane_lt)] = ref;
#pragma unroll
            for (uint32_t k = 0; k < 3u; ++k) {
                run_p[k] += key == k ? np : 0u;
                run_r[k] += key == k ? nr : 0u;
            }
        }
        __syncthreads();  // LDS staging reused by the next chunk
    }
}
