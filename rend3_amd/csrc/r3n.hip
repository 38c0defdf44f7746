// r3n.hip -- context, device-memory management and the extern "C" entry points of include/r3n.h.
// All device work is enqueued on the context's stream; nothing on the frame path reads back to the host.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "comm.h"
#include "kernels_cull.h"
#include "kernels_raster.h"
#define R3N_SHADE_DECL_ONLY
#include "kernels_shade.h"

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct CamState {
    r3n_camera_header240 hdr{};
    bool has_hdr = false;
    bool has_prev = false;       // results of a previous frame's cull exist in index 1-cur
    bool culled = false;         // culled during the current frame
    int cur = 0;                 // ping-pong index written by this frame's cull
    int last = -1;               // index holding the most recent cull results (for readbacks)
    uint32_t vp_x = 0, vp_y = 0, vp_size = 0;
    DevBuf d_hdr;                // NOT owned: points into the frame-constants block of the current frame slot (canon: into own_hdr)
    DevBuf own_hdr;
    uint64_t hdr_frame = ~0ull;  // frame whose constants block already carries this camera's header (r3n_render_frame)
    uint64_t baked_frame = ~0ull;  // frame of the last r3n_uniform_bake: the header in THIS frame's constants block is only valid then
    DevBuf chain;                // k_object_pass_chained: one ObjChainRec per block, tagged with the launch's epoch
    uint32_t chain_epoch = 0;
    uint64_t object_pass_frame = ~0ull;  // frame whose object pass already ran, fused with the bake (r3n_render_frame)
    DevBuf baked, vis_flags, vis_list, block_sums, block_off;
    DevBuf first_entry;          // kernels_cull.h write_first_entries: the work-list entry owning every R3N_CHUNK_ITERS-th wave slot
    DevBuf slot_base[2], mask[2], predicted[2], sub_counts[2], counts[2];
    uint32_t subcap[2] = {0, 0};  // list entries reserved per (material key, sub-list)
    uint64_t key_objects[2][3] = {};  // enabled objects per material key AS THE CULL OF THAT INDEX SAW THEM: last frame's predicted
                                      // triangles sit in last frame's draw ranges (forward.rs:224-232,286), whatever the keys are now
    DevBuf residual;
    bool range_set = false;      // r3n_set_camera_object_range: this camera's own object range (multi-GPU: shadow views owned whole)
    uint32_t range_begin = 0, range_end = 0xFFFFFFFFu;
    uint32_t band_begin = 0, band_end = 0xFFFFFFFFu;  // shadow view split by rows over several ranks (r3n_render_frame, native exchange): the rows this rank rasterises
};

std::string g_create_error;

}  // namespace

#define R3N_AUX_STREAMS 2  // main + shade + these = the four hardware queues the runtime uses by default; 3 / 4 streams measured no faster (profiles/r02_summary.md section 9)

// Dynamic LDS bytes asked by the work-item rasteriser's launches.  The kernel uses no LDS: the allocation caps its workgroups per CU.
// Shadow views at 32 KB (4 workgroups = 4 waves per SIMD instead of 8): the four launches take 444 instead of 466 us and the frame
// 1.146 instead of 1.169 ms (two runs each) -- at full occupancy its fire-and-forget atomics queue up behind each other and its
// idle waves hold slots the resolve of the previous frame could use.  54 KB (2 workgroups): 622 us.  The same cap on the viewport's
// launches, on the per-triangle pass and on the triangle cull was measured: no gain alone, and any two caps together lose (kernels
// that each reserve a third of a CU's LDS no longer co-reside).
// Round 4, with the resolve at four waves per SIMD and the slimmer per-triangle pass beside them: the cap pays on the viewport's
// launches too, and three workgroups per CU beat four -- frame (same box, two runs each): shadow 32 KB / viewport none 0.982-0.993 ms;
// 32 / 32 0.968-0.972; 48 / 32 0.961 (0.935 on another box against 0.968); 48 / 48 0.940; 64 / 48 (two workgroups) 1.061; no cap 1.049.
// (constants fixed by measurement, not build options)
#define R3N_BIG_GRID 8192  // 4x the resident wave count (8 waves per SIMD at < 64 VGPRs): the hardware dispatcher
                           // then balances the uneven item costs (measured on the bench scene, shadow views:
                           // 2048 -> 419 us, 4096 -> 387 us, 8192 -> 366 us per frame)
// Workgroups of the two rasteriser kernels: 256 threads (neither uses LDS or barriers: a workgroup is only the unit in which wave
// slots are handed back to the dispatcher; 64- and 128-thread workgroups measured the same, profiles/r04_summary.md section 6a).
#define R3N_SMALL_GRID 2048
#define R3N_BIG_LDS 40960           // shadow views (tools/tune_caps.py, round 5: 40960 -- 49152 on the bench scene, 32768 on the million-object one)
#define R3N_SMALL_LDS 24576         // the shadow views' per-triangle pass: six workgroups per CU (bench scene 0.9285 -> 0.9148 ms, round 5)
#define R3N_VIEWPORT_BIG_LDS 49152  // viewport
#define R3N_QLANES R3N_AUX_STREAMS  // work queues beside the viewport's: one per auxiliary stream (a shadow view draws on lane 1 + view mod R3N_AUX_STREAMS)
static_assert(R3N_AUX_STREAMS >= 1, "the shadow views draw on auxiliary streams");

struct r3n_ctx {
    int device = 0;
    hipStream_t stream = nullptr;  // main stream: uploads, viewport chain, resolve, tonemap, collectives
    // Shadow views are independent of each other and of the viewport's pass-1 / Hi-Z / cull chain until the
    // resolve reads the atlas, and their kernels are latency-bound rather than throughput-bound: each shadow camera
    // runs on its own auxiliary stream (forked after the frame's clears, joined before the resolve) so the five
    // chains overlap and the launch gaps of one hide behind the work of the others.
    hipStream_t aux[R3N_AUX_STREAMS] = {};
    hipEvent_t fork_ev[R3N_AUX_STREAMS] = {};
    // Frames in flight.  The resolve is VALU-bound and everything before it (culls, rasterisers, Hi-Z) is latency- and
    // atomic-bound, so frame N's resolve runs on its own stream while the main stream and the lanes already work on frame
    // N + 1.  Everything the resolve reads that the next frame rewrites exists twice; the two sets swap at frame_begin:
    //   vis, atlas, viewport.baked  <->  alt_*;  fu, dir_buf, point_buf, every camera's d_hdr: the frame-constants block of the slot
    // shade_done[slot] (recorded on the shade stream after the resolve of the frame that used `slot`) is what the main
    // stream waits for before it clears that slot's targets two frames later.  R3N_PIPELINE=0 disables the overlap.
    hipStream_t shade = nullptr;
    hipEvent_t vp_ev = nullptr, shade_done[2] = {nullptr, nullptr};
    bool shade_pending[2] = {false, false};
    bool shade_unjoined = false;      // the main stream has not yet been ordered after the latest resolve
    int shade_last = 0;               // the slot of that resolve
    int slot = 0;
    uint64_t frame_no = 0;
    bool overlap = true;
    bool always_fork = false;     // R3N_ALWAYS_FORK=1: the shadow lanes wait for the main stream at every fork (no epoch gate)
    bool resolve_classes = true;  // R3N_RESOLVE_CLASSES=0: the general resolve kernel for every tile (A/B, tests)
    bool fused_frame = false;  // inside r3n_render_frame: a camera's bake and object pass are ONE launch, issued at its r3n_uniform_bake
    DevBuf alt_vis, alt_atlas, alt_vp_baked;
    // Frame-constants block: FrameUniforms, the viewport's camera header, the directional-light buffer, every shadow view's camera
    // header and the point-light buffer of one frame live in ONE device block per frame slot (fu / dir_buf / point_buf / d_hdr
    // point into it), filled through one pinned host image per frame in flight: r3n_render_frame uploads a frame's constants
    // with ONE copy; the per-node path uploads uniforms + lights at r3n_frame_begin and each header at its r3n_uniform_bake.
    static constexpr size_t kFbUniforms = 0, kFbViewportHdr = 512, kFbDir = 768, kFbShadowHdr = 3072,
                            kFbPoint = kFbShadowHdr + 256 * R3N_MAX_SHADOW_VIEWS, kFbBytes = kFbPoint + 8448;
    static constexpr int kFbHost = 4;
    DevBuf fb_dev[2];
    uint8_t *fb_host[kFbHost] = {};
    hipEvent_t fb_ev[kFbHost] = {};
    bool fb_pending[kFbHost] = {};
    // fork_lane is a no-op while nothing the lanes depend on has been enqueued on the main stream since their last fork
    uint64_t main_epoch = 1, lane_epoch[R3N_AUX_STREAMS] = {};
    DevBuf big_count_all;  // the work-queue counters of every lane: zeroed once per frame
    std::vector<uint8_t> h_dir, h_point;   // the light buffers as last written (uploaded into the frame's slot at frame_begin)
    uint64_t lights_version = 1, slot_lights_version[2] = {0, 0};
    hipEvent_t join_ev[R3N_AUX_STREAMS] = {};
    bool aux_used[R3N_AUX_STREAMS] = {};
    bool multi_stream = true;
    std::string err;
    // world data
    DevBuf mesh, objects, materials, material_keys, dir_buf, point_buf, fu;
    // structure-of-arrays view of the objects for the object pass (kernels_cull.h ObjSoA): bounding spheres, and per slot
    // (enabled ? index_count / 3 : 0) | material key << 30 -- written beside the 128-byte records by r3n_objects_write, the key
    // bits refreshed from the host mirrors when r3n_materials_write changed a key (refresh_obj_meta)
    DevBuf spheres, obj_meta;
    bool meta_dirty = false;
    uint32_t capacity = 0, n_materials = 0;
    std::vector<uint32_t> h_ntri;  // host mirror: triangles per enabled object slot
    std::vector<uint32_t> h_material;  // host mirror: material index per object slot
    std::vector<uint8_t> h_material_key;  // host mirror: Material::key() per material slot
    // material classes of the single-sample resolve (kernels_shade.h R3N_FEAT_*): host mirror of the records, per-texture
    // "the sampler's short path applies", the feature word per material on the device and the variants the census found
    std::vector<r3n_material208> h_materials;
    std::vector<uint8_t> h_tex_short;
    DevBuf material_feat;
    DevBuf view_lights[2];  // ViewLights of the frame set (k_stage_view_lights writes it in front of the single-sample resolve)
    uint32_t resolve_variants = 0;
    bool classes_dirty = true;
    // Launch parameters that decide how the frame's kernels SHARE the chip (profiles/r04_summary.md section 6a: the frame is the sum of
    // latency-bound kernels on five streams; leaving room for the neighbours moved it more than any kernel's own speed): dynamic LDS
    // nobody uses = a cap on a kernel's resident workgroups per CU, and the grids of the persistent rasterisers.  The defaults are the
    // result of tools/tune_caps.py's coordinate search on the bench scene; R3N_TUNE="key=value ..." overrides them at r3n_create.
    struct Tune {
        uint32_t big_lds = R3N_BIG_LDS, vp_big_lds = R3N_VIEWPORT_BIG_LDS;  // work-item rasteriser, opaque key: shadow views / viewport
        uint32_t small_lds = R3N_SMALL_LDS, vp_small_lds = 0;               // per-triangle pass, opaque key
        uint32_t cut_big_lds = 0, vp_cut_big_lds = 0, cut_small_lds = 0, vp_cut_small_lds = 0;  // the same for the CUTOUT key's launches (their
                                                                            // alpha test makes them long kernels that want the whole chip: a cap
                                                                            // of 48 KB cost the Bistro-like scene's frame a quarter)
        uint32_t prio_main = 1, prio_shade = 1, prio_aux = 1;  // stream priorities (0 high, 1 normal, 2 low), applied when the streams are created (R3N_TUNE)
        uint32_t timed_pipeline = 0;  // tools/frame_timeline.py: keep the frames in flight while the timing taps are on (the stage figures then overlap)
        uint32_t cull_lds = 0, vp_cull_lds = 0;                             // triangle cull
        uint32_t resolve_lds = 0;                                           // single-sample resolve
        uint32_t big_grid = R3N_BIG_GRID, small_grid = R3N_SMALL_GRID;
        uint32_t comm_serial = 1;  // r3n_comm_init: the three communicators' collectives in ONE total order per device (comm_order_*)
    } tune;
    bool cutout_short_dirty = true, cutout_short = false;  // cutout_alpha_short(): census of the cutout materials' albedo maps
    bool key_census_dirty = true;
    uint64_t key_objects[3] = {0, 0, 0};  // enabled objects per material key
    uint64_t total_tris = 0;
    bool tri_base_dirty = true;
    DevBuf tri_base, slot_table;
    // skinning (row S1): cached skeleton records + the wave -> skeleton map derived from them
    DevBuf skin_inputs, skin_matrices, skin_wave_skeleton, skin_wave_first, skin_joint_counts;
    uint32_t skinning_mode = R3N_SKIN_EXACT, skin_max_joints = 0;
    std::vector<r3n_skinning_input40> h_skin_inputs;
    uint32_t skin_total_waves = 0;
    uint32_t slot_table_size = 0;
    CamState canon;  // scratch camera used to (re)build the canonical tri_base scan
    // frame targets
    uint32_t width = 0, height = 0, samples = 1, atlas_w = 0, atlas_h = 0;
    float clear[4] = {0, 0, 0, 0};
    bool in_frame = false;
    bool blended_this_frame = false;
    bool resolved_this_frame = false;  // the resolve also wrote the tonemapped image
    DevBuf vis, hdr16, out8, out_f32, atlas, hiz;
    r3n_hiz_desc hizd{};
    bool hiz_plane_ready = false;  // mip 0 already holds the (merged) pass-1 depth: r3n_exchange_depth
    // transparent pass (row N3)
    DevBuf tri_rec, tri_seen;  // per-triangle vertex-stage records of the resolve (kernels_raster.h TriRecord)
    DevBuf blend_order, blend_rank_base, frag_keys, frag_vals, frag_count, frag_head, samples16;
    uint32_t *status_host = nullptr, *status_dev = nullptr;  // host-mapped word the kernels raise when a fixed-size buffer ran out
    std::vector<uint32_t> h_blend_order;
    uint32_t n_blend = 0, blend_tris = 0;
    uint32_t frag_capacity = 32u << 20;  // fragment nodes (12 B each), allocated on first use
    DevBuf tex_descs, tex_texels, tex_level_off, srgb8_decode;  // bindless texture array (row N2): descriptors, RGBA8 texel pool, decode tables  // bindless texture array (row N2) + sRGB8 -> linear table
    uint32_t n_textures = 0;
    uint64_t n_texels = 0;
    // rend3-anim tables (row N4) and the pose requests queued for the next r3n_skinning
    DevBuf edge_list, edge_count;  // split MSAA resolve (kernels_raster.h k_resolve_edges)
    uint32_t edge_capacity_override = 0;
    DevBuf anim_rigs, anim_joints, anim_clips, anim_tracks, anim_times, anim_values, pose_requests;
    std::vector<r3n_anim_rig16> h_anim_rigs;
    std::vector<r3n_anim_clip16> h_anim_clips;
    uint32_t n_pose_requests = 0, pose_matrix_end = 0, anim_max_joints = 1;
    DevBuf big_uv[1 + R3N_QLANES];
    DevBuf srgb_thr;  // 255 floats: smallest linear value whose Rgba8UnormSrgb code is >= c (GPU mip generation)
    DevBuf srgb_lut;  // 8-bit output code of every half in [0, 1): kernels_raster.h k_build_srgb_lut (or r3n_set_output_format)
    uint32_t output_format = R3N_OUTPUT_RGBA8_UNORM_SRGB;
    uint32_t shade_mode = R3N_SHADE_EXACT;
    DevBuf big_items[1 + R3N_QLANES], big_count[1 + R3N_QLANES];  // work queues: [0] the viewport's, 1.. one per concurrently drawn shadow view
    uint32_t forward_index_lane[1 + R3N_QLANES] = {};
    uint32_t big_capacity = (2u << 20) / R3N_BIGQ;  // entries (64 B) per work sub-queue (R3N_BIGQ of them)
    CamState viewport;
    std::map<uint32_t, CamState> shadows;
    uint32_t range_begin = 0, range_end = 0xFFFFFFFFu;
    uint32_t row_begin = 0, row_end = 0xFFFFFFFFu;
    bool shard_rows = false;  // R3N_SHARD_ROWS: the viewport camera rasterises its row band only
    // r3n_comm_init: the sort-first exchanges issued from r3n_render_frame over RCCL
    struct Comm {
        bool on = false;
        bool by_objects = false;  // r3n_comm_set_split(R3N_SHARD_OBJECTS): object-range split (depth MAX all-reduce + key MAX reduce-scatter)
        uint32_t rank = 0, world = 1;
        ncclComm_t main = nullptr, shadow = nullptr, rows = nullptr;
        DevBuf stage[R3N_MAX_SHADOW_VIEWS];  // contiguous copies of the shadow rectangles (what a broadcast moves)
        hipEvent_t order_ev = nullptr;       // tune.comm_serial: behind the last collective enqueued, on whichever stream that was
        bool order_pending = false;
    } comm;
    DevBuf owners;  // r3n_set_object_owners: owner rank per object slot (p == nullptr: slot ranges)
    uint32_t owner_rank = 0, owners_n = 0;
    // pinned staging ring for small per-frame uploads (headers, uniforms, light buffers): the caller owns its
    // pointers only for the duration of a call, and the frame path must not synchronise with the GPU.
    static constexpr uint32_t kStageSlots = 256, kStageSlotBytes = 4096;
    uint8_t *stage = nullptr;
    uint32_t stage_next = 0;
    hipEvent_t stage_half_done[2] = {nullptr, nullptr};
    bool stage_half_pending[2] = {false, false};
    // pinned staging for LARGE per-frame uploads (joint matrices, pose requests): the caller owns its pointer only for the duration
    // of the call and the frame path must not wait for the GPU; a buffer is reused once the copy issued from it has executed
    struct Bulk { uint8_t *p = nullptr; size_t bytes = 0; hipEvent_t ev = nullptr; bool pending = false; } bulk[4];
    uint32_t bulk_next = 0;
    // R3N_BREADCRUMBS=<prefix>: one line per stage / collective into <prefix>.pid<pid>.ctx<n> as it is ENQUEUED (write(2), no
    // buffering) -- where a rank that stopped answering was last seen (tests/mp_harness.py, tools/soak_native.py); -1: off
    int crumb_fd = -1;
    bool sync_stages = false;  // R3N_SYNC_STAGES=1: every stage is waited for where it is enqueued (diagnosis of a kernel that does not return)
    // timing taps
    bool timing = false;
    struct Span { hipEvent_t a, b; int stage; hipStream_t stream; };
    std::vector<Span> spans;
    double span_overhead_ms = 0.0;  // two events around an empty launch (median of 32), measured when timing is switched on
    bool span_calibrated = false;
    std::vector<hipEvent_t> event_pool;
    double stage_ms[R3N_STAGE_COUNT] = {0};
    int hbm_best_variant = -1;
    uint64_t stage_launches[R3N_STAGE_COUNT] = {0};
};

namespace {

int fail(r3n_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}

static void crumb(const r3n_ctx *c, const char *what, long long a = -1, long long b = -1);
// a host wait on the device (where a rank that hangs is found): breadcrumb with the line in front and behind
#define HIP_WAIT(c, expr)                                    \
    do {                                                     \
        crumb((c), "wait " #expr, __LINE__);                 \
        HIP_TRY((c), expr);                                  \
        crumb((c), "wait done", __LINE__);                   \
    } while (0)
#define HIP_TRY(c, expr)                                                                          \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return fail((c), R3N_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    } while (0)

int sync_all(r3n_ctx *c);
int join_shade(r3n_ctx *c);
int join_lanes(r3n_ctx *c);

// Grow-only device buffer.  preserve: keep old contents; fill: byte value for the newly allocated tail.
int ensure(r3n_ctx *c, DevBuf &b, size_t bytes, bool preserve, int fill) {
    if (bytes <= b.bytes && b.p) return R3N_OK;
    size_t want = std::max<size_t>(bytes, 256);
    ++c->main_epoch;  // copies / fills below run on the main stream
    if (b.bytes) want = std::max(want, b.bytes + b.bytes / 2);  // amortised growth
    void *np = nullptr;
    HIP_TRY(c, hipMalloc(&np, want));
    size_t kept = 0;
    if (preserve && b.p && b.bytes) {
        // a lane or the shade stream may still be writing the old allocation: the snapshot must come after them
        int r = join_lanes(c);
        if (r == R3N_OK) r = join_shade(c);
        if (r != R3N_OK) { (void)hipFree(np); return r; }
        HIP_TRY(c, hipMemcpyAsync(np, b.p, b.bytes, hipMemcpyDeviceToDevice, c->stream));
        kept = b.bytes;
    }
    if (fill >= 0) HIP_TRY(c, hipMemsetAsync(static_cast<char *>(np) + kept, fill, want - kept, c->stream));
    if (b.p) {
        int r = sync_all(c);  // any stream may still be reading the old allocation
        if (r != R3N_OK) return r;
        HIP_TRY(c, hipFree(b.p));
    }
    b.p = np;
    b.bytes = want;
    return R3N_OK;
}

// Asynchronous small upload through the pinned ring; falls back to a synchronous copy for large payloads.
int upload_small(r3n_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes > r3n_ctx::kStageSlotBytes || !c->stage) {
        HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_WAIT(c, hipStreamSynchronize(c->stream));
        return R3N_OK;
    }
    const uint32_t half_slots = r3n_ctx::kStageSlots / 2;
    const uint32_t slot = c->stage_next;
    const uint32_t half = slot / half_slots;
    if (slot % half_slots == 0 && c->stage_half_pending[half]) {
        // entering a half of the ring: every copy previously issued from it must have executed
        HIP_WAIT(c, hipEventSynchronize(c->stage_half_done[half]));
        c->stage_half_pending[half] = false;
    }
    uint8_t *h = c->stage + (size_t)slot * r3n_ctx::kStageSlotBytes;
    std::memcpy(h, src, bytes);
    HIP_TRY(c, hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, c->stream));
    ++c->main_epoch;
    c->stage_next = (slot + 1) % r3n_ctx::kStageSlots;
    if (c->stage_next % half_slots == 0) {
        HIP_TRY(c, hipEventRecord(c->stage_half_done[half], c->stream));
        c->stage_half_pending[half] = true;
    }
    return R3N_OK;
}

// Asynchronous upload of any size through the pinned bulk buffers (round robin).
int upload_bulk(r3n_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes <= r3n_ctx::kStageSlotBytes) return upload_small(c, dst, src, bytes);
    r3n_ctx::Bulk &b = c->bulk[c->bulk_next];
    c->bulk_next = (c->bulk_next + 1u) % 4u;
    if (b.pending) {
        HIP_WAIT(c, hipEventSynchronize(b.ev));
        b.pending = false;
    }
    if (b.bytes < bytes) {
        if (b.p) (void)hipHostFree(b.p);
        b.p = nullptr; b.bytes = 0;
        const size_t want = bytes + bytes / 4;
        HIP_TRY(c, hipHostMalloc((void **)&b.p, want, hipHostMallocDefault));
        b.bytes = want;
    }
    if (!b.ev) HIP_TRY(c, hipEventCreateWithFlags(&b.ev, hipEventDisableTiming));
    std::memcpy(b.p, src, bytes);
    HIP_TRY(c, hipMemcpyAsync(dst, b.p, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(b.ev, c->stream));
    b.pending = true;
    ++c->main_epoch;
    return R3N_OK;
}

// A kernel of an EARLIER call ran out of a fixed-size buffer (transparent pass: fragment nodes / work items).  Nothing on the
// frame path reads counts back, so the condition surfaces here: at the next frame, sync or read-back.
int check_async_status(r3n_ctx *c) {
    if (c->status_host && c->status_host[0] != 0u) {
        c->status_host[0] = 0u;
        return fail(c, R3N_ERR_CAPACITY, "an earlier transparent pass overflowed the fragment buffer or the raster work queue (r3n_config.max_big_items); that frame's image is incomplete");
    }
    return R3N_OK;
}

#define TRY(expr)                      \
    do {                               \
        int _r = (expr);               \
        if (_r != R3N_OK) return _r;   \
    } while (0)

static const char *const kStageNames[R3N_STAGE_COUNT] = {"bake", "object_cull", "triangle_cull", "hiz", "raster", "shade", "tonemap", "clear",
    "raster_big", "shadow_raster", "shadow_raster_big", "skinning", "vertex", "pose", "exchange_shadow", "exchange_depth",
    "exchange_rows", "exchange_keys", "raster_cut", "raster_big_cut"};
static void crumb(const r3n_ctx *c, const char *what, long long a, long long b) {
    if (c->crumb_fd < 0) return;
    char line[160];
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const int n = std::snprintf(line, sizeof line, "%lld.%06ld frame %llu %s %lld %lld\n", (long long)ts.tv_sec, ts.tv_nsec / 1000,
                                (unsigned long long)c->frame_no, what, a, b);
    if (n > 0) (void)!write(c->crumb_fd, line, (size_t)n);
}

struct Timed {
    r3n_ctx *c;
    int stage;
    hipStream_t stream;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(r3n_ctx *ctx, int st, hipStream_t on = nullptr) : c(ctx), stage(st), stream(on ? on : ctx->stream) {
        c->stage_launches[stage]++;
        crumb(c, kStageNames[stage]);
        if (!c->timing) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        a = get(); b = get();
        (void)hipEventRecord(a, stream);
    }
    ~Timed() {
        if (c->sync_stages) {  // R3N_SYNC_STAGES=1 (diagnosis): wait for the stage here, so that the last breadcrumb names the kernel that does not return
            (void)hipStreamSynchronize(stream);
            crumb(c, "stage done", stage);
        }
        if (!a) return;
        (void)hipEventRecord(b, stream);
        c->spans.push_back({a, b, stage, stream});
    }
};

// stream lane of a camera: 0 = main stream, 1 + k = auxiliary stream k
int cam_lane(const r3n_ctx *c, r3n_camera cam) {
    if (cam == R3N_CAMERA_VIEWPORT || !c->multi_stream) return 0;
    return 1 + (int)(cam % R3N_AUX_STREAMS);
}
hipStream_t lane_stream(const r3n_ctx *c, int lane) { return lane == 0 ? c->stream : c->aux[lane - 1]; }

// Order the lane's stream after everything enqueued on the main stream so far (clears, uploads).
int fork_lane(r3n_ctx *c, int lane) {
    if (lane == 0) return R3N_OK;
    const int k = lane - 1;
    c->aux_used[k] = true;
    // already ordered behind everything the main stream holds for it.  (Every main-stream producer the lanes read must bump
    // main_epoch; R3N_ALWAYS_FORK=1 forks unconditionally -- the test suite runs once that way, so a producer that forgets
    // the bump shows up as a difference between the two runs instead of a silent race.)
    if (c->lane_epoch[k] == c->main_epoch && !c->always_fork) return R3N_OK;
    HIP_TRY(c, hipEventRecord(c->fork_ev[k], c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->aux[k], c->fork_ev[k], 0));
    c->lane_epoch[k] = c->main_epoch;
    return R3N_OK;
}
// Order the main stream after everything enqueued on the auxiliary streams so far.
int join_lanes(r3n_ctx *c) {
    for (int k = 0; k < R3N_AUX_STREAMS; ++k)
        if (c->aux_used[k]) {
            HIP_TRY(c, hipEventRecord(c->join_ev[k], c->aux[k]));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->join_ev[k], 0));
            c->aux_used[k] = false;
        }
    return R3N_OK;
}
// Order the main stream after the latest resolve (shade stream).
int join_shade(r3n_ctx *c) {
    if (c->shade_unjoined) {
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->shade_done[c->shade_last], 0));
        c->shade_unjoined = false;
    }
    return R3N_OK;
}
int sync_all(r3n_ctx *c) {
    crumb(c, "sync_all");
    int r = join_lanes(c);
    if (r != R3N_OK) return r;
    r = join_shade(c);
    if (r != R3N_OK) return r;
    HIP_WAIT(c, hipStreamSynchronize(c->stream));
    if (c->shade) HIP_WAIT(c, hipStreamSynchronize(c->shade));
    crumb(c, "sync_all done");
    return R3N_OK;
}

int check_launch(r3n_ctx *c, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, R3N_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    return R3N_OK;
}

CamState *find_cam(r3n_ctx *c, r3n_camera cam, bool create) {
    if (cam == R3N_CAMERA_VIEWPORT) return &c->viewport;
    if (cam >= R3N_MAX_SHADOW_VIEWS) return nullptr;
    auto it = c->shadows.find(cam);
    if (it != c->shadows.end()) return &it->second;
    if (!create) return nullptr;
    return &c->shadows[cam];
}

void free_cam(CamState &s) {
    DevBuf *bufs[] = {&s.own_hdr, &s.chain, &s.baked, &s.vis_flags, &s.vis_list, &s.first_entry, &s.block_sums, &s.block_off, &s.slot_base[0],
                      &s.slot_base[1], &s.mask[0], &s.mask[1], &s.predicted[0], &s.predicted[1], &s.sub_counts[0],
                      &s.sub_counts[1], &s.counts[0], &s.counts[1], &s.residual};
    for (DevBuf *b : bufs)
        if (b->p) { (void)hipFree(b->p); b->p = nullptr; b->bytes = 0; }
}

// the object slots a camera culls / draws here: its own range when it has one (a shadow view owned whole), else the context's
// owner bytes (spatial partition) or slot range
ObjOwn camera_own(const r3n_ctx *c, const CamState &s) {
    if (s.range_set) return ObjOwn{s.range_begin, s.range_end, nullptr, 0u};
    if (c->owners.p) return ObjOwn{0u, 0u, c->owners.as<uint8_t>(), c->owner_rank};
    return ObjOwn{c->range_begin, c->range_end, nullptr, 0u};
}

uint32_t max_waves(const r3n_ctx *c) { return (uint32_t)(c->total_tris / 64u) + c->capacity + 1u; }

// ObjSoA.meta of a slot from the host mirrors: triangles of the enabled object | Material::key() of its material << 30 (a material
// index beyond the table reads key 0, as the kernels' former `mi < n_materials ? keys[mi] : 0` did)
uint32_t obj_meta_word(const r3n_ctx *c, uint32_t slot) {
    const uint32_t mi = c->h_material[slot];
    uint32_t key = mi < c->h_material_key.size() ? c->h_material_key[mi] : 0u;
    if (key > 2u) key = 2u;
    return c->h_ntri[slot] | (key << R3N_META_KEY_SHIFT);
}
ObjSoA obj_soa(const r3n_ctx *c) { return ObjSoA{c->spheres.as<float4>(), c->obj_meta.as<uint32_t>()}; }
// r3n_materials_write changed a key (or the table grew under objects that already named the new slots): every slot's key bits
// from the mirrors, one upload.  World-edit rate, in front of the next object pass.
int refresh_obj_meta(r3n_ctx *c) {
    if (!c->meta_dirty) return R3N_OK;
    c->meta_dirty = false;
    if (c->capacity == 0) return R3N_OK;
    std::vector<uint32_t> meta(c->capacity);
    for (uint32_t i = 0; i < c->capacity; ++i) meta[i] = obj_meta_word(c, i);
    TRY(join_lanes(c));  // the lanes' object passes of the previous frame read the old words
    ++c->main_epoch;
    HIP_TRY(c, hipMemcpyAsync(c->obj_meta.p, meta.data(), (size_t)c->capacity * 4u, hipMemcpyHostToDevice, c->stream));
    HIP_WAIT(c, hipStreamSynchronize(c->stream));  // `meta` is a temporary
    return R3N_OK;
}

// the first-entry table of a camera (kernels_cull.h write_first_entries): one word per R3N_CHUNK_ITERS wave slots
size_t first_entry_bytes(const r3n_ctx *c) { return ((size_t)max_waves(c) / R3N_CHUNK_ITERS + 2u) * 4u; }

// Object pass for one camera state (frustum cull + slot assignment); range (0,0) builds only tri_base.
int run_object_pass(r3n_ctx *c, CamState &s, int idx, ObjOwn own, uint32_t *tri_base,
                    hipStream_t stream) {
    const uint32_t cap = c->capacity;
    const uint32_t nblocks = (cap + 255u) / 256u;
    TRY(ensure(c, s.vis_flags, cap, true, 0));
    TRY(ensure(c, s.vis_list, (size_t)(cap + 1u) * sizeof(r3n_vis_entry), false, -1));
    TRY(ensure(c, s.block_sums, (size_t)nblocks * sizeof(ObjBlockSums), false, -1));
    TRY(ensure(c, s.block_off, (size_t)nblocks * sizeof(ObjBlockOffsets), false, -1));
    TRY(ensure(c, s.slot_base[idx], (size_t)cap * 4u, true, 0xFF));
    TRY(ensure(c, s.sub_counts[idx], sizeof(r3n_sub_counts), false, 0));
    TRY(ensure(c, s.counts[idx], sizeof(r3n_cull_counts), false, 0));
    if (tri_base == nullptr) TRY(ensure(c, s.first_entry, first_entry_bytes(c), false, -1));  // (the canonical scan feeds no triangle cull)
    uint32_t *const first_entry = tri_base == nullptr ? s.first_entry.as<uint32_t>() : nullptr;
    Timed t(c, R3N_STAGE_OBJECT_CULL, stream);
    if (cap <= R3N_FUSED_OBJECT_PASS_MAX) {  // small worlds: the three passes in one single-block launch
        hipLaunchKernelGGL(k_object_pass_fused, dim3(1), dim3(1024), 0, stream, s.d_hdr.as<r3n_camera_header240>(), obj_soa(c), own,
                           s.vis_flags.as<uint8_t>(), s.counts[idx].as<r3n_cull_counts>(), s.vis_list.as<r3n_vis_entry>(),
                           s.sub_counts[idx].as<r3n_sub_counts>(), s.slot_base[idx].as<uint32_t>(), tri_base, first_entry);
        return check_launch(c, "object pass (fused)");
    }
    hipLaunchKernelGGL(k_object_count, dim3(nblocks), dim3(256), 0, stream, s.d_hdr.as<r3n_camera_header240>(), obj_soa(c), own,
                       s.vis_flags.as<uint8_t>(), s.block_sums.as<ObjBlockSums>());
    hipLaunchKernelGGL(k_object_scan, dim3(1), dim3(nblocks <= 64u ? 64 : 1024), 0, stream, s.block_sums.as<ObjBlockSums>(), nblocks,
                       s.block_off.as<ObjBlockOffsets>(), s.counts[idx].as<r3n_cull_counts>(),
                       s.vis_list.as<r3n_vis_entry>(), s.sub_counts[idx].as<r3n_sub_counts>());
    hipLaunchKernelGGL(k_object_scatter, dim3(nblocks), dim3(256), 0, stream,
                       s.d_hdr.as<r3n_camera_header240>(), obj_soa(c), s.vis_flags.as<uint8_t>(),
                       s.block_off.as<ObjBlockOffsets>(), s.vis_list.as<r3n_vis_entry>(),
                       s.slot_base[idx].as<uint32_t>(), tri_base, first_entry);
    return check_launch(c, "object pass");
}

// Bake + object pass of one camera as ONE launch (k_object_pass_chained); false when the world is too large for the chained form.
// rounds of 256 slots per block: one up to MAX_BLOCKS x 256 slots, then as many as keep the grid at MAX_BLOCKS
uint32_t chained_rounds(const r3n_ctx *c) {
    const uint32_t nblocks = (c->capacity + 255u) / 256u;
    return std::max(1u, (nblocks + R3N_CHAINED_OBJECT_PASS_MAX_BLOCKS - 1u) / R3N_CHAINED_OBJECT_PASS_MAX_BLOCKS);
}
bool chained_pass_fits(const r3n_ctx *c) {
    return c->capacity >= 1u && chained_rounds(c) <= R3N_CHAINED_OBJECT_PASS_MAX_ROUNDS;
}
int run_bake_and_object_pass(r3n_ctx *c, CamState &s, int idx, ObjOwn own, hipStream_t stream, bool viewport) {
    const uint32_t cap = c->capacity;
    const uint32_t rounds = chained_rounds(c);
    const uint32_t grid = ((cap + 255u) / 256u + rounds - 1u) / rounds;
    TRY(ensure(c, s.vis_flags, cap, true, 0));
    TRY(ensure(c, s.vis_list, (size_t)(cap + 1u) * sizeof(r3n_vis_entry), false, -1));
    TRY(ensure(c, s.slot_base[idx], (size_t)cap * 4u, true, 0xFF));
    TRY(ensure(c, s.sub_counts[idx], sizeof(r3n_sub_counts), false, 0));
    TRY(ensure(c, s.counts[idx], sizeof(r3n_cull_counts), false, 0));
    TRY(ensure(c, s.first_entry, first_entry_bytes(c), false, -1));
    TRY(ensure(c, s.chain, (size_t)grid * sizeof(ObjChainRec), false, 0));  // zero tags: no launch has epoch 0
    if (++s.chain_epoch == 0u) ++s.chain_epoch;
    ObjChainArgs a{};
    a.hdr = s.d_hdr.as<r3n_camera_header240>();
    a.soa = obj_soa(c);
    a.objects = c->objects.as<r3n_object128>();
    a.own = own;
    a.vis_flags = s.vis_flags.as<uint8_t>();
    a.chain = s.chain.as<ObjChainRec>();
    a.epoch = s.chain_epoch;
    a.rounds = rounds;
    a.counts = s.counts[idx].as<r3n_cull_counts>();
    a.vis_list = s.vis_list.as<r3n_vis_entry>();
    a.sub_counts = s.sub_counts[idx].as<r3n_sub_counts>();
    a.slot_base = s.slot_base[idx].as<uint32_t>();
    // last frame's predicted list (this frame's first pass) names objects that passed the frustum test then: vis_flags still
    // holds that object pass's bytes (slots the world has grown by since read 0: the buffer grows zero-filled)
    a.use_prev = (viewport && s.has_prev) ? 1u : 0u;
    a.baked = s.baked.as<r3n_baked128>();
    a.first_entry = s.first_entry.as<uint32_t>();
    Timed t(c, R3N_STAGE_OBJECT_CULL, stream);
    hipLaunchKernelGGL(k_object_pass_chained<true>, dim3(grid), dim3(256), 0, stream, a);
    return check_launch(c, "k_object_pass_chained");
}

int refresh_tri_base(r3n_ctx *c) {
    if (!c->tri_base_dirty) return R3N_OK;
    if (c->capacity == 0) { c->tri_base_dirty = false; return R3N_OK; }
    TRY(join_shade(c));  // frames in flight: the previous frame's resolve reads tri_base / slot_table
    TRY(ensure(c, c->tri_base, (size_t)c->capacity * 4u, false, 0));
    r3n_camera_header240 h{};
    h.object_count = c->capacity;
    h.shadow_index = 0;
    TRY(ensure(c, c->canon.own_hdr, sizeof h, false, -1));
    c->canon.d_hdr = c->canon.own_hdr;
    ++c->main_epoch;  // tri_base / slot_table are rebuilt on the main stream: the lanes' rasterisers read them
    HIP_TRY(c, hipMemcpyAsync(c->canon.d_hdr.p, &h, sizeof h, hipMemcpyHostToDevice, c->stream));
    HIP_WAIT(c, hipStreamSynchronize(c->stream));  // h is a stack temporary
    TRY(run_object_pass(c, c->canon, 0, ObjOwn{0u, 0u, nullptr, 0u}, c->tri_base.as<uint32_t>(), c->stream));
    c->slot_table_size = (uint32_t)(c->total_tris >> R3N_SLOT_TABLE_SHIFT) + 1u;
    TRY(ensure(c, c->slot_table, (size_t)c->slot_table_size * 4u, false, -1));
    hipLaunchKernelGGL(k_build_slot_table, dim3((c->slot_table_size + 255u) / 256u), dim3(256), 0, c->stream,
                       c->tri_base.as<uint32_t>(), c->capacity, c->slot_table.as<uint32_t>(), c->slot_table_size);
    TRY(check_launch(c, "k_build_slot_table"));
    c->tri_base_dirty = false;
    return R3N_OK;
}

void build_hiz_desc(r3n_hiz_desc &d, uint32_t w, uint32_t h) {
    d.width = w; d.height = h;
    uint32_t m = std::max(w, h), n = 0;
    while (m) { ++n; m >>= 1; }
    d.mips = std::min<uint32_t>(n, R3N_MAX_HIZ_MIPS);
    uint32_t off = 0;
    for (uint32_t k = 0; k < R3N_MAX_HIZ_MIPS; ++k) {
        d.offset[k] = off;
        if (k < d.mips) off += std::max(1u, w >> k) * std::max(1u, h >> k);
    }
}
size_t hiz_elements(const r3n_hiz_desc &d) {
    size_t n = 0;
    for (uint32_t k = 0; k < d.mips; ++k) n += (size_t)std::max(1u, d.width >> k) * std::max(1u, d.height >> k);
    return n;
}

int drain_timing(r3n_ctx *c) {
    if (c->spans.empty()) return R3N_OK;
    TRY(sync_all(c));
    for (auto &s : c->spans) {
        float ms = 0.f;
        // the span of an EMPTY launch between two events (calibrated in r3n_timing_enable) is the events' own cost: taken off, so
        // that a stage's figure is its kernels' time -- what rocprofv3's kernel trace reports -- not kernels + dispatch gaps
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) c->stage_ms[s.stage] += std::max(0.0, (double)ms - c->span_overhead_ms);
        c->event_pool.push_back(s.a);
        c->event_pool.push_back(s.b);
    }
    c->spans.clear();
    return R3N_OK;
}

// tools/frame_timeline.py: the pending spans as (stage, stream index in order of first use, start, end) in milliseconds from the first
// span's start -- the schedule the streams actually ran, which rocprofv3's kernel trace cannot show (it serialises the dispatches).
// Consumes the spans like drain_timing.  Returns the number of entries written (<= max) or a negative error.
int read_timeline(r3n_ctx *c, float *out, int max) {
    if (c->spans.empty()) return 0;
    if (sync_all(c) != R3N_OK) return -1;
    std::vector<hipStream_t> streams;
    int n = 0;
    for (auto &s : c->spans) {
        float t0 = 0.f, t1 = 0.f;
        size_t si = std::find(streams.begin(), streams.end(), s.stream) - streams.begin();
        if (si == streams.size()) streams.push_back(s.stream);
        if (n < max && hipEventElapsedTime(&t0, c->spans[0].a, s.a) == hipSuccess && hipEventElapsedTime(&t1, c->spans[0].a, s.b) == hipSuccess) {
            out[4 * n + 0] = (float)s.stage; out[4 * n + 1] = (float)si; out[4 * n + 2] = t0; out[4 * n + 3] = t1;
            ++n;
        }
        c->event_pool.push_back(s.a);
        c->event_pool.push_back(s.b);
    }
    c->spans.clear();
    return n;
}

// "key=value key=value ..." -> ctx.tune (unknown keys and out-of-range values are refused: nothing is applied then)
int apply_tuning(r3n_ctx *c, const char *kv) {
    r3n_ctx::Tune t = c->tune;
    std::string text(kv ? kv : "");
    for (size_t at = 0; at < text.size();) {
        while (at < text.size() && (text[at] == ' ' || text[at] == ',')) ++at;
        if (at >= text.size()) break;
        const size_t end = text.find_first_of(" ,", at), eq = text.find('=', at);
        if (eq == std::string::npos || (end != std::string::npos && eq > end)) return fail(c, R3N_ERR_INVALID_ARG, "tuning: expected key=value");
        const std::string key = text.substr(at, eq - at);
        char *stop = nullptr;
        const unsigned long val = std::strtoul(text.c_str() + eq + 1, &stop, 10);
        if (stop == text.c_str() + eq + 1 || (*stop && *stop != ' ' && *stop != ',')) return fail(c, R3N_ERR_INVALID_ARG, "tuning: " + key + " wants an unsigned number");
        struct { const char *name; uint32_t *p; unsigned long lo, hi, step; } keys[] = {
            {"big_lds", &t.big_lds, 0, 65536, 1}, {"vp_big_lds", &t.vp_big_lds, 0, 65536, 1}, {"small_lds", &t.small_lds, 0, 65536, 1},
            {"vp_small_lds", &t.vp_small_lds, 0, 65536, 1}, {"cut_big_lds", &t.cut_big_lds, 0, 65536, 1}, {"vp_cut_big_lds", &t.vp_cut_big_lds, 0, 65536, 1},
            {"cut_small_lds", &t.cut_small_lds, 0, 65536, 1}, {"vp_cut_small_lds", &t.vp_cut_small_lds, 0, 65536, 1}, {"cull_lds", &t.cull_lds, 0, 65000, 1}, {"vp_cull_lds", &t.vp_cull_lds, 0, 65000, 1},
            {"resolve_lds", &t.resolve_lds, 0, 61440, 1}, {"big_grid", &t.big_grid, 256, 65536, 1},
            {"small_grid", &t.small_grid, R3N_SUBQ, 32768, R3N_SUBQ}, {"timed_pipeline", &t.timed_pipeline, 0, 1, 1},
            {"comm_serial", &t.comm_serial, 0, 1, 1}, {"prio_main", &t.prio_main, 0, 2, 1}, {"prio_shade", &t.prio_shade, 0, 2, 1}, {"prio_aux", &t.prio_aux, 0, 2, 1}};  // (read at r3n_create only)  // (a multiple of R3N_SUBQ: whole blocks per sub-list)
        bool known = false;
        for (auto &k : keys)
            if (key == k.name) {
                if (val < k.lo || val > k.hi || val % k.step) return fail(c, R3N_ERR_INVALID_ARG, "tuning: value out of range for " + key);
                *k.p = (uint32_t)val;
                known = true;
            }
        if (!known) return fail(c, R3N_ERR_INVALID_ARG, "tuning: unknown key " + key);
        at = end == std::string::npos ? text.size() : end;
    }
    c->tune = t;
    return R3N_OK;
}

}  // namespace

// tools/tune_caps.py: the launch parameters of ctx.tune between frames of one context (a search over hundreds of settings in one
// process); R3N_TUNE applies the same string at r3n_create.
extern "C" int r3n_internal_set_tuning(r3n_ctx *c, const char *kv) { return c ? apply_tuning(c, kv) : R3N_ERR_INVALID_ARG; }
extern "C" int r3n_internal_read_timeline(r3n_ctx *c, float *out, int max) { return c && out ? read_timeline(c, out, max) : -1; }

// The frame's clears in one launch: three zero fills (16-byte stores; a buffer's last < 4 words go singly).
__global__ __launch_bounds__(256) static void k_frame_clear(uint32_t *__restrict__ a, size_t a_words, uint32_t *__restrict__ b, size_t b_words,
                                                            uint32_t *__restrict__ c3, size_t c_words) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    const size_t stride = (size_t)gridDim.x * 256u, t = (size_t)blockIdx.x * 256u + threadIdx.x;
    for (size_t i = t; i < a_words / 4u; i += stride) reinterpret_cast<uint4 *>(a)[i] = z;
    for (size_t i = t; i < b_words / 4u; i += stride) reinterpret_cast<uint4 *>(b)[i] = z;
    for (size_t i = t; i < c_words / 4u; i += stride) reinterpret_cast<uint4 *>(c3)[i] = z;
    if (t < 4u) {  // the last < 4 words of each
        if ((a_words & ~(size_t)3u) + t < a_words) a[(a_words & ~(size_t)3u) + t] = 0u;
        if ((b_words & ~(size_t)3u) + t < b_words) b[(b_words & ~(size_t)3u) + t] = 0u;
        if ((c_words & ~(size_t)3u) + t < c_words) c3[(c_words & ~(size_t)3u) + t] = 0u;
    }
}

extern "C" {

const char *r3n_create_error(void) { return g_create_error.c_str(); }

r3n_ctx *r3n_create(int hip_device, const r3n_config *config) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        g_create_error = std::string("no HIP device: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (hip_device < 0 || hip_device >= n) { g_create_error = "hip_device out of range"; return nullptr; }
    e = hipSetDevice(hip_device);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, hip_device);
    if (e != hipSuccess) { g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e); return nullptr; }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        g_create_error = std::string("this build targets gfx950 (MI355X) only; device is ") + prop.gcnArchName;
        return nullptr;
    }
    r3n_ctx *c = new r3n_ctx();
    c->device = hip_device;
    if (config && config->struct_size >= sizeof(r3n_config) && config->max_big_items)
        c->big_capacity = std::max(1024u, config->max_big_items / R3N_BIGQ);
    if (config && config->struct_size >= sizeof(r3n_config) && config->shade_mode <= R3N_SHADE_FAST) c->shade_mode = config->shade_mode;
    if (const char *et = std::getenv("R3N_TUNE"))
        if (apply_tuning(c, et) != R3N_OK) {
            g_create_error = "R3N_TUNE: " + c->err;
            delete c;
            return nullptr;
        }
    // (stream priorities, ctx.tune.prio_*: 0 high, 1 normal, 2 low -- the runtime's range is [-1, 1])
    e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, (int)c->tune.prio_main - 1);
    if (e != hipSuccess) {
        g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        delete c;
        return nullptr;
    }
    if (hipHostMalloc((void **)&c->stage, (size_t)r3n_ctx::kStageSlots * r3n_ctx::kStageSlotBytes, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&c->stage_half_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->stage_half_done[1], hipEventDisableTiming) != hipSuccess) {
        g_create_error = "pinned staging ring allocation failed";
        r3n_destroy(c);
        return nullptr;
    }
    if (hipHostMalloc((void **)&c->status_host, 64, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&c->status_dev, c->status_host, 0) != hipSuccess) {
        g_create_error = "host-mapped status word allocation failed";
        r3n_destroy(c);
        return nullptr;
    }
    c->status_host[0] = 0u;
    if (const char *eb = std::getenv("R3N_BREADCRUMBS")) {
        static std::atomic<int> n_ctx{0};
        const std::string path = std::string(eb) + ".pid" + std::to_string((long long)getpid()) + ".ctx" + std::to_string(n_ctx.fetch_add(1));
        c->crumb_fd = open(path.c_str(), O_CREAT | O_WRONLY | O_APPEND, 0644);
        crumb(c, "created");
    }
    if (const char *es = std::getenv("R3N_SYNC_STAGES")) c->sync_stages = es[0] == '1';
    if (const char *e1 = std::getenv("R3N_SINGLE_STREAM")) c->multi_stream = !(e1[0] == '1');
    if (const char *e2 = std::getenv("R3N_PIPELINE")) c->overlap = !(e2[0] == '0');
    if (const char *e5 = std::getenv("R3N_RESOLVE_CLASSES")) c->resolve_classes = !(e5[0] == '0');
    if (const char *e6 = std::getenv("R3N_ALWAYS_FORK")) c->always_fork = e6[0] == '1';
    if (const char *e3 = std::getenv("R3N_EDGE_CAPACITY")) c->edge_capacity_override = (uint32_t)std::strtoul(e3, nullptr, 10);
    if (hipStreamCreateWithPriority(&c->shade, hipStreamNonBlocking, (int)c->tune.prio_shade - 1) != hipSuccess ||
        hipEventCreateWithFlags(&c->vp_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->shade_done[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->shade_done[1], hipEventDisableTiming) != hipSuccess) {
        g_create_error = "shade stream creation failed";
        r3n_destroy(c);
        return nullptr;
    }
    for (int k = 0; k < R3N_AUX_STREAMS; ++k)
        if (hipStreamCreateWithPriority(&c->aux[k], hipStreamNonBlocking, (int)c->tune.prio_aux - 1) != hipSuccess ||
            hipEventCreateWithFlags(&c->fork_ev[k], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->join_ev[k], hipEventDisableTiming) != hipSuccess) {
            g_create_error = "auxiliary stream creation failed";
            r3n_destroy(c);
            return nullptr;
        }
    // Bind every stream to its hardware queue NOW, in this order: the runtime hands queues out at a stream's first submission, and
    // the frame is tuned for main / shade / two pairs of lanes on the four queues the runtime uses by default (measured: with a
    // communication library initialised first its streams take queues ahead of these and the same frame takes 1.65 ms instead of
    // 1.19 -- so create the context BEFORE torch.distributed / RCCL is initialised, bench.py does).
    {
        hipStream_t order[2 + R3N_AUX_STREAMS] = {c->stream, c->shade};
        for (int k = 0; k < R3N_AUX_STREAMS; ++k) order[2 + k] = c->aux[k];
        void *probe = nullptr;
        if (hipMalloc(&probe, 256) == hipSuccess) {
            for (hipStream_t st : order) {
                (void)hipMemsetAsync(probe, 0, 256, st);
                (void)hipStreamSynchronize(st);
            }
            (void)hipFree(probe);
        }
    }
    // empty light buffers: count = 0
    bool ok = ensure(c, c->srgb_lut, R3N_SRGB_LUT_SIZE, false, -1) == R3N_OK && ensure(c, c->srgb8_decode, 512 * 4, false, -1) == R3N_OK &&
              ensure(c, c->tex_descs, sizeof(r3n_texture_desc32), false, 0) == R3N_OK && ensure(c, c->tex_texels, 4, false, 0) == R3N_OK && ensure(c, c->tex_level_off, 64, false, 0) == R3N_OK &&
              ensure(c, c->material_keys, 256, false, 0) == R3N_OK && ensure(c, c->materials, sizeof(r3n_material208), false, 0) == R3N_OK;
    // frame-constants blocks (zero: light counts 0) and their pinned host images
    for (int k = 0; ok && k < 2; ++k) ok = ensure(c, c->fb_dev[k], r3n_ctx::kFbBytes, false, 0) == R3N_OK;
    for (int k = 0; ok && k < r3n_ctx::kFbHost; ++k)
        ok = hipHostMalloc((void **)&c->fb_host[k], r3n_ctx::kFbBytes, hipHostMallocDefault) == hipSuccess &&
             hipEventCreateWithFlags(&c->fb_ev[k], hipEventDisableTiming) == hipSuccess;
    if (ok) {
        for (int k = 0; k < r3n_ctx::kFbHost; ++k) std::memset(c->fb_host[k], 0, r3n_ctx::kFbBytes);
        c->fu.p = c->fb_dev[0].as<uint8_t>() + r3n_ctx::kFbUniforms; c->fu.bytes = 512;
        c->dir_buf.p = c->fb_dev[0].as<uint8_t>() + r3n_ctx::kFbDir; c->dir_buf.bytes = r3n_ctx::kFbShadowHdr - r3n_ctx::kFbDir;
        c->point_buf.p = c->fb_dev[0].as<uint8_t>() + r3n_ctx::kFbPoint; c->point_buf.bytes = 8448;
    }
    ok = ok && ensure(c, c->big_count_all, (size_t)(1 + R3N_QLANES) * 64 * R3N_BIGQ * 4, false, 0) == R3N_OK;
    for (int lane = 0; ok && lane < 1 + R3N_QLANES; ++lane) {
        c->big_count[lane].p = c->big_count_all.as<uint32_t>() + (size_t)lane * 64 * R3N_BIGQ;  // not owned
        c->big_count[lane].bytes = 64 * R3N_BIGQ * 4;
        ok = ensure(c, c->big_items[lane], (size_t)c->big_capacity * R3N_BIGQ * sizeof(r3n_big_item), false, -1) == R3N_OK &&
             ensure(c, c->big_uv[lane], (size_t)c->big_capacity * R3N_BIGQ * sizeof(r3n_big_uv), false, -1) == R3N_OK;
    }
    if (ok) {
        (void)r3n_internal_build_srgb_lut(c->srgb_lut.as<unsigned char>(), c->stream);
        // sRGB8 -> linear decode table for texture fetches: built on the HOST (libm powf, like the oracle's), because
        // every entry feeds f32 filtering arithmetic directly -- a last-ulp difference between libm and the device
        // math library in any of the 256 entries would show up as 1-ulp HDR differences
        float decode[512];  // [0, 256): unorm8 -> float, [256, 512): sRGB8 -> linear
        for (int i = 0; i < 256; ++i) {
            const float e = (float)i / 255.0f;
            decode[i] = e;
            decode[256 + i] = e > 0.04045f ? std::pow((e + 0.055f) / 1.055f, 2.4f) : e / 12.92f;
        }
        ok = hipMemcpyAsync(c->srgb8_decode.p, decode, sizeof decode, hipMemcpyHostToDevice, c->stream) == hipSuccess;
        // encode thresholds for the GPU mip chain: thr[c - 1] = the smallest float x with (uint8)(oetf(x) * 255 + 0.5) >= c,
        // by bisection over the float bit patterns of [0, 1] with the libm expression the oracle uses
        static float thr[255];
        auto code_of = [](float x) {
            float e = !(x > 0.0f) ? 0.0f : (x >= 1.0f ? 1.0f : (x <= 0.0031308f ? x * 12.92f : 1.055f * std::pow(x, 1.0f / 2.4f) - 0.055f));
            return (uint32_t)(e * 255.0f + 0.5f);
        };
        for (uint32_t cc = 1; cc <= 255; ++cc) {
            uint32_t lo = 0u, hi = 0x3F800000u;  // code_of(hi) = 255 >= cc
            while (lo < hi) {
                const uint32_t mid = lo + (hi - lo) / 2u;
                float x;
                std::memcpy(&x, &mid, 4);
                if (code_of(x) >= cc) hi = mid; else lo = mid + 1u;
            }
            std::memcpy(&thr[cc - 1u], &lo, 4);
        }
        ok = ok && ensure(c, c->srgb_thr, sizeof thr, false, -1) == R3N_OK &&
             hipMemcpyAsync(c->srgb_thr.p, thr, sizeof thr, hipMemcpyHostToDevice, c->stream) == hipSuccess;
        ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
        if (!ok) c->err = "k_build_srgb_lut failed";
    }
    if (!ok) {
        g_create_error = c->err;
        r3n_destroy(c);
        return nullptr;
    }
    return c;
}

void r3n_destroy(r3n_ctx *c) {
    if (!c) return;
    crumb(c, "destroy");
    (void)hipSetDevice(c->device);
    (void)r3n_comm_destroy(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->shade) (void)hipStreamSynchronize(c->shade);
    for (int k = 0; k < R3N_AUX_STREAMS; ++k)
        if (c->aux[k]) (void)hipStreamSynchronize(c->aux[k]);
    for (int lane = 0; lane < 1 + R3N_QLANES; ++lane)
        for (DevBuf *b : {&c->big_items[lane], &c->big_uv[lane]})
            if (b->p) (void)hipFree(b->p);
    for (int k = 0; k < r3n_ctx::kFbHost; ++k) {
        if (c->fb_host[k]) (void)hipHostFree(c->fb_host[k]);
        if (c->fb_ev[k]) (void)hipEventDestroy(c->fb_ev[k]);
    }
    DevBuf *bufs[] = {&c->mesh, &c->objects, &c->spheres, &c->obj_meta, &c->materials, &c->material_keys, &c->fb_dev[0], &c->fb_dev[1], &c->big_count_all, &c->owners,
                      &c->tri_base, &c->slot_table, &c->skin_inputs, &c->skin_matrices, &c->skin_wave_skeleton,
                      &c->skin_wave_first, &c->skin_joint_counts, &c->vis, &c->hdr16, &c->out8, &c->out_f32, &c->atlas, &c->hiz, &c->alt_vis, &c->alt_atlas, &c->alt_vp_baked, &c->srgb_lut, &c->srgb_thr, &c->tex_descs, &c->tex_texels, &c->tex_level_off, &c->srgb8_decode,
                      &c->tri_rec, &c->tri_seen, &c->blend_order, &c->blend_rank_base, &c->frag_keys, &c->frag_vals, &c->frag_head,
                      &c->frag_count, &c->samples16, &c->anim_rigs, &c->anim_joints, &c->anim_clips, &c->anim_tracks,
                      &c->anim_times, &c->anim_values, &c->pose_requests, &c->edge_list, &c->edge_count, &c->material_feat, &c->view_lights[0],
                      &c->view_lights[1]};
    for (DevBuf *b : bufs)
        if (b->p) (void)hipFree(b->p);
    free_cam(c->canon);
    free_cam(c->viewport);
    for (auto &kv : c->shadows) free_cam(kv.second);
    if (c->vp_ev) (void)hipEventDestroy(c->vp_ev);
    for (auto e : c->shade_done)
        if (e) (void)hipEventDestroy(e);
    if (c->shade) (void)hipStreamDestroy(c->shade);
    for (auto &s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    for (int k = 0; k < R3N_AUX_STREAMS; ++k) {
        if (c->fork_ev[k]) (void)hipEventDestroy(c->fork_ev[k]);
        if (c->join_ev[k]) (void)hipEventDestroy(c->join_ev[k]);
        if (c->aux[k]) (void)hipStreamDestroy(c->aux[k]);
    }
    if (c->stage) (void)hipHostFree(c->stage);
    if (c->status_host) (void)hipHostFree(c->status_host);
    for (auto &b : c->bulk) {
        if (b.p) (void)hipHostFree(b.p);
        if (b.ev) (void)hipEventDestroy(b.ev);
    }
    for (auto e : c->stage_half_done)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *r3n_last_error(const r3n_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int r3n_sync(r3n_ctx *c) {
    if (!c) return R3N_ERR_INVALID_ARG;
    TRY(sync_all(c));
    return check_async_status(c);
}

void *r3n_stream(r3n_ctx *c) { return c ? (void *)c->stream : nullptr; }

// ------------------------------------------------------------------------------------------------ uploads
int r3n_mesh_buffer_write(r3n_ctx *c, uint64_t byte_offset, const void *data, uint64_t bytes) {
    if (!c || (!data && bytes) || (byte_offset & 3u) || (bytes & 3u)) return fail(c, R3N_ERR_INVALID_ARG, "mesh write: bad args");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(join_shade(c));  // frames in flight: world buffers are read by the previous frame's resolve
    TRY(ensure(c, c->mesh, byte_offset + bytes, true, 0));
    if (bytes) {
        HIP_TRY(c, hipMemcpyAsync(static_cast<char *>(c->mesh.p) + byte_offset, data, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_WAIT(c, hipStreamSynchronize(c->stream));  // caller owns `data` only for the duration of the call
    }
    return R3N_OK;
}

int r3n_objects_write(r3n_ctx *c, const uint32_t *slots, const r3n_object128 *records, uint32_t n, uint32_t capacity) {
    if (!c || (n && (!slots || !records))) return fail(c, R3N_ERR_INVALID_ARG, "objects write: null");
    if (capacity < c->capacity) return fail(c, R3N_ERR_INVALID_ARG, "objects write: capacity cannot shrink");
    for (uint32_t i = 0; i < n; ++i)
        if (slots[i] >= capacity) return fail(c, R3N_ERR_INVALID_ARG, "objects write: slot >= capacity");
    if (n == 0 && capacity == c->capacity) return R3N_OK;  // nothing dirty: no upload, no synchronisation
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(join_shade(c));
    TRY(ensure(c, c->objects, (size_t)capacity * sizeof(r3n_object128), true, 0));
    TRY(ensure(c, c->spheres, (size_t)capacity * sizeof(float4), true, 0));
    TRY(ensure(c, c->obj_meta, (size_t)capacity * 4u, true, 0));
    for (uint32_t i = 0; i < n; ++i)
        if (records[i].enabled && records[i].index_count / 3u > R3N_META_NTRI_MASK)
            return fail(c, R3N_ERR_UNSUPPORTED, "objects write: more than 2^30 - 1 triangles in one object");
    if (capacity != c->capacity) {
        if (c->owners.p && c->owners_n < capacity) {
            // owner bytes (r3n_set_object_owners) cover the old capacity: the kernels index them by slot, so the table grows with
            // the world and the new slots belong to rank 0 until the caller sends a new partition (never uninitialised bytes:
            // an object owned by no rank would vanish from every rank's frame)
            TRY(sync_all(c));
            TRY(ensure(c, c->owners, capacity, true, 0));
            // (the allocation may already have been large enough, its tail never written)
            HIP_TRY(c, hipMemsetAsync(c->owners.as<uint8_t>() + c->owners_n, 0, capacity - c->owners_n, c->stream));
            HIP_WAIT(c, hipStreamSynchronize(c->stream));
            c->owners_n = capacity;
        }
        c->capacity = capacity;
        c->h_ntri.resize(capacity, 0);
        c->h_material.resize(capacity, 0);
        c->tri_base_dirty = true;
    }
    for (uint32_t i = 0; i < n;) {
        uint32_t run = 1;  // consecutive slots upload as one copy
        while (i + run < n && slots[i + run] == slots[i] + run) ++run;
        HIP_TRY(c, hipMemcpyAsync(c->objects.as<r3n_object128>() + slots[i], records + i, (size_t)run * sizeof(r3n_object128),
                                  hipMemcpyHostToDevice, c->stream));
        i += run;
    }
    std::vector<float4> h_sph(n);
    std::vector<uint32_t> h_meta(n);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t nt = records[i].enabled ? records[i].index_count / 3u : 0u;
        // A slot rewritten in place with another triangle count (not reachable through rend3's own API, where a handle is
        // disabled for a frame before it is reused, but legal through this ABI) must not index last frame's result bits
        // with this frame's triangle numbers: it counts as "not batched last frame" (batching.rs:226).
        if (c->h_ntri[slots[i]] != 0u && nt != c->h_ntri[slots[i]] && c->viewport.has_prev) {
            DevBuf &pb = c->viewport.slot_base[1 - c->viewport.cur];
            if (pb.p && (size_t)(slots[i] + 1u) * 4u <= pb.bytes)
                HIP_TRY(c, hipMemsetAsync(pb.as<uint32_t>() + slots[i], 0xFF, 4, c->stream));
        }
        c->total_tris = c->total_tris - c->h_ntri[slots[i]] + nt;
        c->h_ntri[slots[i]] = nt;
        c->h_material[slots[i]] = records[i].material_index;
        c->key_census_dirty = true;
        // the structure-of-arrays view of the object pass (kernels_cull.h ObjSoA)
        h_sph[i] = make_float4(records[i].bounding_sphere_center[0], records[i].bounding_sphere_center[1],
                               records[i].bounding_sphere_center[2], records[i].bounding_sphere_radius);
        h_meta[i] = obj_meta_word(c, slots[i]);
    }
    for (uint32_t i = 0; i < n;) {
        uint32_t run = 1;
        while (i + run < n && slots[i + run] == slots[i] + run) ++run;
        HIP_TRY(c, hipMemcpyAsync(c->spheres.as<float4>() + slots[i], h_sph.data() + i, (size_t)run * sizeof(float4), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->obj_meta.as<uint32_t>() + slots[i], h_meta.data() + i, (size_t)run * 4u, hipMemcpyHostToDevice, c->stream));
        i += run;
    }
    if (n) c->tri_base_dirty = true;
    HIP_WAIT(c, hipStreamSynchronize(c->stream));
    return R3N_OK;
}

int r3n_materials_write(r3n_ctx *c, const uint32_t *slots, const r3n_material208 *records, const uint8_t *keys, uint32_t n) {
    if (!c || (n && (!slots || !records || !keys))) return fail(c, R3N_ERR_INVALID_ARG, "materials write: null");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(join_shade(c));
    uint32_t need = c->n_materials;
    for (uint32_t i = 0; i < n; ++i) {
        if (keys[i] > R3N_KEY_BLEND) return fail(c, R3N_ERR_INVALID_ARG, "materials write: bad key");
        need = std::max(need, slots[i] + 1u);
    }
    // a work item's `material` word carries the material index in bits 0..28 (edge thresholds in 29..31, kernels_raster.h)
    if (need > (1u << 29)) return fail(c, R3N_ERR_UNSUPPORTED, "materials write: material slots must stay below 2^29");
    TRY(ensure(c, c->materials, (size_t)need * sizeof(r3n_material208), true, 0));
    TRY(ensure(c, c->material_keys, need, true, 0));
    // objects carry their material's key in ObjSoA.meta: a key that changes, or a table that grows under objects which already name
    // the new slots, re-derives those bits (refresh_obj_meta, in front of the next object pass)
    if (need != c->n_materials && c->capacity) c->meta_dirty = true;
    c->n_materials = need;
    c->h_material_key.resize(need, 0);
    for (uint32_t i = 0; i < n; ++i) {
        if (c->h_material_key[slots[i]] != keys[i] && c->capacity) c->meta_dirty = true;
        c->h_material_key[slots[i]] = keys[i];
    }
    c->key_census_dirty = true;
    c->h_materials.resize(need, r3n_material208{});
    for (uint32_t i = 0; i < n; ++i) c->h_materials[slots[i]] = records[i];
    c->classes_dirty = true; c->cutout_short_dirty = true;
    for (uint32_t i = 0; i < n; ++i) {
        HIP_TRY(c, hipMemcpyAsync(c->materials.as<r3n_material208>() + slots[i], records + i, sizeof(r3n_material208),
                                  hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->material_keys.as<uint8_t>() + slots[i], keys + i, 1, hipMemcpyHostToDevice, c->stream));
    }
    HIP_WAIT(c, hipStreamSynchronize(c->stream));
    return R3N_OK;
}

static TextureArgs texture_args(r3n_ctx *c) {
    TextureArgs t;
    t.descs = c->tex_descs.as<r3n_texture_desc32>();
    t.count = c->n_textures;
    t.texels = c->tex_texels.as<uint32_t>();
    t.decode = c->srgb8_decode.as<float>();
    t.level_off = c->tex_level_off.as<uint32_t>();
    t.small_pool = c->n_texels <= (1ull << 30) ? 1u : 0u;  // byte offsets into the pool fit 32 bits: the sampler's short path
    return t;
}

// First word (pool index) of every level of every texture: R3N_TEX_LEVELS entries per texture, so that the sampler does not
// walk the chain.  `descs` = the descriptors as the device holds them (offsets in pool words; a texel is one word, or four
// for R3N_POOL_FLOAT textures).
static int upload_level_offsets(r3n_ctx *c, const r3n_texture_desc32 *descs, uint32_t n, uint64_t n_texels) {
    std::vector<uint32_t> off((size_t)std::max(n, 1u) * R3N_TEX_LEVELS, 0u);
    for (uint32_t i = 0; i < n; ++i) {
        uint64_t at = descs[i].offset;
        for (uint32_t k = 0; k < R3N_TEX_LEVELS; ++k) {
            off[(size_t)i * R3N_TEX_LEVELS + k] = (uint32_t)at;
            if (k < descs[i].mips) at += (uint64_t)std::max(1u, descs[i].width >> k) * std::max(1u, descs[i].height >> k) * (descs[i].format == R3N_POOL_FLOAT ? 4u : 1u);
        }
    }
    // which textures the sampler's short path covers (texture.h tex_sample_grad): power-of-two extents, RGBA8 pool texels
    c->h_tex_short.assign(n, 0);
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t w = descs[i].width, h = descs[i].height;
        // (w * h <= 2^29: tex_level_start_pow2 forms 1 << (log2 w + log2 h + 2) in 32 bits -- a single-level 32768^2 texture would
        // shift by 32; it goes to the general sampler instead)
        c->h_tex_short[i] = (w && h && ((w & (w - 1u)) | (h & (h - 1u))) == 0u && (uint64_t)w * h <= (1ull << 29) && descs[i].format < R3N_POOL_FLOAT) ? 1 : 0;
    }
    c->classes_dirty = true; c->cutout_short_dirty = true;
    TRY(ensure(c, c->tex_level_off, off.size() * 4, false, -1));
    HIP_TRY(c, hipMemcpyAsync(c->tex_level_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_WAIT(c, hipStreamSynchronize(c->stream));  // `off` is a temporary
    c->n_texels = n_texels;
    return R3N_OK;
}

int r3n_textures_write(r3n_ctx *c, const r3n_texture_desc32 *descs, uint32_t n, const uint32_t *texels, uint64_t n_texels) {
    if (!c || (n && (!descs || !texels))) return fail(c, R3N_ERR_INVALID_ARG, "textures write: null");
    for (uint32_t i = 0; i < n; ++i) {
        const r3n_texture_desc32 &d = descs[i];
        if (d.format > R3N_TEXTURE_RGBA8_UNORM_SRGB) return fail(c, R3N_ERR_UNSUPPORTED, "textures write: RGBA8 texels only here; other formats go through r3n_textures_write_encoded");
        if (!d.width || !d.height || !d.mips || d.width > 65535u || d.height > 65535u) return fail(c, R3N_ERR_INVALID_ARG, "textures write: bad extent");
        uint32_t max_mips = 0;
        for (uint32_t m = std::max(d.width, d.height); m; m >>= 1) ++max_mips;
        if (d.mips > max_mips) return fail(c, R3N_ERR_INVALID_ARG, "textures write: more mips than the extent has");
        uint64_t need = d.offset;
        for (uint32_t k = 0; k < d.mips; ++k) need += (uint64_t)std::max(1u, d.width >> k) * std::max(1u, d.height >> k);
        if (need > n_texels) return fail(c, R3N_ERR_INVALID_ARG, "textures write: mip chain outside the texel pool");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(sync_all(c));  // no frame may still sample the old array
    TRY(ensure(c, c->tex_descs, std::max<size_t>(n, 1) * sizeof(r3n_texture_desc32), false, -1));
    TRY(ensure(c, c->tex_texels, std::max<uint64_t>(n_texels, 1) * 4 + 16, false, -1));  // (+16: the sampler reads a footprint row as one 8-byte load -- the pool's last texel has a word behind it)
    if (n) {
        HIP_TRY(c, hipMemcpyAsync(c->tex_descs.p, descs, (size_t)n * sizeof(r3n_texture_desc32), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->tex_texels.p, texels, n_texels * 4, hipMemcpyHostToDevice, c->stream));
        HIP_WAIT(c, hipStreamSynchronize(c->stream));  // caller owns the sources only for the duration of the call
    }
    TRY(upload_level_offsets(c, descs, n, n_texels));
    c->n_textures = n;
    return R3N_OK;
}

extern "C" int r3n_internal_pose_skeletons(const void *requests, uint32_t n, const void *rigs, const void *joints, const void *clips,
                                           const void *tracks, const float *times, const float *values, float *out, uint32_t max_joints,
                                           hipStream_t stream);
extern "C" int r3n_internal_skinning_mfma(uint32_t *mesh, const void *inputs, const float *joint_matrices, const uint32_t *wave_skeleton,
                                          const uint32_t *wave_first, const uint32_t *skeleton_joints, uint32_t total_waves, hipStream_t stream);
extern "C" int r3n_internal_generate_mip(uint32_t srgb, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh, const uint32_t *src,
                                         uint32_t *dst, const float *decode, const float *thr, hipStream_t stream);
extern "C" uint64_t r3n_internal_level_bytes(uint32_t format, uint32_t w, uint32_t h);
extern "C" int r3n_internal_decode_level(uint32_t format, uint32_t w, uint32_t h, const void *src, uint32_t *dst, hipStream_t stream);
extern "C" int r3n_internal_decode_level_f32(uint32_t format, uint32_t w, uint32_t h, const void *src, float *dst, hipStream_t stream);
extern "C" int r3n_internal_format_is_float(uint32_t format);
extern "C" uint32_t r3n_internal_format_align(uint32_t format);
extern "C" int r3n_internal_format_generates_mips_f32(uint32_t format);
extern "C" int r3n_internal_generate_mip_f32(uint32_t format, uint32_t sw, uint32_t sh, uint32_t dw, uint32_t dh, const float *src, float *dst,
                                             hipStream_t stream);

int r3n_textures_write_encoded(r3n_ctx *c, const r3n_texture_desc32 *descs, uint32_t n, const void *payload, uint64_t payload_bytes) {
    if (!c || (n && (!descs || !payload))) return fail(c, R3N_ERR_INVALID_ARG, "textures write (encoded): null");
    std::vector<r3n_texture_desc32> internal(n);
    uint64_t n_texels = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const r3n_texture_desc32 &d = descs[i];
        if (d.format >= R3N_TEXTURE_FORMAT_COUNT) return fail(c, R3N_ERR_UNSUPPORTED, "textures write (encoded): unknown format id");
        const bool is_float = r3n_internal_format_is_float(d.format) != 0;
        if (!d.width || !d.height || !d.mips || d.width > 65535u || d.height > 65535u) return fail(c, R3N_ERR_INVALID_ARG, "textures write (encoded): bad extent");
        uint32_t max_mips = 0;
        for (uint32_t m = std::max(d.width, d.height); m; m >>= 1) ++max_mips;
        if (d.mips > max_mips) return fail(c, R3N_ERR_INVALID_ARG, "textures write (encoded): more mips than the extent has");
        if (d.stored_mips > d.mips) return fail(c, R3N_ERR_INVALID_ARG, "textures write (encoded): more stored levels than mips");
        const uint32_t stored = d.stored_mips ? d.stored_mips : d.mips;
        if (stored < d.mips && d.format >= R3N_TEXTURE_BC1_RGBA_UNORM && !r3n_internal_format_generates_mips_f32(d.format))
            return fail(c, R3N_ERR_UNSUPPORTED, is_float ? "textures write (encoded): among the float-decoded formats mips are generated for R16Float / Rg16Float / Rgba16Float / Rgb10a2Unorm only (filterable render targets); the others must carry their levels"
                                                         : "textures write (encoded): mips are generated for uncompressed formats only (block formats are not render targets)");
        uint64_t end = d.offset, texels = 0;
        for (uint32_t k = 0; k < d.mips; ++k) {
            const uint32_t w = std::max(1u, d.width >> k), h = std::max(1u, d.height >> k);
            if (k < stored) end += r3n_internal_level_bytes(d.format, w, h);
            texels += (uint64_t)w * h * (is_float ? 4u : 1u);  // pool words
        }
        if (end > payload_bytes) return fail(c, R3N_ERR_INVALID_ARG, "textures write (encoded): levels outside the payload");
        if ((d.offset & 3u) != 0u) return fail(c, R3N_ERR_INVALID_ARG, "textures write (encoded): level 0 must start on a 4-byte boundary");
        if (n_texels + texels > 0xFFFFFFFFull) return fail(c, R3N_ERR_CAPACITY, "textures write (encoded): texel pool exceeds 2^32 texels");
        const bool srgb = d.format == R3N_TEXTURE_RGBA8_UNORM_SRGB || d.format == R3N_TEXTURE_BGRA8_UNORM_SRGB ||
                          d.format == R3N_TEXTURE_BC1_RGBA_UNORM_SRGB || d.format == R3N_TEXTURE_BC2_RGBA_UNORM_SRGB ||
                          d.format == R3N_TEXTURE_BC3_RGBA_UNORM_SRGB || d.format == R3N_TEXTURE_BC7_RGBA_UNORM_SRGB;
        n_texels = (n_texels + 3u) & ~3ull;  // 16-byte-aligned texture starts: the block decoder stores whole rows
        internal[i] = d;
        internal[i].stored_mips = 0;
        internal[i].offset = (uint32_t)n_texels;
        internal[i].format = is_float ? R3N_POOL_FLOAT : (srgb ? R3N_TEXTURE_RGBA8_UNORM_SRGB : R3N_TEXTURE_RGBA8_UNORM);
        n_texels += texels;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(sync_all(c));  // no frame may still sample the old array
    TRY(ensure(c, c->tex_descs, std::max<size_t>(n, 1) * sizeof(r3n_texture_desc32), false, -1));
    TRY(ensure(c, c->tex_texels, std::max<uint64_t>(n_texels, 1) * 4 + 16, false, -1));  // (+16: the sampler reads a footprint row as one 8-byte load -- the pool's last texel has a word behind it)
    if (n) {
        void *staged = nullptr;
        HIP_TRY(c, hipMalloc(&staged, std::max<uint64_t>(payload_bytes, 4)));
        hipError_t e = hipMemcpyAsync(staged, payload, payload_bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(c->tex_descs.p, internal.data(), (size_t)n * sizeof(r3n_texture_desc32), hipMemcpyHostToDevice, c->stream);
        for (uint32_t i = 0; i < n && e == hipSuccess; ++i) {
            uint64_t src = descs[i].offset, dst = internal[i].offset;
            const uint32_t stored = descs[i].stored_mips ? descs[i].stored_mips : descs[i].mips;
            for (uint32_t k = 0; k < descs[i].mips && e == hipSuccess; ++k) {
                const uint32_t w = std::max(1u, descs[i].width >> k), h = std::max(1u, descs[i].height >> k);
                if (internal[i].format == R3N_POOL_FLOAT) {
                    float *level = reinterpret_cast<float *>(c->tex_texels.as<uint32_t>() + dst);
                    if (k < stored) {
                        // sources are read through their own width (format_align): every level of those formats is a multiple of it
                        e = (hipError_t)r3n_internal_decode_level_f32(descs[i].format, w, h, static_cast<const char *>(staged) + src, level, c->stream);
                        src += r3n_internal_level_bytes(descs[i].format, w, h);
                    } else {  // MipmapSource::Generated: blit of the level above, written in the texture's format
                        const uint32_t sw = std::max(1u, descs[i].width >> (k - 1u)), sh = std::max(1u, descs[i].height >> (k - 1u));
                        e = (hipError_t)r3n_internal_generate_mip_f32(descs[i].format, sw, sh, w, h, level - (uint64_t)sw * sh * 4u, level, c->stream);
                    }
                    dst += (uint64_t)w * h * 3u;  // + w * h below: four words per texel
                } else if (k < stored) {
                    // block-compressed and 32-bit sources are read as dwords: every level of those formats is a multiple of 4 B
                    e = (hipError_t)r3n_internal_decode_level(descs[i].format, w, h, static_cast<const char *>(staged) + src,
                                                              c->tex_texels.as<uint32_t>() + dst, c->stream);
                    src += r3n_internal_level_bytes(descs[i].format, w, h);
                } else {  // MipmapSource::Generated: blit of the level above, in the texture's format
                    const uint32_t sw = std::max(1u, descs[i].width >> (k - 1u)), sh = std::max(1u, descs[i].height >> (k - 1u));
                    e = (hipError_t)r3n_internal_generate_mip(internal[i].format == R3N_TEXTURE_RGBA8_UNORM_SRGB, sw, sh, w, h,
                                                              c->tex_texels.as<uint32_t>() + dst - (uint64_t)sw * sh,
                                                              c->tex_texels.as<uint32_t>() + dst, c->srgb8_decode.as<float>(),
                                                              c->srgb_thr.as<float>(), c->stream);
                }
                dst += (uint64_t)w * h;
            }
        }
        const hipError_t e2 = hipStreamSynchronize(c->stream);  // the caller owns the sources only for the duration of the call
        (void)hipFree(staged);
        if (e != hipSuccess) return fail(c, R3N_ERR_HIP, std::string("textures write (encoded): ") + hipGetErrorString(e));
        if (e2 != hipSuccess) return fail(c, R3N_ERR_HIP, std::string("textures write (encoded): ") + hipGetErrorString(e2));
    }
    TRY(upload_level_offsets(c, internal.data(), n, n_texels));
    c->n_textures = n;
    return R3N_OK;
}

int r3n_lights_write(r3n_ctx *c, const void *dir, uint64_t dir_bytes, const void *point, uint64_t point_bytes) {
    if (!c) return R3N_ERR_INVALID_ARG;
    // kept on the host; r3n_frame_begin uploads them into the frame's own copy (frames in flight: the previous frame's
    // resolve may still be reading its lights)
    auto take = [&](std::vector<uint8_t> &h, const void *src, uint64_t bytes, uint32_t stride, uint32_t cap) -> int {
        if (!src || bytes < 16) { h.assign(16, 0); return R3N_OK; }
        uint32_t count;
        std::memcpy(&count, src, 4);
        if ((uint64_t)count * stride + 16 > bytes) return fail(c, R3N_ERR_INVALID_ARG, "lights write: count exceeds buffer");
        if (count > cap) return fail(c, R3N_ERR_UNSUPPORTED, "lights write: more lights than the LDS light list holds");
        h.assign(static_cast<const uint8_t *>(src), static_cast<const uint8_t *>(src) + 16u + (size_t)count * stride);  // what the shader reads: count + array
        return R3N_OK;
    };
    const std::vector<uint8_t> old_dir = c->h_dir, old_point = c->h_point;
    TRY(take(c->h_dir, dir, dir_bytes, 128, R3N_MAX_DIR_LIGHTS));
    TRY(take(c->h_point, point, point_bytes, 32, R3N_MAX_POINT_LIGHTS));
    if (old_dir != c->h_dir || old_point != c->h_point) ++c->lights_version;  // unchanged lights: nothing to upload
    return R3N_OK;
}

// ------------------------------------------------------------------------------------------------ frame
// r3n_frame_begin; `d` (r3n_render_frame only) carries every camera header of the frame: they go up with the uniforms in one copy
static int frame_begin_impl(r3n_ctx *c, const r3n_frame_uniforms496 *u, uint32_t w, uint32_t h, uint32_t samples,
                            const float clear_color[4], uint32_t atlas_w, uint32_t atlas_h, const r3n_frame_desc *d) {
    if (!c || !u || !clear_color || !w || !h) return fail(c, R3N_ERR_INVALID_ARG, "frame_begin: bad args");
    if (samples != 1 && samples != 4) return fail(c, R3N_ERR_INVALID_ARG, "frame_begin: samples must be 1 or 4 (SampleCount::One | Four)");
    if (w > 65535 || h > 65535) return fail(c, R3N_ERR_UNSUPPORTED, "frame_begin: target larger than 65535");
    // the rasteriser addresses its targets with 32-bit byte offsets (8 B per sample key, 4 B per atlas texel)
    if ((uint64_t)w * h * samples >= (1ull << 29) || (uint64_t)atlas_w * atlas_h >= (1ull << 30) || atlas_w > 65535 || atlas_h > 65535)
        return fail(c, R3N_ERR_UNSUPPORTED, "frame_begin: target of 2^29 samples or more / shadow atlas of 2^30 texels or more");
    HIP_TRY(c, hipSetDevice(c->device));
    // (an overflow an earlier frame raised is reported by r3n_frame_end / r3n_sync / a read-back, never by refusing THIS frame:
    // the frame that overflowed has been presented already, the next one must still render)
    // (a resolution change keeps the temporal history, like the reference: CullingBufferMap is keyed by the camera alone,
    // culler.rs:53-80 -- last frame's predicted triangles are drawn into the new target, Hi-Z is built at the new size, the cull's
    // residual test reads last frame's bits.  Rounds 1-5 dropped the history here; tools/fuzz_parity.py --mutate found the
    // difference against the oracle, which keeps it.)
    c->width = w; c->height = h; c->samples = samples; c->atlas_w = atlas_w; c->atlas_h = atlas_h;
    std::memcpy(c->clear, clear_color, 16);
    // frames in flight: this frame renders into the other set of targets / per-frame inputs
    const int hs = (int)(c->frame_no % (uint64_t)r3n_ctx::kFbHost);
    c->slot = (int)(c->frame_no++ & 1u);
    std::swap(c->vis, c->alt_vis);
    std::swap(c->atlas, c->alt_atlas);
    std::swap(c->viewport.baked, c->alt_vp_baked);
    uint8_t *dev = c->fb_dev[c->slot].as<uint8_t>();
    c->fu.p = dev + r3n_ctx::kFbUniforms;
    c->dir_buf.p = dev + r3n_ctx::kFbDir;
    c->point_buf.p = dev + r3n_ctx::kFbPoint;
    c->viewport.d_hdr.p = dev + r3n_ctx::kFbViewportHdr; c->viewport.d_hdr.bytes = 256;
    for (auto &kv : c->shadows) { kv.second.d_hdr.p = dev + r3n_ctx::kFbShadowHdr + 256u * (size_t)kv.first; kv.second.d_hdr.bytes = 256; }
    if (c->shade_pending[c->slot]) {  // the resolve that last read this set (two frames ago)
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->shade_done[c->slot], 0));
        c->shade_pending[c->slot] = false;
    }
    // the frame's constants: uniforms + directional lights (they follow the camera: every frame) [+ every camera header], one copy
    if (c->fb_pending[hs]) {  // the copy that last read this pinned image (kFbHost frames ago)
        HIP_WAIT(c, hipEventSynchronize(c->fb_ev[hs]));
        c->fb_pending[hs] = false;
    }
    uint8_t *host = c->fb_host[hs];
    std::memcpy(host + r3n_ctx::kFbUniforms, u, sizeof *u);
    if (c->h_dir.size() < 16) c->h_dir.assign(16, 0);
    if (c->h_point.size() < 16) c->h_point.assign(16, 0);
    std::memcpy(host + r3n_ctx::kFbDir, c->h_dir.data(), c->h_dir.size());  // <= 16 + 128 * R3N_MAX_DIR_LIGHTS (r3n_lights_write)
    size_t upto = r3n_ctx::kFbShadowHdr;
    if (d) {
        std::memcpy(host + r3n_ctx::kFbViewportHdr, d->viewport_header, sizeof(r3n_camera_header240));
        for (uint32_t v = 0; v < d->n_shadow_views; ++v)
            std::memcpy(host + r3n_ctx::kFbShadowHdr + 256u * (size_t)v, &d->shadow_views[v].header, sizeof(r3n_camera_header240));
        upto += 256u * (size_t)d->n_shadow_views;
    }
    HIP_TRY(c, hipMemcpyAsync(dev, host, upto, hipMemcpyHostToDevice, c->stream));
    if (c->slot_lights_version[c->slot] != c->lights_version) {  // point lights: only when they changed
        std::memcpy(host + r3n_ctx::kFbPoint, c->h_point.data(), c->h_point.size());
        HIP_TRY(c, hipMemcpyAsync(dev + r3n_ctx::kFbPoint, host + r3n_ctx::kFbPoint, c->h_point.size(), hipMemcpyHostToDevice, c->stream));
        c->slot_lights_version[c->slot] = c->lights_version;
    }
    HIP_TRY(c, hipEventRecord(c->fb_ev[hs], c->stream));
    c->fb_pending[hs] = true;
    const size_t npix = (size_t)w * h;
    TRY(ensure(c, c->vis, npix * samples * 8, false, -1));
    TRY(ensure(c, c->hdr16, npix * 8, false, -1));
    TRY(ensure(c, c->out8, npix * 4, false, -1));
    build_hiz_desc(c->hizd, w, h);
    TRY(ensure(c, c->hiz, hiz_elements(c->hizd) * 4, false, -1));
    const size_t apix = (size_t)std::max(1u, atlas_w) * std::max(1u, atlas_h);
    TRY(ensure(c, c->atlas, apix * 4, false, -1));
    {
        Timed t(c, R3N_STAGE_CLEAR);
        // ONE launch: depth clear 0.0 / no triangle (base.rs:259-263), shadow atlas clear 0.0 (clear.rs:4-20), and the work-queue
        // counters of every r3n_forward of the frame, every lane (r3n_frame_end ordered the main stream behind the lanes' last use)
        hipLaunchKernelGGL(k_frame_clear, dim3(2048), dim3(256), 0, c->stream, c->vis.as<uint32_t>(), npix * samples * 2, c->atlas.as<uint32_t>(), apix,
                           c->big_count_all.as<uint32_t>(), c->big_count_all.bytes / 4);
        TRY(check_launch(c, "k_frame_clear"));
    }
    ++c->main_epoch;
    TRY(refresh_obj_meta(c));
    TRY(refresh_tri_base(c));
    c->in_frame = true;
    for (auto &f : c->forward_index_lane) f = 0;
    c->resolved_this_frame = false;
    c->blended_this_frame = false;
    c->hiz_plane_ready = false;
    c->viewport.culled = false;
    for (auto &kv : c->shadows) kv.second.culled = false;
    return R3N_OK;
}

int r3n_frame_begin(r3n_ctx *c, const r3n_frame_uniforms496 *u, uint32_t w, uint32_t h, uint32_t samples,
                    const float clear_color[4], uint32_t atlas_w, uint32_t atlas_h) {
    return frame_begin_impl(c, u, w, h, samples, clear_color, atlas_w, atlas_h, nullptr);
}

int r3n_skinning(r3n_ctx *c, const r3n_skinning_input40 *inputs, uint32_t n, const float *joint_matrices, uint32_t n_joints) {
    if (!c) return R3N_ERR_INVALID_ARG;
    if (n == 0) return R3N_OK;  // skinning.rs:216-218: nothing to do without skeletons
    if (!inputs || n_joints == 0) return fail(c, R3N_ERR_INVALID_ARG, "skinning: null inputs");
    if (!joint_matrices && c->n_pose_requests == 0) return fail(c, R3N_ERR_INVALID_ARG, "skinning: no joint matrices and no queued poses");
    if (c->pose_matrix_end > n_joints) return fail(c, R3N_ERR_INVALID_ARG, "skinning: a queued pose writes past the joint matrices");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(join_shade(c));  // frames in flight: the previous frame's resolve reads the skinned attribute runs this rewrites
    const size_t mesh_words = c->mesh.bytes / 4;
    for (uint32_t i = 0; i < n; ++i) {
        const r3n_skinning_input40 &in = inputs[i];
        if (in.joint_indices_offset == R3N_INVALID || in.joint_weight_offset == R3N_INVALID)
            return fail(c, R3N_ERR_INVALID_ARG, "skinning: skeleton without joint indices / weights (SkeletonCreationError)");
        // every attribute run the kernel touches: (byte offset, words per vertex); INVALID inputs read zeros, INVALID outputs are skipped
        const uint32_t runs[9][2] = {{in.base_position_offset, 3}, {in.base_normal_offset, 3}, {in.base_tangent_offset, 3},
                                     {in.joint_indices_offset, 2}, {in.joint_weight_offset, 4}, {in.updated_position_offset, 3},
                                     {in.updated_normal_offset, 3}, {in.updated_tangent_offset, 3}, {R3N_INVALID, 0}};
        for (const auto &run : runs) {
            if (run[0] == R3N_INVALID) continue;
            if ((run[0] & 3u) != 0u || (uint64_t)run[0] / 4 + (uint64_t)in.vertex_count * run[1] > mesh_words)
                return fail(c, R3N_ERR_INVALID_ARG, "skinning: attribute range outside the mesh buffer");
        }
        if (in.joint_matrix_base_offset >= n_joints) return fail(c, R3N_ERR_INVALID_ARG, "skinning: joint matrix base out of range");
    }
    // skeleton records change rarely: rebuild the wave -> skeleton map only when they do
    const bool same = c->h_skin_inputs.size() == n && std::memcmp(c->h_skin_inputs.data(), inputs, (size_t)n * sizeof *inputs) == 0;
    if (!same) {
        c->h_skin_inputs.assign(inputs, inputs + n);
        std::vector<uint32_t> wave_first(n + 1), wave_skel;
        uint32_t w = 0;
        for (uint32_t i = 0; i < n; ++i) {
            wave_first[i] = w;
            const uint32_t nw = (inputs[i].vertex_count + 63u) / 64u;
            wave_skel.insert(wave_skel.end(), nw, i);
            w += nw;
        }
        wave_first[n] = w;
        c->skin_total_waves = w;
        // joints per skeleton: the distance to the next larger matrix base (skinning.rs:96-118 lays the matrices out skeleton by skeleton)
        std::vector<uint32_t> bases(n), counts(n);
        for (uint32_t i = 0; i < n; ++i) bases[i] = inputs[i].joint_matrix_base_offset;
        std::vector<uint32_t> sorted_bases(bases);
        std::sort(sorted_bases.begin(), sorted_bases.end());
        c->skin_max_joints = 0;
        for (uint32_t i = 0; i < n; ++i) {
            auto it = std::upper_bound(sorted_bases.begin(), sorted_bases.end(), bases[i]);
            counts[i] = (it == sorted_bases.end() ? n_joints : *it) - bases[i];
            c->skin_max_joints = std::max(c->skin_max_joints, counts[i]);
        }
        TRY(ensure(c, c->skin_joint_counts, (size_t)n * 4, false, -1));
        HIP_TRY(c, hipMemcpyAsync(c->skin_joint_counts.p, counts.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
        TRY(ensure(c, c->skin_inputs, (size_t)n * sizeof *inputs, false, -1));
        TRY(ensure(c, c->skin_wave_first, (size_t)(n + 1) * 4, false, -1));
        TRY(ensure(c, c->skin_wave_skeleton, std::max<size_t>(w, 1) * 4, false, -1));
        HIP_TRY(c, hipMemcpyAsync(c->skin_inputs.p, inputs, (size_t)n * sizeof *inputs, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->skin_wave_first.p, wave_first.data(), (size_t)(n + 1) * 4, hipMemcpyHostToDevice, c->stream));
        if (w) HIP_TRY(c, hipMemcpyAsync(c->skin_wave_skeleton.p, wave_skel.data(), (size_t)w * 4, hipMemcpyHostToDevice, c->stream));
        HIP_WAIT(c, hipStreamSynchronize(c->stream));  // host vectors above are temporaries
    }
    TRY(ensure(c, c->skin_matrices, (size_t)n_joints * 64, true, -1));
    if (joint_matrices) TRY(upload_bulk(c, c->skin_matrices.p, joint_matrices, (size_t)n_joints * 64));  // pinned staging: no wait for the GPU
    if (c->n_pose_requests) {  // rend3-anim: poses queued by r3n_pose_skeletons overwrite their skeletons' matrices
        Timed t(c, R3N_STAGE_POSE);
        const int e = r3n_internal_pose_skeletons(c->pose_requests.p, c->n_pose_requests, c->anim_rigs.p, c->anim_joints.p, c->anim_clips.p,
                                                  c->anim_tracks.p, c->anim_times.as<float>(), c->anim_values.as<float>(),
                                                  c->skin_matrices.as<float>(), c->anim_max_joints, c->stream);
        c->n_pose_requests = 0;
        c->pose_matrix_end = 0;
        if (e != 0) return fail(c, R3N_ERR_HIP, std::string("k_pose_skeletons: ") + hipGetErrorString((hipError_t)e));
    }
    if (c->skin_total_waves == 0) return R3N_OK;
    ++c->main_epoch;  // the shadow lanes read the skinned attribute runs
    Timed t(c, R3N_STAGE_SKINNING);
    if (c->skinning_mode == R3N_SKIN_MFMA) {
        if (c->skin_max_joints > 4u) return fail(c, R3N_ERR_UNSUPPORTED, "skinning: R3N_SKIN_MFMA handles rigs of at most four joints (one 16 x 16 x 4 tile holds four joint matrices)");
        HIP_TRY(c, (hipError_t)r3n_internal_skinning_mfma(c->mesh.as<uint32_t>(), c->skin_inputs.p, c->skin_matrices.as<float>(), c->skin_wave_skeleton.as<uint32_t>(),
                                                          c->skin_wave_first.as<uint32_t>(), c->skin_joint_counts.as<uint32_t>(), c->skin_total_waves, c->stream));
        return R3N_OK;
    }
    hipLaunchKernelGGL(k_skinning, dim3((c->skin_total_waves + 3u) / 4u), dim3(256), 0, c->stream, c->mesh.as<uint32_t>(),
                       c->skin_inputs.as<r3n_skinning_input40>(), c->skin_matrices.as<float>(),
                       c->skin_wave_skeleton.as<uint32_t>(), c->skin_wave_first.as<uint32_t>(), c->skin_total_waves);
    return check_launch(c, "k_skinning");
}

int r3n_animation_write(r3n_ctx *c, const r3n_anim_rig16 *rigs, uint32_t n_rigs, const r3n_anim_joint80 *joints, uint32_t n_joints,
                        const r3n_anim_clip16 *clips, uint32_t n_clips, const r3n_anim_track80 *tracks, uint32_t n_tracks,
                        const float *times, uint32_t n_times, const float *values, uint32_t n_values) {
    if (!c || (n_rigs && (!rigs || !joints)) || (n_clips && (!clips || !tracks)) || (n_times && !times) || (n_values && !values))
        return fail(c, R3N_ERR_INVALID_ARG, "animation write: null");
    for (uint32_t i = 0; i < n_rigs; ++i) {
        const r3n_anim_rig16 &r = rigs[i];
        if (r.n_joints == 0 || r.n_joints > 512u) return fail(c, R3N_ERR_CAPACITY, "animation write: a rig has 0 or more than 512 joints");
        if ((uint64_t)r.first_joint + r.n_joints > n_joints) return fail(c, R3N_ERR_INVALID_ARG, "animation write: rig joints out of range");
        uint32_t max_depth = 0;
        for (uint32_t j = 0; j < r.n_joints; ++j) {
            const r3n_anim_joint80 &jt = joints[r.first_joint + j];
            if (jt.parent < -2 || jt.parent >= (int32_t)r.n_joints) return fail(c, R3N_ERR_INVALID_ARG, "animation write: parent joint out of range");
            const uint32_t want = jt.parent < 0 ? 0u : joints[r.first_joint + jt.parent].depth + 1u;
            if (jt.depth != want) return fail(c, R3N_ERR_INVALID_ARG, "animation write: joint depth does not match its parent's");
            max_depth = std::max(max_depth, jt.depth);
        }
        if (max_depth != r.max_depth) return fail(c, R3N_ERR_INVALID_ARG, "animation write: rig max_depth does not match its joints");
    }
    for (uint32_t i = 0; i < n_clips; ++i) {
        const r3n_anim_clip16 &cl = clips[i];
        if (cl.rig >= n_rigs) return fail(c, R3N_ERR_INVALID_ARG, "animation write: clip rig out of range");
        if ((uint64_t)cl.first_track + rigs[cl.rig].n_joints > n_tracks) return fail(c, R3N_ERR_INVALID_ARG, "animation write: clip tracks out of range");
        for (uint32_t j = 0; j < rigs[cl.rig].n_joints; ++j) {
            const r3n_anim_track80 &t = tracks[cl.first_track + j];
            for (int k = 0; k < 3; ++k) {
                const uint32_t width = k == 1 ? 4u : 3u;
                if ((uint64_t)t.key_first[k] + t.key_count[k] > n_times || (uint64_t)t.value_first[k] + (uint64_t)t.key_count[k] * width > n_values)
                    return fail(c, R3N_ERR_INVALID_ARG, "animation write: channel keys out of range");
            }
        }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(sync_all(c));
    auto put = [&](DevBuf &b, const void *src, size_t bytes) -> int {
        TRY(ensure(c, b, std::max<size_t>(bytes, 16), false, -1));
        if (bytes) HIP_TRY(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
        return R3N_OK;
    };
    TRY(put(c->anim_rigs, rigs, (size_t)n_rigs * sizeof *rigs));
    TRY(put(c->anim_joints, joints, (size_t)n_joints * sizeof *joints));
    TRY(put(c->anim_clips, clips, (size_t)n_clips * sizeof *clips));
    // device copy of the tracks: bit 0 of `animated` = animated, bits 1..3 = "the translation / rotation / scale key times
    // are non-decreasing", which lets the kernel bisect instead of scan (same result: the first key later than t)
    std::vector<r3n_anim_track80> dev_tracks(tracks, tracks + n_tracks);
    for (auto &t : dev_tracks) {
        uint32_t flags = t.animated ? 1u : 0u;
        for (int k = 0; k < 3; ++k) {
            bool sorted = true;
            for (uint32_t i = 1; i < t.key_count[k] && sorted; ++i) sorted = times[t.key_first[k] + i - 1u] <= times[t.key_first[k] + i];
            if (sorted) flags |= 2u << k;
        }
        t.animated = flags;
    }
    TRY(put(c->anim_tracks, dev_tracks.data(), (size_t)n_tracks * sizeof *tracks));
    TRY(put(c->anim_times, times, (size_t)n_times * 4));
    TRY(put(c->anim_values, values, (size_t)n_values * 4));
    HIP_WAIT(c, hipStreamSynchronize(c->stream));  // the caller owns the sources only for the duration of the call
    c->h_anim_rigs.assign(rigs, rigs + n_rigs);
    c->anim_max_joints = 1;
    for (uint32_t i = 0; i < n_rigs; ++i) c->anim_max_joints = std::max(c->anim_max_joints, rigs[i].n_joints);
    c->h_anim_clips.assign(clips, clips + n_clips);
    c->n_pose_requests = 0;
    c->pose_matrix_end = 0;
    return R3N_OK;
}

int r3n_pose_skeletons(r3n_ctx *c, const r3n_pose_request16 *requests, uint32_t n) {
    if (!c || (n && !requests)) return fail(c, R3N_ERR_INVALID_ARG, "pose skeletons: null");
    uint32_t end = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (requests[i].clip >= c->h_anim_clips.size()) return fail(c, R3N_ERR_INVALID_ARG, "pose skeletons: clip out of range");
        const uint64_t e = (uint64_t)requests[i].matrix_base + c->h_anim_rigs[c->h_anim_clips[requests[i].clip].rig].n_joints;
        if (e > 0xFFFFFFFFull) return fail(c, R3N_ERR_INVALID_ARG, "pose skeletons: matrix base out of range");
        end = std::max(end, (uint32_t)e);
    }
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(ensure(c, c->pose_requests, std::max<size_t>(n, 1) * sizeof *requests, false, -1));
    if (n) {
        TRY(upload_bulk(c, c->pose_requests.p, requests, (size_t)n * sizeof *requests));  // pinned staging: no wait for the GPU
    }
    c->n_pose_requests = n;
    c->pose_matrix_end = end;
    return R3N_OK;
}


int r3n_uniform_bake(r3n_ctx *c, r3n_camera cam, const r3n_camera_header240 *hdr) {
    if (!c || !hdr) return fail(c, R3N_ERR_INVALID_ARG, "uniform_bake: null");
    CamState *s = find_cam(c, cam, true);
    if (!s) return fail(c, R3N_ERR_INVALID_ARG, "uniform_bake: bad camera");
    if (hdr->object_count != c->capacity) return fail(c, R3N_ERR_INVALID_ARG, "uniform_bake: header.object_count != object capacity");
    if ((cam == R3N_CAMERA_VIEWPORT) != (hdr->shadow_index == R3N_INVALID) || (cam != R3N_CAMERA_VIEWPORT && hdr->shadow_index != cam))
        return fail(c, R3N_ERR_INVALID_ARG, "uniform_bake: header.shadow_index does not match camera");
    HIP_TRY(c, hipSetDevice(c->device));
    s->hdr = *hdr;
    s->has_hdr = true;
    s->baked_frame = c->frame_no;
    if (c->capacity == 0) return R3N_OK;  // culler.rs:449-451
    TRY(refresh_obj_meta(c));
    // the camera's header lives in the frame-constants block of the current frame slot
    s->d_hdr.p = c->fb_dev[c->slot].as<uint8_t>() + (cam == R3N_CAMERA_VIEWPORT ? r3n_ctx::kFbViewportHdr : r3n_ctx::kFbShadowHdr + 256u * (size_t)cam);
    s->d_hdr.bytes = 256;
    if (s->hdr_frame != c->frame_no) TRY(upload_small(c, s->d_hdr.p, hdr, sizeof *hdr));  // else: it went up with the frame's constants
    // per-camera buffer regrow preserves old matrices (disabled slots keep stale data, App. D.2)
    TRY(ensure(c, s->baked, (size_t)c->capacity * sizeof(r3n_baked128), true, 0));
    const int lane = cam_lane(c, cam);
    if (c->fused_frame && c->in_frame && chained_pass_fits(c)) {
        // r3n_render_frame: this camera's r3n_cull follows in the same frame with the same header and ranges, and its object pass
        // (frustum test + slot assignment) depends on nothing the frame computes in between -- bake and object pass go out as ONE
        // launch now; r3n_cull then issues the triangle cull alone (for the viewport that takes three launches off the serial
        // chain between Hi-Z and the triangle cull)
        TRY(ensure(c, s->chain, (size_t)R3N_CHAINED_OBJECT_PASS_MAX_BLOCKS * sizeof(ObjChainRec), false, 0));
        TRY(ensure(c, s->vis_flags, c->capacity, true, 0));
        TRY(ensure(c, s->vis_list, (size_t)(c->capacity + 1u) * sizeof(r3n_vis_entry), false, -1));
        TRY(ensure(c, s->slot_base[s->cur], (size_t)c->capacity * 4u, true, 0xFF));
        TRY(ensure(c, s->sub_counts[s->cur], sizeof(r3n_sub_counts), false, 0));
        TRY(ensure(c, s->counts[s->cur], sizeof(r3n_cull_counts), false, 0));
        TRY(ensure(c, s->first_entry, first_entry_bytes(c), false, -1));
        TRY(fork_lane(c, lane));
        TRY(run_bake_and_object_pass(c, *s, s->cur, camera_own(c, *s), lane_stream(c, lane), cam == R3N_CAMERA_VIEWPORT));
        s->object_pass_frame = c->frame_no;
        return R3N_OK;
    }
    TRY(fork_lane(c, lane));  // after the header upload (main stream)
    hipStream_t stream = lane_stream(c, lane);
    Timed t(c, R3N_STAGE_BAKE, stream);
    const uint32_t threads = c->capacity * 4u;
    hipLaunchKernelGGL(k_uniform_bake, dim3((threads + 255u) / 256u), dim3(256), 0, stream,
                       s->d_hdr.as<r3n_camera_header240>(), c->objects.as<r3n_object128>(), s->baked.as<r3n_baked128>());
    return check_launch(c, "k_uniform_bake");
}

static void key_census(r3n_ctx *c) {
    if (!c->key_census_dirty) return;
    c->key_objects[0] = c->key_objects[1] = c->key_objects[2] = 0;
    for (uint32_t o = 0; o < c->capacity; ++o)
        if (c->h_ntri[o]) {
            const uint32_t mi = c->h_material[o];
            c->key_objects[mi < c->h_material_key.size() ? std::min<uint32_t>(c->h_material_key[mi], 2) : 0]++;
        }
    c->key_census_dirty = false;
}

int r3n_cull(r3n_ctx *c, r3n_camera cam) {
    if (!c) return R3N_ERR_INVALID_ARG;
    CamState *s = find_cam(c, cam, false);
    if (!s || !s->has_hdr) return fail(c, R3N_ERR_STATE, "cull: camera has no baked uniforms (culler.rs:572 panics here)");
    if (!c->in_frame) return fail(c, R3N_ERR_STATE, "cull: outside frame_begin/frame_end");
    if (c->capacity == 0) return R3N_OK;  // culler.rs:705-707
    // the camera's header lives in the frame's constants block, which alternates between two sets: without a bake in THIS frame the
    // kernels would read the header of two frames ago (the reference uploads it in front of every cull, base.rs:148-156)
    if (s->baked_frame != c->frame_no) return fail(c, R3N_ERR_STATE, "cull: r3n_uniform_bake has not run for this camera in this frame");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool viewport = cam == R3N_CAMERA_VIEWPORT;
    const int cur = s->cur, prev = 1 - cur;
    key_census(c);
    for (int k = 0; k < 3; ++k) s->key_objects[cur][k] = c->key_objects[k];
    const int lane = cam_lane(c, cam);
    hipStream_t stream = lane_stream(c, lane);
    // (re)allocations below run on the main stream: size everything first, then fork
    {
        const uint32_t cap0 = c->capacity, nblocks0 = (cap0 + 255u) / 256u;
        TRY(ensure(c, s->vis_flags, cap0, true, 0));
        TRY(ensure(c, s->vis_list, (size_t)(cap0 + 1u) * sizeof(r3n_vis_entry), false, -1));
        TRY(ensure(c, s->block_sums, (size_t)nblocks0 * sizeof(ObjBlockSums), false, -1));
        TRY(ensure(c, s->block_off, (size_t)nblocks0 * sizeof(ObjBlockOffsets), false, -1));
        TRY(ensure(c, s->slot_base[cur], (size_t)cap0 * 4u, true, 0xFF));
        // LAST frame's table too: the triangle cull reads prev_slot_base[object] for every object of THIS frame's work list, and an
        // object added since the last frame may sit beyond last frame's capacity (the buffer doubled: handle = old capacity).  The
        // new entries read INVALID = "not batched last frame" (batching.rs:226): all of its passing triangles are residual.
        // (Found by tools/fuzz_parity.py: the read ran past the allocation, the object's first frame was drawn by neither pass.)
        if (viewport && s->has_prev) TRY(ensure(c, s->slot_base[prev], (size_t)cap0 * 4u, true, 0xFF));
        TRY(ensure(c, s->sub_counts[cur], sizeof(r3n_sub_counts), false, 0));
        TRY(ensure(c, s->counts[cur], sizeof(r3n_cull_counts), false, 0));
        TRY(ensure(c, s->first_entry, first_entry_bytes(c), false, -1));
    }
    const uint32_t mw = max_waves(c);
    const uint32_t chunks = (mw + R3N_CHUNK_WAVES - 1u) / R3N_CHUNK_WAVES;
    // every chunk appends to sub-list (chunk % R3N_SUBQ); worst case all of its slots pass and share one key
    const uint32_t subcap = ((chunks + R3N_SUBQ - 1u) / R3N_SUBQ) * (R3N_CHUNK_WAVES * 64u);
    const size_t list_bytes = (size_t)3 * R3N_SUBQ * subcap * sizeof(r3n_tri_ref);
    s->subcap[cur] = subcap;
    TRY(ensure(c, s->mask[cur], (size_t)mw * 8u, false, -1));
    TRY(ensure(c, s->predicted[cur], list_bytes, false, -1));
    if (viewport) TRY(ensure(c, s->residual, list_bytes, false, -1));
    TRY(fork_lane(c, lane));
    if (s->object_pass_frame != c->frame_no)  // else: issued with the bake (r3n_render_frame)
        TRY(run_object_pass(c, *s, cur, camera_own(c, *s), nullptr, stream));
    TriCullArgs a{};
    a.hdr = s->d_hdr.as<r3n_camera_header240>();
    a.objects = c->objects.as<r3n_object128>();
    a.mesh = c->mesh.as<uint32_t>();
    a.baked = s->baked.as<r3n_baked128>();
    a.material_keys = c->material_keys.as<uint8_t>();
    a.n_materials = c->n_materials;
    a.vis_list = s->vis_list.as<r3n_vis_entry>();
    a.first_entry = s->first_entry.as<uint32_t>();
    a.counts = s->counts[cur].as<r3n_cull_counts>();
    a.prev_slot_base = (viewport && s->has_prev) ? s->slot_base[prev].as<uint32_t>() : nullptr;
    a.prev_mask = (viewport && s->has_prev) ? s->mask[prev].as<unsigned long long>() : nullptr;
    a.mask = s->mask[cur].as<unsigned long long>();
    a.predicted = s->predicted[cur].as<r3n_tri_ref>();
    a.residual = viewport ? s->residual.as<r3n_tri_ref>() : nullptr;
    a.sub_counts = s->sub_counts[cur].as<r3n_sub_counts>();
    a.subcap = subcap;
    a.hiz.data = c->hiz.as<float>();
    a.hiz.d = c->hizd;
    const uint32_t grid = std::max(1u, std::min(chunks, 4096u));
    {
        Timed t(c, R3N_STAGE_TRIANGLE_CULL, stream);
        hipLaunchKernelGGL(k_triangle_cull, dim3(grid), dim3(256), viewport ? c->tune.vp_cull_lds : c->tune.cull_lds, stream, a);
    }
    TRY(check_launch(c, "k_triangle_cull"));
    s->culled = true;
    s->last = cur;
    return R3N_OK;
}

int r3n_hi_z(r3n_ctx *c) {
    if (!c || !c->in_frame) return fail(c, R3N_ERR_STATE, "hi_z: outside a frame");
    HIP_TRY(c, hipSetDevice(c->device));
    Timed t(c, R3N_STAGE_HIZ);
    // head: mip0 + as many levels as have even source dimensions (<= 4), one launch; tail: one single-block launch
    uint32_t levels = 0;
    while (levels < 4u && levels + 1u < c->hizd.mips && ((c->width >> levels) % 2u == 0u) && ((c->height >> levels) % 2u == 0u) &&
           (c->width >> levels) >= 2u && (c->height >> levels) >= 2u)
        ++levels;
    hipLaunchKernelGGL(k_hiz_head, dim3((c->width + 31u) / 32u, (c->height + 31u) / 32u), dim3(256), 0, c->stream,
                       c->vis.as<unsigned long long>(), c->hiz.as<float>(), c->hizd, levels, c->hiz_plane_ready ? 0u : c->samples);
    c->hiz_plane_ready = false;
    uint32_t first = levels + 1u;
    if (first < c->hizd.mips) {
        // the first level behind the head is still thousands of texels (120 x 67 at 4K): a grid of workgroups builds it, so the
        // single-workgroup tail -- which shares its CU with whatever else is resident -- starts from a level a quarter the size
        const uint32_t dw = std::max(1u, c->width >> first), dh = std::max(1u, c->height >> first);
        if ((size_t)dw * dh > 2304u) {
            const uint32_t sw = std::max(1u, c->width >> (first - 1u)), sh = std::max(1u, c->height >> (first - 1u));
            hipLaunchKernelGGL(k_hiz_downsample, dim3((dw + 15u) / 16u, (dh + 15u) / 16u), dim3(256), 0, c->stream, c->hiz.as<float>() + c->hizd.offset[first - 1u],
                               c->hiz.as<float>() + c->hizd.offset[first], sw, sh, dw, dh);
            ++first;
        }
    }
    if (first < c->hizd.mips) {
        const uint32_t n0 = std::max(1u, c->width >> first) * std::max(1u, c->height >> first);
        hipLaunchKernelGGL(k_hiz_tail, dim3(1), dim3(n0 <= 2304u ? 256 : 1024), 0, c->stream, c->hiz.as<float>(), c->hizd, first);
    }
    return check_launch(c, "hi_z");
}

int r3n_shadow_viewport(r3n_ctx *c, r3n_camera cam, uint32_t x, uint32_t y, uint32_t size) {
    if (!c || cam == R3N_CAMERA_VIEWPORT) return fail(c, R3N_ERR_INVALID_ARG, "shadow_viewport: needs a shadow camera");
    CamState *s = find_cam(c, cam, true);
    if (!s) return fail(c, R3N_ERR_INVALID_ARG, "shadow_viewport: bad camera");
    s->vp_x = x; s->vp_y = y; s->vp_size = size;
    return R3N_OK;
}

static int forward_blend(r3n_ctx *c);

// Every albedo map a CUTOUT-key material samples for its alpha test (opaque.wgsl:231-235, depth.wgsl:100-127) is on the sampler's
// short path -- power-of-two extents, RGBA8 pool texels, a pool below 2^30 texels, the linear sampler: the rasterisers' textured
// instantiations then carry no general sampler (kernels_raster.h SHORTA: most of their vector registers).  Host mirrors only.
static bool cutout_alpha_short(r3n_ctx *c) {
    if (!c->cutout_short_dirty) return c->cutout_short;
    bool ok = c->n_texels <= (1ull << 30);
    for (uint32_t i = 0; ok && i < c->n_materials && i < c->h_materials.size(); ++i) {
        const r3n_material208 &m = c->h_materials[i];
        if (i >= c->h_material_key.size() || c->h_material_key[i] != R3N_KEY_CUTOUT) continue;
        if (!(m.flags & R3N_FLAGS_ALBEDO_ACTIVE) || m.textures[0] == 0u) continue;
        const uint32_t id = m.textures[0];
        ok = id <= c->n_textures && id <= c->h_tex_short.size() && c->h_tex_short[id - 1u] != 0 && !(m.flags & R3N_FLAGS_NEAREST);
    }
    c->cutout_short = ok;
    c->cutout_short_dirty = false;
    return ok;
}

int r3n_forward(r3n_ctx *c, r3n_camera cam, uint32_t pass, uint32_t source, uint32_t key) {
    if (!c || pass > R3N_PASS_FORWARD || source > R3N_SOURCE_RESIDUAL || key > R3N_KEY_BLEND)
        return fail(c, R3N_ERR_INVALID_ARG, "forward: bad args");
    if (!c->in_frame) return fail(c, R3N_ERR_STATE, "forward: outside a frame");
    CamState *s = find_cam(c, cam, false);
    if (!s || !s->has_hdr || c->capacity == 0) return R3N_OK;  // nothing baked / culled yet: forward.rs:214-242
    key_census(c);
    // no object carried this material key when the draw ranges were made: the range is empty ("no draw calls for this material",
    // forward.rs:285-288) -- known on the host, so no launch at all.  Last frame's ranges for the predicted source: a material
    // whose key Renderer::update_material changed since (renderer/mod.rs:256-266) still has its triangles in the OLD key's range
    if ((source == R3N_SOURCE_PREDICTED ? s->key_objects[1 - s->cur][key] : c->key_objects[key]) == 0) return R3N_OK;
    const bool viewport = cam == R3N_CAMERA_VIEWPORT;
    if ((pass == R3N_PASS_FORWARD) != viewport) return fail(c, R3N_ERR_UNSUPPORTED, "forward: FORWARD needs the viewport, DEPTH a shadow camera");
    if (key == R3N_KEY_BLEND) {
        // shadow views have no blend routine (base.rs:366-396 draws opaque_depth and cutout_depth only)
        if (!viewport || source != R3N_SOURCE_RESIDUAL) return fail(c, R3N_ERR_INVALID_ARG, "forward: the blend routine draws the viewport's residual source (base.rs:451-465)");
        HIP_TRY(c, hipSetDevice(c->device));
        return forward_blend(c);
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int idx;
    const r3n_tri_ref *list;
    const uint32_t *sub_counts;
    if (source == R3N_SOURCE_PREDICTED) {
        // last frame's predicted triangles, this frame's matrices (forward.rs:224-232, App. D.8)
        if (!s->has_prev) return R3N_OK;
        idx = 1 - s->cur;
        list = s->predicted[idx].as<r3n_tri_ref>();
        sub_counts = &s->sub_counts[idx].as<r3n_sub_counts>()->n[0][0][0];
    } else {
        if (!s->culled) return R3N_OK;
        idx = s->cur;
        // residual && viewport -> residual list; otherwise the list written this frame (forward.rs:234,251)
        list = viewport ? s->residual.as<r3n_tri_ref>() : s->predicted[idx].as<r3n_tri_ref>();
        sub_counts = &s->sub_counts[idx].as<r3n_sub_counts>()->n[viewport ? 1 : 0][0][0];
    }
    RasterArgs a{};
    a.hdr = s->d_hdr.as<r3n_camera_header240>();
    a.objects = c->objects.as<r3n_object128>();
    a.mesh = c->mesh.as<uint32_t>();
    a.baked = s->baked.as<r3n_baked128>();
    a.materials = c->materials.as<r3n_material208>();
    a.material_keys = c->material_keys.as<uint8_t>();
    a.n_materials = c->n_materials;
    a.tri_base = c->tri_base.as<uint32_t>();
    a.list = list;
    a.sub_counts = sub_counts;
    a.subcap = s->subcap[idx];
    a.key = key;
    const int lane = cam_lane(c, cam);
    hipStream_t stream = lane_stream(c, lane);
    a.big_items = c->big_items[lane].as<r3n_big_item>();
    const uint32_t fwd = std::min(c->forward_index_lane[lane]++, 63u);
    a.big_count = c->big_count[lane].as<uint32_t>() + (size_t)fwd * R3N_BIGQ;
    a.big_capacity = c->big_capacity;
    a.big_uv = c->big_uv[lane].as<r3n_big_uv>();
    a.tex = texture_args(c);
    const uint32_t small_grid = c->tune.small_grid;  // multiple of R3N_SUBQ and R3N_BIGQ: 64 blocks per sub-list
    TRY(fork_lane(c, lane));
    if (fwd == 63u) HIP_TRY(c, hipMemsetAsync(a.big_count, 0, R3N_BIGQ * 4, stream));  // the first 63 calls of a lane have counters zeroed at frame begin
    a.row_begin = 0; a.row_end = 0xFFFFFFFFu;
    if (viewport) {
        a.vp_x = 0; a.vp_y = 0; a.vp_w = c->width; a.vp_h = c->height; a.target_pitch = c->width;
        a.vis = c->vis.as<unsigned long long>();
        if (c->shard_rows) { a.row_begin = std::min(c->row_begin, c->height); a.row_end = std::min(c->row_end, c->height); }
        // textured variant only where it can matter: cutout key and a non-empty texture array
        const bool tex = key == R3N_KEY_CUTOUT && c->n_textures > 0;
        const bool nocut = key != R3N_KEY_CUTOUT;  // the opaque key's instantiations carry nothing of the cutout test (kernels_raster.h NOCUT)
        auto launch = [&](auto small, auto big) {
            { Timed t(c, nocut ? R3N_STAGE_RASTER : R3N_STAGE_RASTER_CUT, stream); hipLaunchKernelGGL(small, dim3(small_grid), dim3(256), nocut ? c->tune.vp_small_lds : c->tune.vp_cut_small_lds, stream, a); }
            { Timed t(c, nocut ? R3N_STAGE_RASTER_BIG : R3N_STAGE_RASTER_BIG_CUT, stream); hipLaunchKernelGGL(big, dim3(c->tune.big_grid), dim3(256), nocut ? c->tune.vp_big_lds : c->tune.vp_cut_big_lds, stream, a); }
        };
        const bool shorta = tex && cutout_alpha_short(c);
        if (c->samples == 4) {
            if (shorta) launch(k_raster_small<false, 4, true, false, true>, k_raster_big<false, 4, true, false, false, true>);
            else if (tex) launch(k_raster_small<false, 4, true>, k_raster_big<false, 4, true>);
            else if (nocut) launch(k_raster_small<false, 4, false, true>, k_raster_big<false, 4, false, false, true>);
            else launch(k_raster_small<false, 4, false>, k_raster_big<false, 4, false>);
        } else {
            if (shorta) launch(k_raster_small<false, 1, true, false, true>, k_raster_big<false, 1, true, false, false, true>);
            else if (tex) launch(k_raster_small<false, 1, true>, k_raster_big<false, 1, true>);
            else if (nocut) launch(k_raster_small<false, 1, false, true>, k_raster_big<false, 1, false, false, true>);
            else launch(k_raster_small<false, 1, false>, k_raster_big<false, 1, false>);
        }
    } else {
        if (s->vp_size == 0 || s->vp_x + s->vp_size > c->atlas_w || s->vp_y + s->vp_size > c->atlas_h)
            return fail(c, R3N_ERR_STATE, "forward: shadow viewport not set or outside the atlas");
        a.vp_x = s->vp_x; a.vp_y = s->vp_y; a.vp_w = s->vp_size; a.vp_h = s->vp_size; a.target_pitch = c->atlas_w;
        // whole view unless split over ranks -- and the split is r3n_render_frame's (it also issues the exchange that fills in the other
        // bands): a caller of the per-node API after a split frame draws the whole view, not a stale band
        if (c->fused_frame) { a.row_begin = std::min(s->band_begin, s->vp_size); a.row_end = std::min(s->band_end, s->vp_size); }
        a.depth = c->atlas.as<uint32_t>();
        const bool tex = key == R3N_KEY_CUTOUT && c->n_textures > 0;
        if (tex && cutout_alpha_short(c)) {
            { Timed t(c, R3N_STAGE_SHADOW_RASTER, stream); hipLaunchKernelGGL((k_raster_small<true, 1, true, false, true>), dim3(small_grid), dim3(256), c->tune.cut_small_lds, stream, a); }
            { Timed t(c, R3N_STAGE_SHADOW_RASTER_BIG, stream); hipLaunchKernelGGL((k_raster_big<true, 1, true, false, false, true>), dim3(c->tune.big_grid), dim3(256), c->tune.cut_big_lds, stream, a); }
        } else if (tex) {
            { Timed t(c, R3N_STAGE_SHADOW_RASTER, stream); hipLaunchKernelGGL((k_raster_small<true, 1, true>), dim3(small_grid), dim3(256), c->tune.cut_small_lds, stream, a); }
            { Timed t(c, R3N_STAGE_SHADOW_RASTER_BIG, stream); hipLaunchKernelGGL((k_raster_big<true, 1, true>), dim3(c->tune.big_grid), dim3(256), c->tune.cut_big_lds, stream, a); }
        } else if (key != R3N_KEY_CUTOUT) {
            { Timed t(c, R3N_STAGE_SHADOW_RASTER, stream); hipLaunchKernelGGL((k_raster_small<true, 1, false, true>), dim3(small_grid), dim3(256), c->tune.small_lds, stream, a); }
            { Timed t(c, R3N_STAGE_SHADOW_RASTER_BIG, stream); hipLaunchKernelGGL((k_raster_big<true, 1, false, false, true>), dim3(c->tune.big_grid), dim3(256), c->tune.big_lds, stream, a); }
        } else {
            { Timed t(c, R3N_STAGE_SHADOW_RASTER, stream); hipLaunchKernelGGL((k_raster_small<true, 1, false>), dim3(small_grid), dim3(256), c->tune.cut_small_lds, stream, a); }
            { Timed t(c, R3N_STAGE_SHADOW_RASTER_BIG, stream); hipLaunchKernelGGL((k_raster_big<true, 1, false>), dim3(c->tune.big_grid), dim3(256), c->tune.cut_big_lds, stream, a); }
        }
    }
    return check_launch(c, "raster");
}

// Census of the material classes (kernels_shade.h): the feature word of every material slot goes to the device, and the
// variants of the resolve that some material maps to are the ones the frame launches.  A material that binds a texture the
// sampler's short path does not cover (extent not a power of two, float pool texels, pool beyond 2^30 texels, id out of range)
// carries R3N_FEAT_TEX_GENERAL and thereby goes to the general kernel.  Runs when materials or textures changed.
static int refresh_material_classes(r3n_ctx *c) {
    if (!c->classes_dirty) return R3N_OK;
    const uint32_t n = c->n_materials;
    c->h_materials.resize(n, r3n_material208{});
    std::vector<uint32_t> feat(std::max(n, 1u), R3N_FEAT_ALL);
    const bool small_pool = c->n_texels <= (1ull << 30);
    uint32_t variants = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const r3n_material208 &m = c->h_materials[i];
        uint32_t f = r3n_material_features(m);
        for (int k = 0; k < 10; ++k) {
            const uint32_t id = m.textures[k];
            if (id && (!small_pool || id > c->n_textures || id > c->h_tex_short.size() || !c->h_tex_short[id - 1u])) f |= R3N_FEAT_TEX_GENERAL;
        }
        feat[i] = f;
        variants |= 1u << r3n_variant_of(f);
    }
    TRY(ensure(c, c->material_feat, feat.size() * 4, false, -1));
    HIP_TRY(c, hipMemcpyAsync(c->material_feat.p, feat.data(), feat.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIP_WAIT(c, hipStreamSynchronize(c->stream));  // `feat` is a temporary
    c->resolve_variants = variants;
    c->classes_dirty = false;
    return R3N_OK;
}

static ShadeArgs make_shade_args(r3n_ctx *c, uint32_t r0, uint32_t r1) {
    CamState &s = c->viewport;
    ShadeArgs a{};
    a.vis = c->vis.as<unsigned long long>();
    a.width = c->width; a.height = c->height; a.row_begin = r0; a.row_end = r1;
    a.fu = c->fu.as<r3n_frame_uniforms496>();
    a.hdr = s.d_hdr.as<r3n_camera_header240>();
    a.objects = c->objects.as<r3n_object128>();
    a.mesh = c->mesh.as<uint32_t>();
    a.baked = s.baked.as<r3n_baked128>();
    a.materials = c->materials.as<r3n_material208>();
    a.n_materials = c->n_materials;
    a.tri_base = c->tri_base.as<uint32_t>();
    a.slot_table = c->slot_table.as<uint32_t>();
    a.slot_table_size = c->slot_table_size;
    a.dir_buf = c->dir_buf.as<uint8_t>();
    a.point_buf = c->point_buf.as<uint8_t>();
    a.atlas = c->atlas.as<float>();
    a.atlas_w = std::max(1u, c->atlas_w); a.atlas_h = std::max(1u, c->atlas_h);
    std::memcpy(a.clear, c->clear, 16);
    a.hdr_out = c->hdr16.as<ushort4>();
    a.ldr_out = c->out8.as<uchar4>();
    a.srgb_lut = c->srgb_lut.as<unsigned char>();
    a.out_bgr = (c->output_format & 1u) != 0u;
    a.tex = texture_args(c);
    a.samples_out = nullptr;
    a.tri_rec = nullptr;
    a.seen = nullptr;
    a.resolve_lds = c->tune.resolve_lds;  // (the key was parsed and never reached the launches)
    a.total_tris = (uint32_t)c->total_tris;
    a.view_lights = nullptr;
    a.material_feat = nullptr;
    a.variants = 0u;
    return a;
}

int r3n_resolve_opaque(r3n_ctx *c) {
    if (!c || !c->in_frame) return fail(c, R3N_ERR_STATE, "resolve_opaque: outside a frame");
    HIP_TRY(c, hipSetDevice(c->device));
    const uint32_t r0 = std::min(c->row_begin, c->height), r1 = std::min(c->row_end, c->height);
    if (r1 <= r0) return R3N_OK;
    CamState &s = c->viewport;
    if (!s.has_hdr && c->capacity) return fail(c, R3N_ERR_STATE, "resolve_opaque: viewport uniforms not baked");
    // (Re)allocations FIRST: ensure() fills and copies on the MAIN stream, and the event that orders the shade stream behind the
    // main stream is recorded below -- a buffer zeroed after that event would race with the resolve's kernels (first frame /
    // after the world grew: k_mark_visible's flags wiped by the late fill).
    const bool use_records = c->total_tris > 0 && (uint64_t)c->total_tris * sizeof(TriRecord) <= (8ull << 30);
    const uint64_t npix_all = (uint64_t)c->width * c->height;
    const bool blend_samples = c->samples == 4 && c->blend_tris > 0;  // a transparent pass will blend into the individual samples
    const bool split = c->samples == 4 && use_records && !blend_samples && npix_all < (1ull << 29);  // MSAA resolve in three passes (first triangle per pixel / queued edge triangles / edge pixel average)
    uint32_t edge_cap = 0;
    // single-sample record-based resolve of a textured world: one kernel per material class present (R3N_RESOLVE_CLASSES=0: the general kernel)
    const bool classes = use_records && c->samples == 1 && c->n_textures > 0 && c->n_materials > 0 && c->resolve_classes;
    if (classes) TRY(refresh_material_classes(c));
    if (use_records && c->samples == 1) TRY(ensure(c, c->view_lights[c->slot], sizeof(ViewLights), false, -1));
    if (use_records) {
        TRY(ensure(c, c->tri_rec, (size_t)c->total_tris * sizeof(TriRecord), false, -1));
        TRY(ensure(c, c->tri_seen, (size_t)c->total_tris, false, 0));  // zero: k_vertex_stage returns every flag it consumes to 0
    }
    if (blend_samples || split) TRY(ensure(c, c->samples16, (size_t)npix_all * 4 * 8, false, -1));
    if (split) {
        edge_cap = (uint32_t)((uint64_t)(r1 - r0) * c->width * 3u / R3N_EDGEQ) + 4096u;
        if (c->edge_capacity_override) edge_cap = c->edge_capacity_override;  // R3N_EDGE_CAPACITY: exercises the overflow path in tests
        TRY(ensure(c, c->edge_list, (size_t)edge_cap * R3N_EDGEQ * 4, false, -1));
        TRY(ensure(c, c->edge_count, R3N_EDGEQ * 4, false, 0));
    }
    // frames in flight: the resolve goes to the shade stream, ordered after this frame's viewport chain (main stream up
    // to here) and shadow views (lanes); the main stream is free to start the next frame.  Not when a transparent pass
    // follows (it continues on the main stream with the HDR target) or while stage timing is on.
    const bool on_shade = c->overlap && c->multi_stream && (!c->timing || c->tune.timed_pipeline != 0u) && c->blend_tris == 0;
    hipStream_t stream = on_shade ? c->shade : c->stream;
    if (on_shade) {
        HIP_TRY(c, hipEventRecord(c->vp_ev, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->shade, c->vp_ev, 0));
        for (int k = 0; k < R3N_AUX_STREAMS; ++k)
            if (c->aux_used[k]) {  // the shadow atlas must be complete
                HIP_TRY(c, hipEventRecord(c->join_ev[k], c->aux[k]));
                HIP_TRY(c, hipStreamWaitEvent(c->shade, c->join_ev[k], 0));
                // aux_used[k] stays set: r3n_frame_end still has to order the MAIN stream behind the lanes, because the
                // next frame's r3n_uniform_bake uploads the shadow camera headers (single-buffered) on the main stream
            }
    } else {
        TRY(join_lanes(c));  // the shadow atlas must be complete
        TRY(join_shade(c));  // an earlier frame's resolve may still be writing the HDR / output targets
    }
    ShadeArgs a = make_shade_args(c, r0, r1);
    if (blend_samples) a.samples_out = c->samples16.as<ushort4>();
    if (use_records && c->samples == 1) a.view_lights = c->view_lights[c->slot].as<ViewLights>();
    if (classes && c->resolve_variants != 0u) {
        a.material_feat = c->material_feat.as<uint32_t>();
        a.variants = c->resolve_variants;
    }
    c->resolved_this_frame = true;
    const bool tex = c->n_textures > 0;
    // the vertex stage runs once per visible triangle instead of once per pixel / sample (256 B per triangle slot;
    // skipped for worlds whose record array would pass 8 GiB)
    if (use_records) {
        a.tri_rec = c->tri_rec.as<TriRecord>();
        a.seen = c->tri_seen.as<unsigned char>();
        Timed t(c, R3N_STAGE_VERTEX, stream);
        const size_t first = (size_t)r0 * c->width * c->samples, npx = (size_t)(r1 - r0) * c->width * c->samples;  // keys, not pixels
        HIP_TRY(c, (hipError_t)r3n_internal_shade_prepass(&a, tex ? 1 : 0, first, npx, stream));
    }
    {
        Timed t(c, R3N_STAGE_SHADE, stream);
        if (split) {
            // split resolve: first triangle of every pixel, the extra triangles of edge pixels in a dense second pass
            a.samples_out = c->samples16.as<ushort4>();
            a.edge_list = c->edge_list.as<uint32_t>();
            a.edge_count = c->edge_count.as<uint32_t>();
            a.edge_capacity = edge_cap;
            HIP_TRY(c, hipMemsetAsync(a.edge_count, 0, R3N_EDGEQ * 4, stream));
        }
        HIP_TRY(c, (hipError_t)r3n_internal_resolve(&a, c->samples, tex ? 1 : 0, a.tri_rec != nullptr ? 1 : 0, split ? 1 : 0,
                                                    c->shade_mode == R3N_SHADE_FAST ? 1 : 0, stream));
    }
    TRY(check_launch(c, "k_resolve_opaque"));
    if (on_shade) {
        HIP_TRY(c, hipEventRecord(c->shade_done[c->slot], c->shade));
        c->shade_pending[c->slot] = true;
        c->shade_unjoined = true;
        c->shade_last = c->slot;
    }
    return R3N_OK;
}

int r3n_blend_order_write(r3n_ctx *c, const uint32_t *objects, uint32_t n) {
    if (!c || (n && !objects)) return fail(c, R3N_ERR_INVALID_ARG, "blend order: null");
    std::vector<uint32_t> rank(n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) {
        if (objects[i] >= c->capacity) return fail(c, R3N_ERR_INVALID_ARG, "blend order: object slot >= capacity");
        rank[i + 1] = rank[i] + c->h_ntri[objects[i]];
        // a blend work item's `material` word carries the draw order in bits 0..28 and the edge thresholds in bits 29..31
        // (kernels_raster.h pack_thresholds): an order beyond that would corrupt both
        if (rank[i + 1] >= (1u << 29) || rank[i + 1] < rank[i])
            return fail(c, R3N_ERR_UNSUPPORTED, "blend order: more than 2^29 blend triangles in draw order");
    }
    c->n_blend = n;
    c->blend_tris = rank[n];
    if (n == 0) return R3N_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(ensure(c, c->blend_order, (size_t)n * 4, false, -1));
    TRY(ensure(c, c->blend_rank_base, (size_t)(n + 1) * 4, false, -1));
    c->h_blend_order.assign(objects, objects + n);
    // (the buffers are read by the transparent pass on the main stream: in order with this copy)
    TRY(upload_bulk(c, c->blend_order.p, objects, (size_t)n * 4));  // through pinned staging: no wait for the GPU
    TRY(upload_bulk(c, c->blend_rank_base.p, rank.data(), (size_t)(n + 1) * 4));
    return R3N_OK;
}

// Transparent pass (base.rs:181,451-465): collect the fragments into per-sample lists -> ordered blend.  See kernels_raster.h /
// kernels_shade.h.  Nothing is read back: the launches do not depend on the fragment count, a full node buffer / work queue
// raises the status word (check_async_status).
static int forward_blend(r3n_ctx *c) {
    CamState &s = c->viewport;
    if (!s.culled || c->blend_tris == 0) return R3N_OK;  // nothing culled this frame / no blend triangles
    if (!c->resolved_this_frame) return fail(c, R3N_ERR_STATE, "forward: the transparent pass must follow r3n_resolve_opaque");
    const uint32_t r0 = std::min(c->row_begin, c->height), r1 = std::min(c->row_end, c->height);
    if (r1 <= r0) return R3N_OK;
    const uint32_t S = c->samples;
    const size_t n_samples_all = (size_t)c->width * c->height * S;
    if (n_samples_all > 0xFFFFFFFEull) return fail(c, R3N_ERR_UNSUPPORTED, "forward: transparent pass needs width * height * samples < 2^32");
    TRY(ensure(c, c->frag_keys, (size_t)c->frag_capacity * 8, false, -1));
    TRY(ensure(c, c->frag_vals, (size_t)c->frag_capacity * 4, false, -1));
    TRY(ensure(c, c->frag_head, n_samples_all * 4, false, -1));
    TRY(ensure(c, c->frag_count, 16, false, 0));
    const int idx = s.cur;
    RasterArgs a{};
    a.hdr = s.d_hdr.as<r3n_camera_header240>();
    a.objects = c->objects.as<r3n_object128>();
    a.mesh = c->mesh.as<uint32_t>();
    a.baked = s.baked.as<r3n_baked128>();
    a.materials = c->materials.as<r3n_material208>();
    a.material_keys = c->material_keys.as<uint8_t>();
    a.n_materials = c->n_materials;
    a.tri_base = c->tri_base.as<uint32_t>();
    a.key = R3N_KEY_BLEND;
    a.vp_x = 0; a.vp_y = 0; a.vp_w = c->width; a.vp_h = c->height; a.target_pitch = c->width;
    a.vis = c->vis.as<unsigned long long>();
    a.big_items = c->big_items[0].as<r3n_big_item>();
    const uint32_t fwd = std::min(c->forward_index_lane[0]++, 63u);
    a.big_count = c->big_count[0].as<uint32_t>() + (size_t)fwd * R3N_BIGQ;
    a.big_capacity = c->big_capacity;
    a.big_uv = c->big_uv[0].as<r3n_big_uv>();
    a.tex = texture_args(c);
    a.frag_keys = c->frag_keys.as<unsigned long long>();
    a.frag_vals = c->frag_vals.as<uint32_t>();
    a.frag_count = c->frag_count.as<uint32_t>();
    a.frag_capacity = c->frag_capacity;
    a.frag_head = c->frag_head.as<uint32_t>();
    a.status = c->status_dev;
    a.row_begin = r0; a.row_end = r1;
    BlendSetupArgs b{};
    b.order = c->blend_order.as<uint32_t>();
    b.rank_base = c->blend_rank_base.as<uint32_t>();
    b.n_objects = c->n_blend;
    b.mask = s.mask[idx].as<unsigned long long>();
    b.slot_base = s.slot_base[idx].as<uint32_t>();
    hipStream_t stream = c->stream;
    const size_t first_sample = (size_t)r0 * c->width * S, n_samples = (size_t)(r1 - r0) * c->width * S;
    if (fwd == 63u) HIP_TRY(c, hipMemsetAsync(a.big_count, 0, R3N_BIGQ * 4, stream));
    HIP_TRY(c, hipMemsetAsync(a.frag_count, 0, 4, stream));
    HIP_TRY(c, hipMemsetAsync(a.frag_head + first_sample, 0xFF, n_samples * 4, stream));  // R3N_INVALID: empty lists
    {
        Timed t(c, R3N_STAGE_RASTER, stream);
        hipLaunchKernelGGL(k_blend_setup, dim3((c->blend_tris + 255u) / 256u), dim3(256), 0, stream, a, b);
    }
    {
        Timed t(c, R3N_STAGE_RASTER_BIG, stream);
        if (S == 4) hipLaunchKernelGGL((k_raster_big<false, 4, false, true>), dim3(R3N_BIG_GRID), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_raster_big<false, 1, false, true>), dim3(R3N_BIG_GRID), dim3(256), 0, stream, a);
    }
    TRY(check_launch(c, "blend collect"));
    ShadeArgs sa = make_shade_args(c, r0, r1);
    BlendApplyArgs ba{};
    ba.keys = c->frag_keys.as<unsigned long long>();
    ba.vals = c->frag_vals.as<uint32_t>();
    ba.head = c->frag_head.as<uint32_t>();
    ba.first_sample = (uint32_t)first_sample;
    ba.n_samples = (uint32_t)n_samples;
    ba.capacity = c->frag_capacity;
    ba.samples = S == 4 ? c->samples16.as<ushort4>() : c->hdr16.as<ushort4>();
    if (S == 4 && !c->samples16.p) return fail(c, R3N_ERR_STATE, "forward: r3n_blend_order_write must precede r3n_resolve_opaque");
    {
        Timed t(c, R3N_STAGE_SHADE, stream);
        const bool tex = c->n_textures > 0;
        HIP_TRY(c, (hipError_t)r3n_internal_blend_apply(&sa, &ba, S, tex ? 1 : 0, stream));
        if (S == 4) {
            const size_t first = (size_t)r0 * c->width, npx = (size_t)(r1 - r0) * c->width;
            HIP_TRY(c, (hipError_t)r3n_internal_resolve_samples(c->samples16.as<ushort4>(), c->hdr16.as<ushort4>(), first, npx, stream));
        }
    }
    c->resolved_this_frame = false;  // the fused blit shows the HDR target before blending: r3n_tonemap re-runs it
    c->blended_this_frame = true;
    return check_launch(c, "blend apply");
}

static int launch_tonemap(r3n_ctx *c, float4 *f32_out) {
    const uint32_t r0 = std::min(c->row_begin, c->height), r1 = std::min(c->row_end, c->height);
    if (r1 <= r0) return R3N_OK;
    const size_t first = (size_t)r0 * c->width, n = (size_t)(r1 - r0) * c->width;
    Timed t(c, R3N_STAGE_TONEMAP);
    HIP_TRY(c, (hipError_t)r3n_internal_tonemap(c->hdr16.as<ushort4>(), c->out8.as<uchar4>(), f32_out, first, n, c->srgb_lut.as<unsigned char>(),
                                                c->output_format, c->stream));
    return R3N_OK;
}

int r3n_hdr_write(r3n_ctx *c, const uint16_t *rgba16f, uint64_t first_pixel, uint64_t n_pixels) {
    if (!c || !c->in_frame) return fail(c, R3N_ERR_STATE, "hdr_write: outside a frame");
    const uint64_t total = (uint64_t)c->width * c->height;
    if (!rgba16f || first_pixel > total || n_pixels > total - first_pixel) return fail(c, R3N_ERR_INVALID_ARG, "hdr_write: range outside the target");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(join_shade(c));
    if (n_pixels) {
        HIP_TRY(c, hipMemcpyAsync(c->hdr16.as<uint16_t>() + first_pixel * 4u, rgba16f, n_pixels * 8u, hipMemcpyHostToDevice, c->stream));
        HIP_WAIT(c, hipStreamSynchronize(c->stream));  // caller owns the source only for the duration of the call
    }
    c->resolved_this_frame = false;  // the fused blit no longer matches the HDR target
    return R3N_OK;
}

int r3n_set_output_format(r3n_ctx *c, uint32_t format) {
    if (!c || format > R3N_OUTPUT_BGRA8_UNORM) return fail(c, R3N_ERR_INVALID_ARG, "set_output_format: unknown format");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(sync_all(c));
    if (((format ^ c->output_format) & 2u) != 0u) {  // the transfer function changes: rebuild the half -> 8-bit table
        if (format & 2u) {
            // blit.wgsl fs_main_monitor: 1.055 * pow(x, 0.4166) - 0.055 (math/color.wgsl:13-19).  Built on the host so that
            // the table is what libm gives the oracle too (a device powf may differ in the last bit on a rounding boundary)
            std::vector<unsigned char> lut(R3N_SRGB_LUT_SIZE);
            for (uint32_t h = 0; h < R3N_SRGB_LUT_SIZE; ++h) {
                const uint32_t e = (h >> 10) & 31u, m = h & 1023u;
                const float x = e == 0u ? std::ldexp((float)m, -24) : std::ldexp((float)(m | 1024u), (int)e - 25);
                float v = x > 0.0031308f ? 1.055f * std::pow(x, 0.4166f) - 0.055f : x * 12.92f;
                v = !(v > 0.0f) ? 0.0f : (v >= 1.0f ? 1.0f : v);
                lut[h] = (unsigned char)(v * 255.0f + 0.5f);
            }
            HIP_TRY(c, hipMemcpy(c->srgb_lut.p, lut.data(), lut.size(), hipMemcpyHostToDevice));
        } else {
            HIP_TRY(c, (hipError_t)r3n_internal_build_srgb_lut(c->srgb_lut.as<unsigned char>(), c->stream));
            HIP_WAIT(c, hipStreamSynchronize(c->stream));
        }
    }
    c->output_format = format;
    return R3N_OK;
}

int r3n_set_skinning_mode(r3n_ctx *c, uint32_t mode) {
    if (!c || mode > R3N_SKIN_MFMA) return fail(c, R3N_ERR_INVALID_ARG, "set_skinning_mode: unknown mode");
    c->skinning_mode = mode;
    return R3N_OK;
}

int r3n_set_shade_mode(r3n_ctx *c, uint32_t mode) {
    if (!c || mode > R3N_SHADE_FAST) return fail(c, R3N_ERR_INVALID_ARG, "set_shade_mode: unknown mode");
    c->shade_mode = mode;  // read when the next resolve is enqueued
    return R3N_OK;
}

int r3n_tonemap(r3n_ctx *c, void *host_rgba8, uint64_t pitch) {
    if (!c || !c->in_frame) return fail(c, R3N_ERR_STATE, "tonemap: outside a frame");
    HIP_TRY(c, hipSetDevice(c->device));
    // the blit is fused into r3n_resolve_opaque (same arithmetic on the Rgba16Float-rounded value); the separate
    // kernel only runs when the HDR buffer was produced some other way
    if (!c->resolved_this_frame) {
        TRY(join_shade(c));
        TRY(launch_tonemap(c, nullptr));
    }
    if (host_rgba8) {
        TRY(join_shade(c));
        if (pitch < (uint64_t)c->width * 4) return fail(c, R3N_ERR_INVALID_ARG, "tonemap: pitch too small");
        HIP_TRY(c, hipMemcpy2DAsync(host_rgba8, pitch, c->out8.p, (size_t)c->width * 4, (size_t)c->width * 4, c->height,
                                    hipMemcpyDeviceToHost, c->stream));
        HIP_WAIT(c, hipStreamSynchronize(c->stream));
    }
    return R3N_OK;
}

// Closes the frame (flushes the batched shadow stages, joins the lanes, flips every camera's ping-pong state) WITHOUT looking at the
// asynchronous status word: r3n_frame_end reports that word once, and an error path that merely has to close the frame
// (r3n_render_frame's Closer) must not consume the report of an earlier frame's overflow.
static int close_frame(r3n_ctx *c) {
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(join_lanes(c));  // the next frame's clears (main stream) must not overtake this frame's shadow work
    auto flip = [](CamState &s) {
        if (s.culled) { s.cur = 1 - s.cur; s.has_prev = true; }
        s.culled = false;
    };
    flip(c->viewport);
    for (auto &kv : c->shadows) flip(kv.second);
    c->in_frame = false;
    return R3N_OK;
}

int r3n_frame_end(r3n_ctx *c) {
    if (!c || !c->in_frame) return fail(c, R3N_ERR_STATE, "frame_end: no frame in flight");
    TRY(close_frame(c));
    // this frame is closed and complete as far as the host can know; what the status word holds was raised by an EARLIER frame's
    // kernels (fragment buffer / work queue full): report it here, once, without having refused any frame
    return check_async_status(c);
}

// ------------------------------------------------------------------------------------------------ native exchange (r3n_comm_*)
#define NCCL_TRY(c, expr)                                                                                          \
    do {                                                                                                           \
        crumb(c, "> " #expr);                                                                                      \
        ncclResult_t _r = (expr);                                                                                  \
        crumb(c, "< returned", (long long)_r);                                                                     \
        if (_r != ncclSuccess) return fail(c, R3N_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(_r)); \
    } while (0)

// rows of band `r` of `world` (rend3_amd/parallel.py::row_ranges: as even as possible, the first height mod world one row taller)
static void band_rows(uint32_t height, uint32_t world, uint32_t r, uint32_t &b, uint32_t &e) {
    const uint32_t base = height / world, rem = height % world;
    b = r * base + std::min(r, rem);
    e = b + base + (r < rem ? 1u : 0u);
}

int r3n_comm_unique_id(uint8_t *id) {
    static_assert(R3N_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "r3n.h mirrors ncclUniqueId's size");
    if (!id || !rccl().load()) return R3N_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return R3N_ERR_HIP;
    std::memcpy(id, u.internal, R3N_COMM_ID_BYTES);
    return R3N_OK;
}
int r3n_comm_init(r3n_ctx *c, const uint8_t *ids, uint32_t rank, uint32_t world) {
    if (!c || !ids || world == 0 || rank >= world) return fail(c, R3N_ERR_INVALID_ARG, "comm_init: bad arguments");
    if (c->comm.on) return fail(c, R3N_ERR_STATE, "comm_init: the context already has communicators");
    if (!rccl().load()) return fail(c, R3N_ERR_UNSUPPORTED, "comm_init: " + rccl().error);
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(sync_all(c));
    ncclComm_t *dst[R3N_COMM_IDS] = {&c->comm.main, &c->comm.shadow, &c->comm.rows};
    for (int k = 0; k < R3N_COMM_IDS; ++k) {
        crumb(c, "comm_init communicator", k, world);
        ncclUniqueId u;
        std::memcpy(u.internal, ids + (size_t)k * R3N_COMM_ID_BYTES, R3N_COMM_ID_BYTES);
        const ncclResult_t rc = rccl().CommInitRank(dst[k], (int)world, u, (int)rank);
        if (rc != ncclSuccess) {  // the communicators made so far must not outlive the failure (comm.on stays false: nobody else frees them)
            const std::string why = rccl().GetErrorString(rc);
            for (int j = 0; j < k; ++j) { (void)rccl().CommDestroy(*dst[j]); *dst[j] = nullptr; }
            *dst[k] = nullptr;
            return fail(c, R3N_ERR_HIP, "comm_init: ncclCommInitRank (communicator " + std::to_string(k) + "): " + why);
        }
    }
    c->comm.on = true; c->comm.rank = rank; c->comm.world = world;
    c->shard_rows = true;
    crumb(c, "comm_init done", rank, world);
    return R3N_OK;
}
int r3n_comm_destroy(r3n_ctx *c) {
    if (!c) return R3N_ERR_INVALID_ARG;
    if (!c->comm.on) return R3N_OK;
    crumb(c, "comm_destroy");
    (void)hipSetDevice(c->device);
    (void)sync_all(c);
    crumb(c, "comm_destroy synced");
    for (ncclComm_t *k : {&c->comm.main, &c->comm.shadow, &c->comm.rows})
        if (*k) { (void)rccl().CommDestroy(*k); *k = nullptr; }
    for (DevBuf &b : c->comm.stage)
        if (b.p) { (void)hipFree(b.p); b = DevBuf{}; }
    if (c->comm.order_ev) { (void)hipEventDestroy(c->comm.order_ev); c->comm.order_ev = nullptr; }
    c->comm.order_pending = false;
    c->comm.on = false; c->comm.by_objects = false; c->comm.world = 1; c->comm.rank = 0;
    c->shard_rows = false; c->row_begin = 0; c->row_end = 0xFFFFFFFFu;
    for (auto &kv : c->shadows) { kv.second.band_begin = 0; kv.second.band_end = 0xFFFFFFFFu; }  // (shadow views split over ranks: whole again)
    return R3N_OK;
}
// Who draws what of the shadow views (native exchange).  With at least as many views as ranks a view belongs to rank v mod N, whole.
// With MORE ranks than views (N = 8, four views: VERDICT r4 weak #8 -- "a whole shadow view per rank" was part of what does not
// divide) a view's rows are split into parts = N / V bands and band p of view v belongs to rank v + p V: the rank culls the view's
// casters like its owner would (the lists are the unsharded ones) and its rasterisers scan the band's rows only
// (RasterArgs.row_begin / row_end, the clamp the sort-first viewport split uses); ranks beyond parts * V own nothing.
static uint32_t shadow_parts(uint32_t world, uint32_t n_views) { return n_views && world > n_views ? world / n_views : 1u; }
static uint32_t shadow_owner(uint32_t world, uint32_t n_views, uint32_t v, uint32_t part) {
    return world > n_views ? v + part * n_views : v % world;
}
static int shadow_part_of(uint32_t world, uint32_t n_views, uint32_t rank, uint32_t v) {  // the part of view v this rank draws, or -1
    const uint32_t parts = shadow_parts(world, n_views);
    for (uint32_t p = 0; p < parts; ++p)
        if (shadow_owner(world, n_views, v, p) == rank) return (int)p;
    return -1;
}
// ONE total order of the collectives per device (tune.comm_serial, default on).  The three communicators are driven from three streams
// -- shadow lane, main, resolve -- so that an exchange overlaps the other lanes' kernels; nothing orders their collective KERNELS
// against each other, and RCCL (like NCCL) promises progress for concurrent communicators only if every rank's device can hold
// all of their kernels at once: rank A running `shadow` first while rank B runs `main` first, neither with room for the other,
// would wait for each other for ever.  One GPU per rank with three small communicators has that room today; an 8-GPU node has
// never run this code, so the default does not lean on it: every collective waits (on its own stream, through one event) for the
// collective enqueued before it -- whichever stream that was on -- and since every rank enqueues them in the same program order,
// every device executes them in the same order.  The kernels between the collectives overlap as before.
// tests: tests/rccl_shim.cpp in its asynchronous mode executes a collective when the stream REACHES it, one at a time per process
// (the worst case above), so an order dependence between communicators shows up as a time-out there.
static int comm_order_begin(r3n_ctx *c, hipStream_t on) {
    if (!c->tune.comm_serial) return R3N_OK;
    if (!c->comm.order_ev) HIP_TRY(c, hipEventCreateWithFlags(&c->comm.order_ev, hipEventDisableTiming));
    if (c->comm.order_pending) HIP_TRY(c, hipStreamWaitEvent(on, c->comm.order_ev, 0));
    return R3N_OK;
}
static int comm_order_end(r3n_ctx *c, hipStream_t on) {
    if (!c->tune.comm_serial) return R3N_OK;
    HIP_TRY(c, hipEventRecord(c->comm.order_ev, on));
    c->comm.order_pending = true;
    return R3N_OK;
}
// The owners' atlas rows to every rank, on the shadow lane's stream (r3n_exchange_shadow_stream): one broadcast per (view, band).
static int comm_exchange_shadows(r3n_ctx *c, const r3n_frame_desc *d) {
    void *atlas = nullptr, *sp = nullptr;
    uint64_t n = 0;
    TRY(r3n_exchange_shadow_stream(c, &atlas, &n, &sp));
    hipStream_t on = (hipStream_t)sp;
    const uint32_t world = c->comm.world, rank = c->comm.rank, aw = c->atlas_w;
    for (uint32_t v = 0; v < d->n_shadow_views; ++v) {  // staging first: allocations synchronise
        const size_t want = (size_t)d->shadow_views[v].size * d->shadow_views[v].size * 4;
        DevBuf &b = c->comm.stage[v];
        if (b.bytes < want) {
            if (b.p) { TRY(sync_all(c)); HIP_TRY(c, hipFree(b.p)); b = DevBuf{}; }
            HIP_TRY(c, hipMalloc(&b.p, want));
            b.bytes = want;
        }
    }
    Timed t(c, R3N_STAGE_EXCHANGE_SHADOW, on);
    const uint32_t nv = d->n_shadow_views, parts = shadow_parts(world, nv);
    // rows [b, e) of view v = band `part`; its place in the atlas and in the view's staging buffer (rows of `size` texels)
    auto rows_of = [&](const r3n_shadow_view272 &sv, uint32_t part, uint32_t &b, uint32_t &e) { band_rows(sv.size, parts, part, b, e); };
    auto in_atlas = [&](const r3n_shadow_view272 &sv, uint32_t row) { return static_cast<char *>(atlas) + ((size_t)(sv.y + row) * aw + sv.x) * 4; };
    auto in_stage = [&](uint32_t v, const r3n_shadow_view272 &sv, uint32_t row) { return static_cast<char *>(c->comm.stage[v].p) + (size_t)row * sv.size * 4; };
    for (uint32_t v = 0; v < nv; ++v)
        for (uint32_t p = 0; p < parts; ++p)
            if (shadow_owner(world, nv, v, p) == rank) {
                const r3n_shadow_view272 &sv = d->shadow_views[v];
                uint32_t b, e;
                rows_of(sv, p, b, e);
                if (e > b) HIP_TRY(c, hipMemcpy2DAsync(in_stage(v, sv, b), (size_t)sv.size * 4, in_atlas(sv, b), (size_t)aw * 4, (size_t)sv.size * 4, e - b, hipMemcpyDeviceToDevice, on));
            }
    TRY(comm_order_begin(c, on));
    NCCL_TRY(c, rccl().GroupStart());  // the broadcasts progress together
    for (uint32_t v = 0; v < nv; ++v)
        for (uint32_t p = 0; p < parts; ++p) {
            const r3n_shadow_view272 &sv = d->shadow_views[v];
            uint32_t b, e;
            rows_of(sv, p, b, e);
            if (e > b) NCCL_TRY(c, rccl().Broadcast(in_stage(v, sv, b), in_stage(v, sv, b), (size_t)(e - b) * sv.size, ncclFloat, (int)shadow_owner(world, nv, v, p), c->comm.shadow, on));
        }
    NCCL_TRY(c, rccl().GroupEnd());
    TRY(comm_order_end(c, on));
    for (uint32_t v = 0; v < nv; ++v)
        for (uint32_t p = 0; p < parts; ++p)
            if (shadow_owner(world, nv, v, p) != rank) {
                const r3n_shadow_view272 &sv = d->shadow_views[v];
                uint32_t b, e;
                rows_of(sv, p, b, e);
                if (e > b) HIP_TRY(c, hipMemcpy2DAsync(in_atlas(sv, b), (size_t)aw * 4, in_stage(v, sv, b), (size_t)sv.size * 4, (size_t)sv.size * 4, e - b, hipMemcpyDeviceToDevice, on));
            }
    return R3N_OK;
}
// `base`: a buffer of height rows of row_bytes each whose rows [band of this rank) are final: afterwards every rank holds every band
static int comm_gather_bands(r3n_ctx *c, void *base, size_t row_bytes, ncclComm_t comm, hipStream_t on) {
    const uint32_t world = c->comm.world, rank = c->comm.rank, h = c->height;
    char *p = static_cast<char *>(base);
    TRY(comm_order_begin(c, on));
    if (h % world == 0) {
        const size_t chunk = (size_t)(h / world) * row_bytes;
        NCCL_TRY(c, rccl().AllGather(p + (size_t)rank * chunk, p, chunk, ncclUint8, comm, on));  // in place
        return comm_order_end(c, on);
    }
    NCCL_TRY(c, rccl().GroupStart());  // ragged bands: one broadcast per band
    for (uint32_t r = 0; r < world; ++r) {
        uint32_t b, e;
        band_rows(h, world, r, b, e);
        if (e > b) NCCL_TRY(c, rccl().Broadcast(p + (size_t)b * row_bytes, p + (size_t)b * row_bytes, (size_t)(e - b) * row_bytes, ncclUint8, (int)r, comm, on));
    }
    NCCL_TRY(c, rccl().GroupEnd());
    return comm_order_end(c, on);
}
static int comm_exchange_pass1(r3n_ctx *c) {
    if (c->samples == 1) {
        void *plane = nullptr;
        uint64_t n = 0;
        TRY(r3n_exchange_depth(c, &plane, &n));  // mip 0 of the Hi-Z pyramid from the keys: this rank's rows are final
        Timed t(c, R3N_STAGE_EXCHANGE_DEPTH, c->stream);
        return comm_gather_bands(c, plane, (size_t)c->width * 4, c->comm.main, c->stream);
    }
    Timed t(c, R3N_STAGE_EXCHANGE_DEPTH, c->stream);  // multisampled: Hi-Z reads the keys
    return comm_gather_bands(c, c->vis.p, (size_t)c->width * c->samples * 8, c->comm.main, c->stream);
}
// Object-range split (BASELINE.json north_star): every rank drew ITS objects over the whole target.
// Pass 1: the Hi-Z cull needs the global depth -- element-wise MAX over ranks (reverse-Z: nearest = largest) of the f32 plane, or of
// the keys under MSAA (Hi-Z reads them).  Pass 2: the nearest fragment of every pixel = MAX over ranks of the u64 keys
// (depth bits << 32 | triangle), needed only by the rank that resolves the pixel's row: an in-place reduce-scatter onto the row
// bands when they are equal, an all-reduce otherwise.
static int comm_reduce_pass1(r3n_ctx *c) {
    if (c->samples == 1) {
        void *plane = nullptr;
        uint64_t n = 0;
        TRY(r3n_exchange_depth(c, &plane, &n));
        Timed t(c, R3N_STAGE_EXCHANGE_DEPTH, c->stream);
        TRY(comm_order_begin(c, c->stream));
        NCCL_TRY(c, rccl().AllReduce(plane, plane, (size_t)n, ncclFloat32, ncclMax, c->comm.main, c->stream));
        return comm_order_end(c, c->stream);
    }
    Timed t(c, R3N_STAGE_EXCHANGE_DEPTH, c->stream);
    TRY(comm_order_begin(c, c->stream));
    NCCL_TRY(c, rccl().AllReduce(c->vis.p, c->vis.p, (size_t)c->width * c->height * c->samples, ncclUint64, ncclMax, c->comm.main, c->stream));
    return comm_order_end(c, c->stream);
}
static int comm_reduce_pass2(r3n_ctx *c) {
    const uint32_t world = c->comm.world, rank = c->comm.rank, h = c->height;
    const size_t row_keys = (size_t)c->width * c->samples;
    unsigned long long *keys = c->vis.as<unsigned long long>();
    Timed t(c, R3N_STAGE_EXCHANGE_KEYS, c->stream);
    TRY(comm_order_begin(c, c->stream));
    if (h % world == 0) {
        const size_t chunk = (size_t)(h / world) * row_keys;
        NCCL_TRY(c, rccl().ReduceScatter(keys, keys + (size_t)rank * chunk, chunk, ncclUint64, ncclMax, c->comm.main, c->stream));  // in place
    } else {
        NCCL_TRY(c, rccl().AllReduce(keys, keys, (size_t)h * row_keys, ncclUint64, ncclMax, c->comm.main, c->stream));
    }
    return comm_order_end(c, c->stream);
}
static int comm_gather_rows(r3n_ctx *c) {
    void *out = nullptr, *sp = nullptr;
    uint64_t bytes = 0;
    TRY(r3n_output_buffer_async(c, &out, &bytes, &sp));
    {
        Timed t(c, R3N_STAGE_EXCHANGE_ROWS, (hipStream_t)sp);
        TRY(comm_gather_bands(c, out, (size_t)c->width * 4, c->comm.rows, (hipStream_t)sp));
    }
    return r3n_output_work_enqueued(c);
}

// ------------------------------------------------------------------------------------------------ the frame in one call
// BaseRenderGraph::add_to_graph's node list (base.rs:135-185) issued from here: the per-node entry points above, in the reference's
// order, without a host-language graph (or ~45 FFI crossings) between them.
int r3n_render_frame(r3n_ctx *c, const r3n_frame_desc *d) {
    if (!c || !d || d->struct_size < sizeof(r3n_frame_desc)) return fail(c, R3N_ERR_INVALID_ARG, "render_frame: bad descriptor");
    if (!d->uniforms || !d->viewport_header || (d->n_shadow_views && !d->shadow_views) || d->n_shadow_views > R3N_MAX_SHADOW_VIEWS)
        return fail(c, R3N_ERR_INVALID_ARG, "render_frame: null uniforms / headers, or too many shadow views");
    if (c->in_frame) return fail(c, R3N_ERR_STATE, "render_frame: a frame is already open");
    const bool native = c->comm.on;
    crumb(c, "render_frame", d->width, d->height);
    if (native) {
        if (d->exchange) return fail(c, R3N_ERR_INVALID_ARG, "render_frame: an exchange callback AND r3n_comm_init communicators");
        band_rows(d->height, c->comm.world, c->comm.rank, c->row_begin, c->row_end);
    }
    for (uint32_t v = 0; v < d->n_shadow_views; ++v)
        if (d->shadow_views[v].header.shadow_index != v) return fail(c, R3N_ERR_INVALID_ARG, "render_frame: shadow view i must carry shadow_index i");
    if (d->directional_buffer) TRY(r3n_lights_write(c, d->directional_buffer, d->directional_bytes, d->point_buffer, d->point_bytes));
    // clear_shadow_buffers + create_frame_uniforms (base.rs:139,142)
    TRY(frame_begin_impl(c, d->uniforms, d->width, d->height, d->samples, d->clear_color, d->shadow_atlas_width, d->shadow_atlas_height, d));
    // every camera header went up with the uniforms
    c->viewport.hdr_frame = c->frame_no;
    for (uint32_t v = 0; v < d->n_shadow_views; ++v)
        if (CamState *s = find_cam(c, v, true)) {
            s->hdr_frame = c->frame_no;
            s->d_hdr.p = c->fb_dev[c->slot].as<uint8_t>() + r3n_ctx::kFbShadowHdr + 256u * (size_t)v;
            s->d_hdr.bytes = 256;
        }
    struct Closer {  // an error in the middle must not leave the frame open
        r3n_ctx *c; bool armed = true;
        ~Closer() { if (armed && c->in_frame) { const std::string keep = c->err; (void)close_frame(c); c->err = keep; } }
    } closer{c};
    struct Fused { r3n_ctx *c; ~Fused() { c->fused_frame = false; } } fused{c};
    c->fused_frame = true;
    const bool masked = native || (d->flags & R3N_FRAME_SHADOW_MASK) != 0u;
    auto mine = [&](uint32_t v) {
        if (native) return shadow_part_of(c->comm.world, d->n_shadow_views, c->comm.rank, v) >= 0;
        return !masked || ((d->shadow_view_mask >> v) & 1ull) != 0ull;
    };
    for (uint32_t v = 0; v < d->n_shadow_views; ++v) {
        const r3n_shadow_view272 &sv = d->shadow_views[v];
        TRY(r3n_shadow_viewport(c, v, sv.x, sv.y, sv.size));
        {  // the rows of the view this rank rasterises: all of them, or its band when the view is split over ranks (set every frame)
            CamState *s = find_cam(c, v, true);
            if (!s) return fail(c, R3N_ERR_INVALID_ARG, "render_frame: bad shadow camera");
            s->band_begin = 0; s->band_end = 0xFFFFFFFFu;
            const int part = native ? shadow_part_of(c->comm.world, d->n_shadow_views, c->comm.rank, v) : -1;
            if (part >= 0) band_rows(sv.size, shadow_parts(c->comm.world, d->n_shadow_views), (uint32_t)part, s->band_begin, s->band_end);
        }
        // multi-GPU: a view this rank owns is drawn WHOLE here, whatever the viewport's object range is -- set every frame, so the
        // ownership test and the range cannot disagree (a view that changed owner returns to the global range)
        if (masked) TRY(r3n_set_camera_object_range(c, v, mine(v) ? 0u : 0xFFFFFFFFu, mine(v) ? 0xFFFFFFFEu : 0xFFFFFFFFu));
    }
    // skinning (base.rs:145)
    if (d->n_skeletons) TRY(r3n_skinning(c, d->skin_inputs, d->n_skeletons, d->joint_matrices, d->n_joint_matrices));
    auto shadow_nodes = [&]() -> int {
        for (uint32_t v = 0; v < d->n_shadow_views; ++v)  // shadow_object_uniform_upload (base.rs:148)
            if (mine(v)) TRY(r3n_uniform_bake(c, v, &d->shadow_views[v].header));
        for (uint32_t v = 0; v < d->n_shadow_views; ++v)  // pbr_shadow_culling (base.rs:150)
            if (mine(v)) TRY(r3n_cull(c, v));
        for (uint32_t v = 0; v < d->n_shadow_views; ++v)  // pbr_shadow_rendering (base.rs:153,366-396)
            if (mine(v)) {
                TRY(r3n_forward(c, v, R3N_PASS_DEPTH, R3N_SOURCE_RESIDUAL, R3N_KEY_OPAQUE));
                TRY(r3n_forward(c, v, R3N_PASS_DEPTH, R3N_SOURCE_RESIDUAL, R3N_KEY_CUTOUT));
            }
        if (d->exchange && d->n_shadow_views && d->exchange(d->exchange_user, R3N_EXCHANGE_SHADOW) != 0)
            return fail(c, R3N_ERR_STATE, "render_frame: the shadow exchange callback failed");
        return R3N_OK;
    };
    auto viewport_pass1 = [&]() -> int {
        TRY(r3n_uniform_bake(c, R3N_CAMERA_VIEWPORT, d->viewport_header));  // object_uniform_upload (base.rs:156)
        // pbr_render_opaque_predicted_triangles (base.rs:159)
        TRY(r3n_forward(c, R3N_CAMERA_VIEWPORT, R3N_PASS_FORWARD, R3N_SOURCE_PREDICTED, R3N_KEY_OPAQUE));
        TRY(r3n_forward(c, R3N_CAMERA_VIEWPORT, R3N_PASS_FORWARD, R3N_SOURCE_PREDICTED, R3N_KEY_CUTOUT));
        return R3N_OK;
    };
    if (d->flags & R3N_FRAME_VIEWPORT_FIRST) { TRY(viewport_pass1()); TRY(shadow_nodes()); }
    else { TRY(shadow_nodes()); TRY(viewport_pass1()); }
    if (d->exchange && d->exchange(d->exchange_user, R3N_EXCHANGE_PASS1) != 0) return fail(c, R3N_ERR_STATE, "render_frame: the pass-1 exchange callback failed");
    if (native) TRY(c->comm.by_objects ? comm_reduce_pass1(c) : comm_exchange_pass1(c));
    // The shadow views' broadcast is ENQUEUED here, behind the pass-1 exchange, although the views were drawn first: under the one
    // total order of the collectives (comm_order_begin) the main stream's pass-1 exchange would otherwise wait for the shadow lanes'
    // broadcast -- i.e. for the shadow views, the longest chain of the frame -- before Hi-Z could start.  In this order the broadcast
    // waits for the (early, short) pass-1 exchange instead, and nothing but the resolve waits for the atlas, as before.
    if (native && d->n_shadow_views) TRY(comm_exchange_shadows(c, d));
    TRY(r3n_hi_z(c));                         // hi_z (base.rs:162)
    TRY(r3n_cull(c, R3N_CAMERA_VIEWPORT));    // pbr_culling (base.rs:169)
    // pbr_render_opaque_residual_triangles (base.rs:172)
    TRY(r3n_forward(c, R3N_CAMERA_VIEWPORT, R3N_PASS_FORWARD, R3N_SOURCE_RESIDUAL, R3N_KEY_OPAQUE));
    TRY(r3n_forward(c, R3N_CAMERA_VIEWPORT, R3N_PASS_FORWARD, R3N_SOURCE_RESIDUAL, R3N_KEY_CUTOUT));
    if (d->exchange && d->exchange(d->exchange_user, R3N_EXCHANGE_PASS2) != 0) return fail(c, R3N_ERR_STATE, "render_frame: the pass-2 exchange callback failed");
    if (native && c->comm.by_objects) TRY(comm_reduce_pass2(c));
    TRY(r3n_resolve_opaque(c));               // the opaque passes' fragment stages, deferred
    // pbr_forward_rendering_transparent (base.rs:181)
    TRY(r3n_forward(c, R3N_CAMERA_VIEWPORT, R3N_PASS_FORWARD, R3N_SOURCE_RESIDUAL, R3N_KEY_BLEND));
    TRY(r3n_tonemap(c, nullptr, 0));          // tonemapping (base.rs:184)
    if (native) TRY(comm_gather_rows(c));
    closer.armed = false;
    crumb(c, "render_frame enqueued");
    return r3n_frame_end(c);
}

// ------------------------------------------------------------------------------------------------ multi-GPU
int r3n_set_object_range(r3n_ctx *c, uint32_t begin, uint32_t end) {
    if (!c || begin > end) return fail(c, R3N_ERR_INVALID_ARG, "set_object_range: begin > end");
    c->range_begin = begin; c->range_end = end;
    return R3N_OK;
}
int r3n_set_object_owners(r3n_ctx *c, const uint8_t *owners, uint32_t n, uint32_t rank) {
    if (!c) return R3N_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(sync_all(c));  // world-edit rate, not per frame
    if (!owners || n == 0) {  // back to r3n_set_object_range
        if (c->owners.p) { (void)hipFree(c->owners.p); c->owners.p = nullptr; c->owners.bytes = 0; }
        c->owners_n = 0;
        return R3N_OK;
    }
    if (n < c->capacity) return fail(c, R3N_ERR_INVALID_ARG, "set_object_owners: one owner byte per object slot (n >= capacity)");
    TRY(ensure(c, c->owners, n, false, -1));
    HIP_TRY(c, hipMemcpy(c->owners.p, owners, n, hipMemcpyHostToDevice));
    c->owners_n = n;
    c->owner_rank = rank;
    return R3N_OK;
}
int r3n_set_camera_object_range(r3n_ctx *c, r3n_camera cam, uint32_t begin, uint32_t end) {
    if (!c) return R3N_ERR_INVALID_ARG;
    CamState *s = find_cam(c, cam, true);
    if (!s) return fail(c, R3N_ERR_INVALID_ARG, "set_camera_object_range: bad camera");
    if (begin == 0xFFFFFFFFu && end == 0xFFFFFFFFu) { s->range_set = false; return R3N_OK; }  // follow r3n_set_object_range again
    if (begin > end) return fail(c, R3N_ERR_INVALID_ARG, "set_camera_object_range: begin > end");
    s->range_set = true; s->range_begin = begin; s->range_end = end;
    return R3N_OK;
}
// Pass-1 depth for the Hi-Z exchange (single-sample targets): mip 0 of the pyramid is derived from the keys now and its address
// handed out; the r3n_hi_z that follows builds the upper levels from that plane (MAX-merged across ranks in between) instead of
// from this rank's keys.
int r3n_exchange_depth(r3n_ctx *c, void **depth_f32, uint64_t *count) {
    if (!c || !c->in_frame || !c->vis.p) return fail(c, R3N_ERR_STATE, "exchange_depth: outside a frame");
    if (c->samples != 1) return fail(c, R3N_ERR_UNSUPPORTED, "exchange_depth: single-sample targets only (a multisampled pass-1 exchange carries the keys)");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->width * c->height;
    {
        Timed t(c, R3N_STAGE_HIZ);
        hipLaunchKernelGGL(k_hiz_mip0, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535u * 16u)), dim3(256), 0, c->stream,
                           c->vis.as<unsigned long long>(), c->hiz.as<float>(), n);
    }
    TRY(check_launch(c, "k_hiz_mip0"));
    c->hiz_plane_ready = true;
    if (depth_f32) *depth_f32 = c->hiz.p;
    if (count) *count = n;
    return R3N_OK;
}
int r3n_exchange_buffers(r3n_ctx *c, void **vis, uint64_t *vis_count, void **atlas, uint64_t *atlas_count) {
    if (!c || !c->vis.p) return fail(c, R3N_ERR_STATE, "exchange_buffers: no frame targets yet");
    TRY(join_lanes(c));  // collectives are ordered on the main stream
    if (vis) *vis = c->vis.p;
    if (vis_count) *vis_count = (uint64_t)c->width * c->height * c->samples;
    if (atlas) *atlas = c->atlas.p;
    if (atlas_count) *atlas_count = (uint64_t)c->atlas_w * c->atlas_h;
    return R3N_OK;
}
int r3n_exchange_shadow_stream(r3n_ctx *c, void **atlas, uint64_t *atlas_count, void **stream) {
    if (!c || !c->atlas.p) return fail(c, R3N_ERR_STATE, "exchange_shadow_stream: no shadow atlas yet");
    hipStream_t on = c->stream;
    if (c->multi_stream) {
        // lane 1 carries the exchange: behind the main stream's clears / uploads of this frame, and behind the other lanes' views
        TRY(fork_lane(c, 1));
        for (int k = 1; k < R3N_AUX_STREAMS; ++k)
            if (c->aux_used[k]) {
                HIP_TRY(c, hipEventRecord(c->join_ev[k], c->aux[k]));
                HIP_TRY(c, hipStreamWaitEvent(c->aux[0], c->join_ev[k], 0));
            }
        on = c->aux[0];  // aux_used[0] is set (fork_lane): the resolve and r3n_frame_end wait for what is enqueued here
    }
    if (atlas) *atlas = c->atlas.p;
    if (atlas_count) *atlas_count = (uint64_t)c->atlas_w * c->atlas_h;
    if (stream) *stream = (void *)on;
    return R3N_OK;
}
int r3n_comm_set_split(r3n_ctx *c, uint32_t mode) {
    if (!c || mode > R3N_SHARD_ROWS) return fail(c, R3N_ERR_INVALID_ARG, "comm_set_split: unknown split");
    if (!c->comm.on) return fail(c, R3N_ERR_STATE, "comm_set_split: no communicators (r3n_comm_init)");
    c->comm.by_objects = mode == R3N_SHARD_OBJECTS;
    c->shard_rows = !c->comm.by_objects;  // rows: the viewport rasterises its band only; objects: its objects over the whole target
    return R3N_OK;
}
int r3n_set_shard_mode(r3n_ctx *c, uint32_t mode) {
    if (!c || mode > R3N_SHARD_ROWS) return fail(c, R3N_ERR_INVALID_ARG, "set_shard_mode: unknown mode");
    c->shard_rows = mode == R3N_SHARD_ROWS;
    return R3N_OK;
}
int r3n_set_row_range(r3n_ctx *c, uint32_t b, uint32_t e) {
    if (!c || b > e) return fail(c, R3N_ERR_INVALID_ARG, "set_row_range: begin > end");
    c->row_begin = b; c->row_end = e;
    return R3N_OK;
}
int r3n_output_buffer(r3n_ctx *c, void **rgba8, uint64_t *bytes) {
    if (!c || !c->out8.p) return fail(c, R3N_ERR_STATE, "output_buffer: no frame targets yet");
    TRY(join_shade(c));  // whoever reads the buffer orders itself on the main stream (r3n_stream)
    if (rgba8) *rgba8 = c->out8.p;
    if (bytes) *bytes = (uint64_t)c->width * c->height * 4;
    return R3N_OK;
}
int r3n_output_buffer_async(r3n_ctx *c, void **rgba8, uint64_t *bytes, void **stream) {
    if (!c || !c->out8.p) return fail(c, R3N_ERR_STATE, "output_buffer: no frame targets yet");
    if (rgba8) *rgba8 = c->out8.p;
    if (bytes) *bytes = (uint64_t)c->width * c->height * 4;
    if (stream) *stream = c->shade_unjoined ? (void *)c->shade : (void *)c->stream;
    return R3N_OK;
}
int r3n_output_work_enqueued(r3n_ctx *c) {
    if (!c) return R3N_ERR_INVALID_ARG;
    // the event later waits go through (join_shade, r3n_frame_begin's reuse of this frame slot) now also covers that work
    if (c->shade_unjoined) HIP_TRY(c, hipEventRecord(c->shade_done[c->shade_last], c->shade));
    return R3N_OK;
}

// ------------------------------------------------------------------------------------------------ readbacks
static int d2h(r3n_ctx *c, void *dst, const void *src, size_t bytes) {
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(join_lanes(c));
    TRY(join_shade(c));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_WAIT(c, hipStreamSynchronize(c->stream));
    return check_async_status(c);
}

int r3n_readback_visible_objects(r3n_ctx *c, r3n_camera cam, uint8_t *flags, uint32_t capacity) {
    CamState *s = c ? find_cam(c, cam, false) : nullptr;
    if (!s || s->last < 0 || !flags) return fail(c, R3N_ERR_STATE, "readback_visible_objects: camera never culled");
    if (capacity < c->capacity) return fail(c, R3N_ERR_INVALID_ARG, "readback_visible_objects: buffer too small");
    TRY(d2h(c, flags, s->vis_flags.p, c->capacity));
    for (uint32_t i = 0; i < c->capacity; ++i) flags[i] &= (uint8_t)R3N_VIS_DRAWN;  // (bit 1: frustum test alone, kernels_cull.h R3N_VIS_INSIDE)
    return R3N_OK;
}

int r3n_readback_draw_calls(r3n_ctx *c, r3n_camera cam, r3n_indirect_call calls[6]) {
    CamState *s = c ? find_cam(c, cam, false) : nullptr;
    if (!s || s->last < 0) return fail(c, R3N_ERR_STATE, "readback_draw_calls: camera never culled");
    r3n_sub_counts sc;
    r3n_cull_counts counts;
    TRY(d2h(c, &sc, s->sub_counts[s->last].p, sizeof sc));
    TRY(d2h(c, &counts, s->counts[s->last].p, sizeof counts));
    for (int list = 0; list < 2; ++list)
        for (int k = 0; k < 3; ++k) {
            uint32_t n = 0;
            for (uint32_t q = 0; q < R3N_SUBQ; ++q) n += sc.n[list][k][q];
            // cull.wgsl:47-73: vertex_count = 3 per appended triangle, instance_count 1, base_index = first slot * 3
            calls[list * 3 + k] = {n * 3u, 1u, counts.region_base[k] * 3u, 0, 0u};
        }
    return R3N_OK;
}

int r3n_readback_triangle_sets(r3n_ctx *c, r3n_camera cam, uint8_t *pass, uint8_t *residual, uint64_t n) {
    CamState *s = c ? find_cam(c, cam, false) : nullptr;
    if (!s || s->last < 0) return fail(c, R3N_ERR_STATE, "readback_triangle_sets: camera never culled");
    if (n < c->total_tris) return fail(c, R3N_ERR_INVALID_ARG, "readback_triangle_sets: buffer too small");
    const int idx = s->last;
    const uint32_t cap = c->capacity;
    r3n_cull_counts counts;
    TRY(d2h(c, &counts, s->counts[idx].p, sizeof counts));
    std::vector<uint32_t> slot_base(cap), tri_base(cap);
    TRY(d2h(c, slot_base.data(), s->slot_base[idx].p, (size_t)cap * 4));
    TRY(d2h(c, tri_base.data(), c->tri_base.p, (size_t)cap * 4));
    std::vector<unsigned long long> mask(std::max(1u, counts.total_waves));
    if (counts.total_waves) TRY(d2h(c, mask.data(), s->mask[idx].p, (size_t)counts.total_waves * 8));
    if (pass) {
        std::memset(pass, 0, n);
        for (uint32_t o = 0; o < cap; ++o) {
            if (slot_base[o] == R3N_INVALID) continue;
            for (uint32_t t = 0; t < c->h_ntri[o]; ++t) {
                const uint64_t bit = (uint64_t)slot_base[o] + t;
                pass[(uint64_t)tri_base[o] + t] = (uint8_t)((mask[bit / 64] >> (bit % 64)) & 1ull);
            }
        }
    }
    if (residual) {
        std::memset(residual, 0, n);
        if (cam == R3N_CAMERA_VIEWPORT) {
            r3n_sub_counts sc;
            TRY(d2h(c, &sc, s->sub_counts[idx].p, sizeof sc));
            for (uint32_t k = 0; k < 3; ++k)
                for (uint32_t q = 0; q < R3N_SUBQ; ++q) {
                    const uint32_t cnt = sc.n[1][k][q];
                    if (!cnt) continue;
                    std::vector<r3n_tri_ref> refs(cnt);
                    TRY(d2h(c, refs.data(), s->residual.as<r3n_tri_ref>() + (size_t)(k * R3N_SUBQ + q) * s->subcap[idx],
                            (size_t)cnt * sizeof(r3n_tri_ref)));
                    for (const auto &r : refs) residual[(uint64_t)tri_base[r.object] + r.triangle] = 1;
                }
        }
    }
    return R3N_OK;
}

int r3n_readback_raster_stats(r3n_ctx *c, uint32_t big_items[64]) {
    if (!c || !big_items) return R3N_ERR_INVALID_ARG;
    // entries [0..16): the viewport's work queue; [16 + 12*k ..): shadow work queue k
    std::vector<uint32_t> raw(64 * R3N_BIGQ);
    for (int f = 0; f < 64; ++f) big_items[f] = 0;
    for (int lane = 0; lane < 1 + R3N_QLANES; ++lane) {
        TRY(d2h(c, raw.data(), c->big_count[lane].p, raw.size() * 4));
        const int base = lane == 0 ? 0 : 16 + 12 * (lane - 1), n = lane == 0 ? 16 : 12;
        for (int f = 0; f < n; ++f)
            for (int q = 0; q < R3N_BIGQ; ++q) big_items[base + f] += raw[f * R3N_BIGQ + q];
    }
    return R3N_OK;
}

int r3n_readback_baked(r3n_ctx *c, r3n_camera cam, float *out, uint32_t capacity) {
    CamState *s = c ? find_cam(c, cam, false) : nullptr;
    if (!s || !s->baked.p || !out) return fail(c, R3N_ERR_STATE, "readback_baked: camera never baked");
    if (capacity < c->capacity) return fail(c, R3N_ERR_INVALID_ARG, "readback_baked: buffer too small");
    return d2h(c, out, s->baked.p, (size_t)c->capacity * sizeof(r3n_baked128));
}

int r3n_readback_mesh(r3n_ctx *c, uint64_t byte_offset, void *dst, uint64_t bytes) {
    if (!c || !dst || byte_offset + bytes > c->mesh.bytes) return fail(c, R3N_ERR_INVALID_ARG, "readback_mesh: range outside the mesh buffer");
    return d2h(c, dst, static_cast<char *>(c->mesh.p) + byte_offset, bytes);
}

int r3n_readback_joint_matrices(r3n_ctx *c, uint32_t first, float *dst, uint32_t n) {
    if (!c || !dst || ((uint64_t)first + n) * 64 > c->skin_matrices.bytes) return fail(c, R3N_ERR_INVALID_ARG, "readback_joint_matrices: range outside the matrix buffer");
    return d2h(c, dst, c->skin_matrices.as<float>() + (size_t)first * 16, (size_t)n * 64);
}

int r3n_readback_texels(r3n_ctx *c, uint64_t first_texel, uint32_t *rgba8, uint64_t n_texels) {
    if (!c || !rgba8 || (first_texel + n_texels) * 4 > c->tex_texels.bytes) return fail(c, R3N_ERR_INVALID_ARG, "readback_texels: range outside the texel pool");
    return d2h(c, rgba8, c->tex_texels.as<uint32_t>() + first_texel, n_texels * 4);
}

int r3n_readback_visibility(r3n_ctx *c, uint64_t *keys) {
    if (!c || !c->vis.p || !keys) return fail(c, R3N_ERR_STATE, "readback_visibility: no frame");
    return d2h(c, keys, c->vis.p, (size_t)c->width * c->height * c->samples * 8);
}

int r3n_readback_depth(r3n_ctx *c, float *depth) {
    if (!c || !c->vis.p || !depth) return fail(c, R3N_ERR_STATE, "readback_depth: no frame");
    const size_t n = (size_t)c->width * c->height, S = c->samples;
    std::vector<uint64_t> keys(n * S);
    TRY(d2h(c, keys.data(), c->vis.p, n * S * 8));
    for (size_t i = 0; i < n; ++i) {
        if (S == 1) {
            const uint32_t zb = (uint32_t)(keys[i] >> 32);
            std::memcpy(depth + i, &zb, 4);
        } else {  // resolve_depth_min.wgsl:19-27
            float nearest = 1.0f;
            for (size_t sm = 0; sm < S; ++sm) {
                const uint32_t zb = (uint32_t)(keys[i * S + sm] >> 32);
                float z;
                std::memcpy(&z, &zb, 4);
                nearest = std::fmin(nearest, z);
            }
            depth[i] = nearest;
        }
    }
    return R3N_OK;
}

int r3n_readback_hiz(r3n_ctx *c, float *pyr, uint64_t count) {
    if (!c || !c->hiz.p || !pyr) return fail(c, R3N_ERR_STATE, "readback_hiz: no frame");
    if (count < hiz_elements(c->hizd)) return fail(c, R3N_ERR_INVALID_ARG, "readback_hiz: buffer too small");
    return d2h(c, pyr, c->hiz.p, hiz_elements(c->hizd) * 4);
}

int r3n_readback_shadow_atlas(r3n_ctx *c, float *atlas) {
    if (!c || !c->atlas.p || !atlas) return fail(c, R3N_ERR_STATE, "readback_shadow_atlas: no frame");
    return d2h(c, atlas, c->atlas.p, (size_t)c->atlas_w * c->atlas_h * 4);
}

int r3n_readback_hdr(r3n_ctx *c, uint16_t *rgba16f) {
    if (!c || !c->hdr16.p || !rgba16f) return fail(c, R3N_ERR_STATE, "readback_hdr: no frame");
    return d2h(c, rgba16f, c->hdr16.p, (size_t)c->width * c->height * 8);
}

int r3n_readback_output(r3n_ctx *c, uint8_t *rgba8, float *rgba_f32) {
    if (!c || !c->out8.p) return fail(c, R3N_ERR_STATE, "readback_output: no frame");
    const size_t n = (size_t)c->width * c->height;
    if (rgba_f32) {
        HIP_TRY(c, hipSetDevice(c->device));
        TRY(join_shade(c));
        TRY(ensure(c, c->out_f32, n * 16, false, -1));
        TRY(launch_tonemap(c, c->out_f32.as<float4>()));
        TRY(d2h(c, rgba_f32, c->out_f32.p, n * 16));
    }
    if (rgba8) TRY(d2h(c, rgba8, c->out8.p, n * 4));
    return R3N_OK;
}

// ------------------------------------------------------------------------------------------------ timing
__global__ static void k_empty() {}
int r3n_timing_enable(r3n_ctx *c, int enable) {
    if (!c) return R3N_ERR_INVALID_ARG;
    TRY(drain_timing(c));
    if (enable && !c->span_calibrated) {
        // what a timed span holds besides its kernels: the two event packets and the dispatch of a launch between them
        HIP_TRY(c, hipSetDevice(c->device));
        TRY(sync_all(c));
        hipEvent_t ea[32], eb[32];
        for (int k = 0; k < 32; ++k) { HIP_TRY(c, hipEventCreate(&ea[k])); HIP_TRY(c, hipEventCreate(&eb[k])); }
        for (int k = 0; k < 32; ++k) {
            HIP_TRY(c, hipEventRecord(ea[k], c->stream));
            hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, c->stream);
            HIP_TRY(c, hipEventRecord(eb[k], c->stream));
        }
        HIP_WAIT(c, hipStreamSynchronize(c->stream));
        std::vector<float> v;
        for (int k = 0; k < 32; ++k) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ea[k], eb[k]) == hipSuccess) v.push_back(ms);
            (void)hipEventDestroy(ea[k]); (void)hipEventDestroy(eb[k]);
        }
        std::sort(v.begin(), v.end());
        c->span_overhead_ms = v.empty() ? 0.0 : (double)v[v.size() / 2];
        c->span_calibrated = true;
    }
    c->timing = enable != 0;
    return R3N_OK;
}
int r3n_timing_overhead(r3n_ctx *c, double *ms_per_span) {
    if (!c || !ms_per_span) return R3N_ERR_INVALID_ARG;
    *ms_per_span = c->span_overhead_ms;
    return R3N_OK;
}

int r3n_set_multi_stream(r3n_ctx *c, int enable) {
    if (!c || c->in_frame) return fail(c, R3N_ERR_STATE, "set_multi_stream: only between frames");
    TRY(sync_all(c));
    c->multi_stream = enable != 0;
    return R3N_OK;
}

}  // extern "C"

// HBM copy kernels for the measured roofline denominator.  Variants, best of all is reported: grid-stride with four / eight
// independent 16-byte loads in flight per thread, plain or nontemporal (streaming: no reuse, keep L2 / Infinity Cache lines out
// of the way of the opposite stream), and a block-contiguous walk (each workgroup streams its own contiguous span, which
// keeps a DRAM page open per workgroup instead of striding the whole buffer).
typedef float f4v __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) static void k_copy_f4(const f4v *__restrict__ src, f4v *__restrict__ dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256u;
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    for (; i + (size_t)(UNROLL - 1) * stride < n; i += (size_t)UNROLL * stride) {
        f4v v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + (size_t)k * stride) : src[i + (size_t)k * stride];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            if (NT) __builtin_nontemporal_store(v[k], dst + i + (size_t)k * stride);
            else dst[i + (size_t)k * stride] = v[k];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
template <bool NT>
__global__ __launch_bounds__(256) static void k_copy_f4_span(const f4v *__restrict__ src, f4v *__restrict__ dst, size_t n) {
    const size_t per = (n + gridDim.x - 1u) / gridDim.x;  // float4s per workgroup, contiguous
    const size_t b = (size_t)blockIdx.x * per, e = b + per < n ? b + per : n;
    size_t i = b + threadIdx.x;
    for (; i + 768u < e; i += 1024u) {
        f4v v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + 256u * k) : src[i + 256u * k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (NT) __builtin_nontemporal_store(v[k], dst + i + 256u * k);
            else dst[i + 256u * k] = v[k];
        }
    }
    for (; i < e; i += 256u) dst[i] = src[i];
}

extern "C" {

int r3n_hbm_copy_rate(r3n_ctx *c, uint64_t bytes, uint32_t repeats, double *gb_per_s) {
    if (!c || !gb_per_s || bytes < (64ull << 20) || repeats == 0) return fail(c, R3N_ERR_INVALID_ARG, "hbm_copy_rate: bytes >= 64 MiB, repeats >= 1");
    HIP_TRY(c, hipSetDevice(c->device));
    TRY(sync_all(c));
    void *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const size_t n = bytes / 16;
    int rc = R3N_OK;
    float best = 0.0f;
    if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess || hipMemsetAsync(a, 1, n * 16, c->stream) != hipSuccess) {
        rc = fail(c, R3N_ERR_HIP, "hbm_copy_rate: scratch allocation failed");
    } else {
        const f4v *src = (const f4v *)a;
        f4v *dst = (f4v *)b;
        const unsigned grids[4] = {256u * 4u, 256u * 8u, 256u * 16u, 256u * 32u};
        const int kVariants = 4 * 6 + 1;  // 6 kernels x 4 grids + the runtime's own copy
        for (uint32_t r = 0; r <= (uint32_t)kVariants * repeats && rc == R3N_OK; ++r) {  // first pass untimed (page mapping, clocks)
            const int v = (int)(r % (uint32_t)kVariants);
            (void)hipEventRecord(e0, c->stream);
            if (v == kVariants - 1) (void)hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, c->stream);
            else {
                const dim3 g(grids[v % 4]), t(256);
                switch (v / 4) {
                    case 0: hipLaunchKernelGGL((k_copy_f4<4, false>), g, t, 0, c->stream, src, dst, n); break;
                    case 1: hipLaunchKernelGGL((k_copy_f4<4, true>), g, t, 0, c->stream, src, dst, n); break;
                    case 2: hipLaunchKernelGGL((k_copy_f4<8, false>), g, t, 0, c->stream, src, dst, n); break;
                    case 3: hipLaunchKernelGGL((k_copy_f4<8, true>), g, t, 0, c->stream, src, dst, n); break;
                    case 4: hipLaunchKernelGGL((k_copy_f4_span<false>), g, t, 0, c->stream, src, dst, n); break;
                    default: hipLaunchKernelGGL((k_copy_f4_span<true>), g, t, 0, c->stream, src, dst, n); break;
                }
            }
            (void)hipEventRecord(e1, c->stream);
            float ms = 0.0f;
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = fail(c, R3N_ERR_HIP, "hbm_copy_rate: timing failed");
            else if (r > 0 && (best == 0.0f || ms < best)) {
                best = ms;
                c->hbm_best_variant = v;
            }
        }
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (rc == R3N_OK) *gb_per_s = best > 0.0f ? 2.0 * (double)(n * 16) / ((double)best * 1e-3) / 1e9 : 0.0;
    if (rc == R3N_OK && std::getenv("R3N_VERBOSE")) std::fprintf(stderr, "r3n_hbm_copy_rate: best variant %d (kernel %d, grid %u): %.1f GB/s\n", c->hbm_best_variant, c->hbm_best_variant / 4, c->hbm_best_variant < 24 ? 256u * (4u << (c->hbm_best_variant % 4)) : 0u, *gb_per_s);
    return rc;
}

int r3n_stage_times(r3n_ctx *c, double ms[R3N_STAGE_COUNT], uint64_t launches[R3N_STAGE_COUNT], int reset) {
    if (!c) return R3N_ERR_INVALID_ARG;
    TRY(drain_timing(c));
    for (int i = 0; i < R3N_STAGE_COUNT; ++i) {
        if (ms) ms[i] = c->stage_ms[i];
        if (launches) launches[i] = c->stage_launches[i];
        if (reset) { c->stage_ms[i] = 0; c->stage_launches[i] = 0; }
    }
    return R3N_OK;
}

}  // extern "C"
