// layouts.h -- byte layouts shared by the HIP kernels and the C-ABI shim (product-internal).
// rend3's encase/std430 layouts are restated in include/r3n.h; the static_asserts pin them
// (SURVEY.md App. A).  Everything else here is this implementation's own HBM data layout
// (DESIGN.md "Data layout in HBM").
#pragma once
#include <cstddef>
#include <cstdint>

#include "../../include/r3n.h"

static_assert(sizeof(r3n_object128) == 128, "Object record must be 128 B (object.rs:23-36)");
static_assert(offsetof(r3n_object128, bounding_sphere_center) == 64, "sphere @64");
static_assert(offsetof(r3n_object128, first_index) == 80, "first_index @80");
static_assert(offsetof(r3n_object128, vertex_attribute_start_offsets) == 92, "attr offsets @92");
static_assert(offsetof(r3n_object128, enabled) == 116, "enabled @116");
static_assert(sizeof(r3n_material208) == 208, "material record must be 208 B");
static_assert(offsetof(r3n_material208, albedo) == 144, "albedo @144");
static_assert(offsetof(r3n_material208, flags) == 204, "flags @204");
static_assert(sizeof(r3n_texture_desc32) == 32, "texture descriptor must be 32 B");
static_assert(sizeof(r3n_camera_header240) == 240, "PerCameraUniform header must be 240 B");
static_assert(offsetof(r3n_camera_header240, frustum) == 144, "frustum @144");
static_assert(offsetof(r3n_camera_header240, flags) == 232, "flags @232");
static_assert(sizeof(r3n_frame_uniforms496) == 496, "FrameUniforms must be 496 B");
static_assert(sizeof(r3n_indirect_call) == 20, "IndirectCall must be 20 B");
// the one-call frame boundary (r3n_render_frame / r3n_host_evaluate_frame): sizes the ctypes / Rust mirrors repeat
static_assert(sizeof(r3n_shadow_view272) == 272, "shadow view = 240-byte header + viewport");
static_assert(sizeof(r3n_frame_desc) == 152 && offsetof(r3n_frame_desc, uniforms) == 48 && offsetof(r3n_frame_desc, exchange) == 136, "r3n_frame_desc layout");
static_assert(sizeof(r3n_host_camera144) == 144, "host camera inputs");
static_assert(sizeof(r3n_host_directional_light48) == 48, "host directional light");
static_assert(sizeof(r3n_host_frame) == 26712 && offsetof(r3n_host_frame, shadow_views) == 832 && offsetof(r3n_host_frame, directional_buffer) == 18504, "r3n_host_frame layout");

#define R3N_INVALID 0xFFFFFFFFu

// material.wgsl:1-15
#define R3N_FLAGS_ALBEDO_ACTIVE 0x0001u
#define R3N_FLAGS_ALBEDO_BLEND 0x0002u
#define R3N_FLAGS_ALBEDO_VERTEX_SRGB 0x0004u
#define R3N_FLAGS_BICOMPONENT_NORMAL 0x0008u
#define R3N_FLAGS_SWIZZLED_NORMAL 0x0010u
#define R3N_FLAGS_YDOWN_NORMAL 0x0020u
#define R3N_FLAGS_AOMR_COMBINED 0x0040u
#define R3N_FLAGS_AOMR_SWIZZLED_SPLIT 0x0080u
#define R3N_FLAGS_AOMR_SPLIT 0x0100u
#define R3N_FLAGS_AOMR_BW_SPLIT 0x0200u
#define R3N_FLAGS_CC_GLTF_COMBINED 0x0400u
#define R3N_FLAGS_CC_GLTF_SPLIT 0x0800u
#define R3N_FLAGS_CC_BW_SPLIT 0x1000u
#define R3N_FLAGS_UNLIT 0x2000u
#define R3N_FLAGS_NEAREST 0x4000u
// structures.wgsl:64-72
#define R3N_PCU_POSITIVE_AREA_VISIBLE 0x1u
#define R3N_PCU_MULTISAMPLED 0x2u

// PerCameraUniformObjectData (culler.rs:177-183): what uniform_prep.wgsl writes per object.
struct r3n_baked128 {
    float model_view[16];
    float model_view_proj[16];
};
static_assert(sizeof(r3n_baked128) == 128, "baked record must be 128 B");

// Directional light record, 128 B stride (directional.rs:38-53 / structures.wgsl:74-88)
struct r3n_dir_light128 {
    float view_proj[16];
    float color[3];
    float _p0;
    float direction[3];
    float _p1;
    float inv_resolution[2];
    float atlas_offset[2];
    float atlas_size[2];
    float _p2[2];
};
static_assert(sizeof(r3n_dir_light128) == 128, "directional light stride must be 128 B");
// Point light record, 32 B (point.rs:21-26)
struct r3n_point_light32 {
    float position[4];
    float color[3];
    float radius;
};
static_assert(sizeof(r3n_point_light32) == 32, "point light stride must be 32 B");

// ---- this implementation's own device structures ------------------------------------------------
// One entry per frustum-visible object, in object-slot order.  Work is distributed in "wave slots":
// object e owns wave slots [wave_start, next.wave_start), 64 triangles each, so a wavefront never
// straddles two objects and its matrices / material key are wave-uniform (scalar loads).
struct r3n_vis_entry {
    uint32_t object;
    uint32_t wave_start;
};

// Per-camera counters produced by the object pass (device resident; the host never reads them on the hot path).
struct r3n_cull_counts {
    uint32_t visible_objects;
    uint32_t total_waves;
    uint32_t total_triangles;     // all enabled objects (canonical slot space)
    uint32_t key_triangles[3];    // visible triangles per material key = region capacities
    uint32_t region_base[3];      // first list entry of each region
    uint32_t _pad[3];
};

// One compacted triangle reference (the "index buffer" of this implementation: 8 B instead of the
// reference's 3 packed u32 because the rasteriser re-fetches indices through the object record).
struct alignas(8) r3n_tri_ref {  // 8-byte aligned: one dwordx2 load / store per entry
    uint32_t object;
    uint32_t triangle;
};

// Raster work item for triangles larger than 8x8 px: one wavefront scans a <=64x64 px region.  The item carries
// the finished triangle setup (the producer computed it to classify the triangle), so the consumer has no
// dependent gather chain in front of its scan: one 64-byte record (one s_load_dwordx16), broadcast through SGPRs.
struct r3n_big_item {
    float e[3][3];      // oriented edge functions
    float z[3];         // the depth plane: (z[0] * x + z[1] * y) + z[2] (device_math.h setup_triangle)
    uint32_t slot1;     // canonical slot + 1 (forward only)
    uint32_t material;  // material index (cutout key) / draw order (blend) | the edge thresholds' bits << R3N_BIG_THR_SHIFT
    uint32_t xy0;       // x0 | y0 << 16
    uint32_t xy1;       // x1 | y1 << 16 (inclusive)
};
static_assert(sizeof(r3n_big_item) == 64, "big item is 16 dwords: one s_load_dwordx16, one cache line");
// Same index as the item; written and read only for cutout triangles (vertex alpha; the uvs when the alpha comes from the
// albedo texture).
struct r3n_big_uv {
    float uv[3][2];
    float va[3];        // vertex alpha
    uint32_t _pad[3];
};
static_assert(sizeof(r3n_big_uv) == 48, "big item cutout record is 12 dwords");

// Output lists and work queues are split into sub-queues so that appends do not serialise on one counter: a
// returning atomic on a single address retires at only ~88 per microsecond on MI355X (MI355X_MICROARCH.md,
// "dequeue" row), which was the bound of the cull kernel with a single counter per list.
#define R3N_SUBQ 32   // sub-lists per (list, material key) of the cull output
#define R3N_BIGQ 32   // sub-queues of the rasteriser's large-triangle work queue; the consumers index their
                      // concatenation, so the split only spreads the producers' counter traffic
struct r3n_sub_counts {
    uint32_t n[2][3][R3N_SUBQ];  // [predicted|residual][material key][sub-list] = triangles appended
};

#define R3N_SLOT_TABLE_SHIFT 8  // canonical triangle slots per bucket of the slot -> object table

#define R3N_MAX_HIZ_MIPS 16
#define R3N_TEX_LEVELS 16  // extents <= 65535: at most 16 levels per texture (level-offset table of the sampler)
// device-side r3n_texture_desc32.format: 0 / 1 = RGBA8 texels (unorm / sRGB), one pool word each; R3N_POOL_FLOAT = four f32 per
// texel (r3n_textures_write_encoded puts every non-8-bit-unorm format there); `offset` counts pool words either way
#define R3N_POOL_FLOAT 2u
struct r3n_hiz_desc {
    uint32_t width, height, mips, _pad;
    uint32_t offset[R3N_MAX_HIZ_MIPS];  // element offset of each mip
};

#define R3N_MAX_DIR_LIGHTS 16
#define R3N_MAX_POINT_LIGHTS 256
