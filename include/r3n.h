/*
 * r3n.h -- C ABI of the MI355X-native rend3 object pipeline (librend3_amd.so).
 *
 * rend3 has no FFI boundary of its own (100 % Rust + WGSL); its extension API is the set of
 * `add_*_to_graph` routine methods whose node bodies record wgpu work.  This header is the
 * boundary those node bodies would call instead.  Each entry point cites the reference
 * interface it replaces; INTEGRATION.md shows the Rust `extern "C"` block and the adaptor
 * structs (BaseRenderGraph / GpuCuller / ForwardRoutine / HiZRoutine / TonemappingRoutine)
 * a maintainer would add on the rend3 side.
 *
 * Conventions
 *   - opaque context, one HIP device + one HIP stream per context;
 *   - every call returns 0 on success, <0 on error (never unwinds); r3n_last_error() has text;
 *   - the caller owns every host pointer for the duration of the call only; the context owns
 *     all device memory; all calls are asynchronous on the context's stream except r3n_readback_*,
 *     r3n_sync and calls documented as synchronising;
 *   - byte layouts are rend3's own (encase std430), restated in SURVEY.md App. A and checked by
 *     static_asserts in rend3_amd/csrc/layouts.h;
 *   - calls are made from one thread at a time (the reference holds the data_core mutex for the
 *     whole graph execution, rend3/src/graph/graph.rs:265).
 */
#ifndef R3N_H
#define R3N_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R3N_OK 0
#define R3N_ERR_INVALID_ARG (-1)
#define R3N_ERR_HIP (-2)
#define R3N_ERR_NO_DEVICE (-3)
#define R3N_ERR_STATE (-4)
#define R3N_ERR_UNSUPPORTED (-5)
#define R3N_ERR_CAPACITY (-6) /* a fixed-size internal buffer (raster work queue, blend fragment list) overflowed.  Raised by kernels, so it
                                * surfaces later: r3n_frame_end / r3n_render_frame / r3n_sync / a read-back return it ONCE for an EARLIER frame
                                * whose image is incomplete; the frame just closed is unaffected, and no frame is ever refused because of it */

/* CameraSpecifier (rend3-routine/src/common/camera.rs:3-35): shadow index, or R3N_CAMERA_VIEWPORT
 * (== CameraSpecifier::Viewport.to_shader_index() == u32::MAX). */
typedef uint32_t r3n_camera;
#define R3N_CAMERA_VIEWPORT 0xFFFFFFFFu
#define R3N_MAX_SHADOW_VIEWS 64u

/* RoutineType (rend3-routine/src/forward.rs:40-44) */
#define R3N_PASS_DEPTH 0u
#define R3N_PASS_FORWARD 1u
/* CullingSource (rend3-routine/src/forward.rs:64-70) */
#define R3N_SOURCE_PREDICTED 0u
#define R3N_SOURCE_RESIDUAL 1u
/* Material::key() of PbrMaterial = TransparencyType (rend3-routine/src/pbr/material.rs:383-392,497-499) */
#define R3N_KEY_OPAQUE 0u
#define R3N_KEY_CUTOUT 1u
#define R3N_KEY_BLEND 2u

typedef struct r3n_ctx r3n_ctx;

/* ShaderObject<PbrMaterial>, 128 B (rend3/src/managers/object.rs:23-36) */
typedef struct r3n_object128 {
    float transform[16];
    float bounding_sphere_center[3];
    float bounding_sphere_radius;
    uint32_t first_index;
    uint32_t index_count;
    uint32_t material_index;
    uint32_t vertex_attribute_start_offsets[6];
    uint32_t enabled;
    uint32_t _pad[2];
} r3n_object128;

/* GpuPoweredShaderWrapper<PbrMaterial> = 10 texture ids + ShaderMaterial, 208 B
 * (rend3/src/managers/material.rs:25-29, rend3-routine/src/pbr/material.rs:526-543) */
typedef struct r3n_material208 {
    uint32_t textures[10];
    uint32_t _pad[2];
    float uv_transform0[12];
    float uv_transform1[12];
    float albedo[4];
    float emissive[3];
    float roughness;
    float metallic;
    float reflectance;
    float clear_coat;
    float clear_coat_roughness;
    float anisotropy;
    float ambient_occlusion;
    float alpha_cutout;
    uint32_t flags;
} r3n_material208;

/* One entry of the bindless texture array (rend3/src/managers/texture.rs; Texture in rend3-types/src/lib.rs:
 * data, format, size, mip_count).  For r3n_textures_write: RGBA8 texels, `offset` = first texel (u32) of mip 0 in the texel
 * pool, the mips follow contiguously; material records refer to entry i as texture id i + 1 (0 = none). */
typedef struct r3n_texture_desc32 {
    uint32_t offset;
    uint32_t width, height, mips;
    uint32_t format; /* R3N_TEXTURE_* */
    uint32_t stored_mips; /* r3n_textures_write_encoded only: 0 = all `mips` levels are in the payload (MipmapSource::Uploaded);
                             k < mips = the first k levels are, the others are GENERATED on the GPU (MipmapSource::Generated,
                             rend3/src/util/mipmap.rs + mipmap.wgsl: Linear / ClampToEdge blit per level in the texture's
                             format); uncompressed 8-bit formats, 16-bit float formats and RGB10A2 only */
    uint32_t _pad[2];
} r3n_texture_desc32;
#define R3N_TEXTURE_RGBA8_UNORM 0u
#define R3N_TEXTURE_RGBA8_UNORM_SRGB 1u
/* formats accepted by r3n_textures_write_encoded only (rend3-types TextureFormat names; what rend3-gltf's
 * util::map_ktx2_format / map_dxgi_format / map_d3d_format / convert_dynamic_image produce for 8-bit colour data,
 * rend3-gltf/src/lib.rs:1157-1175,1176-1410).  R / RG read (r, 0, 0, 1) / (r, g, 0, 1) like a sampled texture. */
#define R3N_TEXTURE_R8_UNORM 2u
#define R3N_TEXTURE_RG8_UNORM 3u
#define R3N_TEXTURE_BGRA8_UNORM 4u
#define R3N_TEXTURE_BGRA8_UNORM_SRGB 5u
#define R3N_TEXTURE_BC1_RGBA_UNORM 6u
#define R3N_TEXTURE_BC1_RGBA_UNORM_SRGB 7u
#define R3N_TEXTURE_BC2_RGBA_UNORM 8u
#define R3N_TEXTURE_BC2_RGBA_UNORM_SRGB 9u
#define R3N_TEXTURE_BC3_RGBA_UNORM 10u
#define R3N_TEXTURE_BC3_RGBA_UNORM_SRGB 11u
#define R3N_TEXTURE_BC4_R_UNORM 12u
#define R3N_TEXTURE_BC5_RG_UNORM 13u
#define R3N_TEXTURE_BC7_RGBA_UNORM 14u
#define R3N_TEXTURE_BC7_RGBA_UNORM_SRGB 15u
/* formats whose values are not 8-bit unorm: decoded into four f32 per texel (16 B per texel in the pool instead of 4).
 * Missing channels read (0, 0, 1).  stored_mips < mips (generated levels) is accepted for R16 / RG16 / RGBA16_FLOAT and
 * RGB10A2_UNORM -- the ones rend3-gltf generates chains for (filterable render targets; the level is rounded to the format:
 * binary16 to nearest even); the others must carry every level.  rend3-gltf/src/lib.rs:1204-1330 (KTX2), :1480-1485 (D3D),
 * :1499-1597 (DXGI) produce them. */
#define R3N_TEXTURE_R8_SNORM 16u
#define R3N_TEXTURE_RG8_SNORM 17u
#define R3N_TEXTURE_RGBA8_SNORM 18u
#define R3N_TEXTURE_R16_FLOAT 19u
#define R3N_TEXTURE_RG16_FLOAT 20u
#define R3N_TEXTURE_RGBA16_FLOAT 21u
#define R3N_TEXTURE_R32_FLOAT 22u
#define R3N_TEXTURE_RG32_FLOAT 23u
#define R3N_TEXTURE_RGBA32_FLOAT 24u
#define R3N_TEXTURE_RGBA16_UNORM 25u
#define R3N_TEXTURE_RGBA16_SNORM 26u
#define R3N_TEXTURE_RGB10A2_UNORM 27u
#define R3N_TEXTURE_RG11B10_FLOAT 28u
#define R3N_TEXTURE_RGB9E5_UFLOAT 29u
#define R3N_TEXTURE_BC4_R_SNORM 30u
#define R3N_TEXTURE_BC5_RG_SNORM 31u
#define R3N_TEXTURE_BC6H_RGB_UFLOAT 32u
#define R3N_TEXTURE_BC6H_RGB_FLOAT 33u
#define R3N_TEXTURE_FORMAT_COUNT 34u

/* PerCameraUniform header, 240 B (rend3-routine/src/culling/culler.rs:158-175) */
typedef struct r3n_camera_header240 {
    float view[16];
    float view_proj[16];
    uint32_t shadow_index;
    uint32_t _pad[3];
    float frustum[20];
    float resolution[2];
    uint32_t flags; /* bit0 positive area visible, bit1 multisampled (culler.rs:151-156) */
    uint32_t object_count;
} r3n_camera_header240;

/* FrameUniforms, 496 B (rend3-routine/src/uniforms.rs:17-27) */
typedef struct r3n_frame_uniforms496 {
    float view[16];
    float view_proj[16];
    float origin_view_proj[16];
    float inv_view[16];
    float inv_view_proj[16];
    float inv_origin_view_proj[16];
    float frustum[20];
    float ambient[4];
    uint32_t resolution[2];
    uint32_t _pad[2];
} r3n_frame_uniforms496;

/* wgpu DrawIndexedIndirect as written by cull.wgsl:47-61 (structures.wgsl:20-26), 20 B.
 * One per region; here a region is one material key. */
typedef struct r3n_indirect_call {
    uint32_t vertex_count;
    uint32_t instance_count;
    uint32_t base_index;
    int32_t vertex_offset;
    uint32_t base_instance;
} r3n_indirect_call;

/* GpuSkinningInput, 40 B (rend3-routine/src/skinning.rs:23-46, shaders/src/skinning.wgsl:3-25).  Offsets are byte
 * offsets into the mesh buffer; 0xFFFFFFFF = attribute absent. */
typedef struct r3n_skinning_input40 {
    uint32_t base_position_offset, base_normal_offset, base_tangent_offset;
    uint32_t joint_indices_offset, joint_weight_offset;
    uint32_t updated_position_offset, updated_normal_offset, updated_tangent_offset;
    uint32_t joint_matrix_base_offset, vertex_count;
} r3n_skinning_input40;

/* ---- rend3-anim (rend3-anim/src/lib.rs) tables, row N4.  A RIG is one skin: its joints in the skin's order, each with
 * its parent joint and its depth in the joint hierarchy (AnimationData::from_gltf_scene, :77-145, flattened); a CLIP is
 * one animation applied to one rig: a track per joint with the key ranges of its translation / rotation / scale
 * channels and the bind components used where a channel is absent (pose_animation_frame, :226-236). */
typedef struct r3n_anim_rig16 {
    uint32_t first_joint, n_joints; /* into the joints array; n_joints <= 512 */
    uint32_t max_depth, _pad;
} r3n_anim_rig16;
typedef struct r3n_anim_joint80 {
    int32_t parent;  /* joint index inside the rig; -1: the node has no parent (global = local); -2: its parent node is
                        not a joint of this skin (global = IDENTITY * local), rend3-anim/src/lib.rs:246-256 */
    uint32_t depth;  /* 0 for parent < 0, else depth(parent) + 1 */
    uint32_t _pad[2];
    float inverse_bind[16];
} r3n_anim_joint80;
typedef struct r3n_anim_clip16 {
    uint32_t rig, first_track; /* tracks [first_track, first_track + rig.n_joints) */
    float duration;            /* Animation::duration: the time is clamped to [0, duration] (:189) */
    uint32_t _pad;
} r3n_anim_clip16;
typedef struct r3n_anim_track80 {
    uint32_t animated;        /* 0: the clip has no channel for this joint's node -> local matrix IDENTITY (:220); else nonzero */
    uint32_t key_first[3];    /* translation, rotation, scale: first key time (index into `times`) */
    uint32_t key_count[3];    /* 0: channel absent -> bind component */
    uint32_t value_first[3];  /* first value (index into `values`: 3 floats per vec3 key, 4 per quaternion key, xyzw) */
    float bind_t[3], bind_r[4], bind_s[3]; /* Mat4::to_scale_rotation_translation of the node's local transform */
} r3n_anim_track80;
typedef struct r3n_pose_request16 {
    uint32_t clip;
    float time;
    uint32_t matrix_base; /* first joint matrix of the skeleton (GpuSkinningInput.joint_matrix_base_offset) */
    uint32_t _pad;
} r3n_pose_request16;

/* Arithmetic of the PBR fragment stage (r3n_resolve_opaque).  EXACT (default): every f32 operation rounds once, IEEE division and
 * square root -- the contract under which visibility, HDR and framebuffer are bit-identical to the CPU oracle.  FAST (opt-in):
 * fused multiply-add and hardware reciprocal / rsqrt in interpolation, texture filtering and BRDF; coverage, depth, level of
 * detail, shadow coordinates and comparisons stay exact.  The framebuffer then stays within the north-star tolerance
 * (|delta| <= 1e-3 after tonemap), not bit-identical.  Honoured by the single-sample resolve; MSAA always runs EXACT. */
#define R3N_SHADE_EXACT 0u
#define R3N_SHADE_FAST 1u

typedef struct r3n_config {
    uint32_t struct_size;      /* sizeof(r3n_config) */
    uint32_t max_big_items;    /* raster work-queue capacity (0 = default 4 Mi items) */
    uint32_t shade_mode;       /* R3N_SHADE_EXACT | R3N_SHADE_FAST */
    uint32_t _pad;
    uint64_t reserved[2];
} r3n_config;

/* ---- lifetime (replaces rend3::create_iad + BaseRenderGraph::new / PbrRoutine::new state:
 *      rend3-routine/src/base.rs:111-124, culling/culler.rs:198-425) */
r3n_ctx *r3n_create(int hip_device, const r3n_config *config);
void r3n_destroy(r3n_ctx *ctx);
const char *r3n_last_error(const r3n_ctx *ctx);
/* text of the last error of a failed r3n_create (ctx == NULL) */
const char *r3n_create_error(void);
int r3n_sync(r3n_ctx *ctx);
/* the context's hipStream_t, so callers (torch, RCCL) can order work against it */
void *r3n_stream(r3n_ctx *ctx);

/* ---- world data uploads (the managers' GPU buffers the hot path reads)
 *      MeshManager::add upload, rend3/src/managers/mesh.rs:123-184 (layout: SoA attribute runs + u32 indices) */
int r3n_mesh_buffer_write(r3n_ctx *ctx, uint64_t byte_offset, const void *data, uint64_t bytes);
/*      ObjectManager::evaluate scatter upload, rend3/src/managers/object.rs:344-364.
 *      `capacity` = object-buffer capacity in records (FreelistDerivedBuffer, util/freelist/buffer.rs:48-52);
 *      growing it preserves existing records, new records are zero (enabled == 0). */
int r3n_objects_write(r3n_ctx *ctx, const uint32_t *slots, const r3n_object128 *records, uint32_t n,
                      uint32_t capacity);
/*      MaterialManager::evaluate, rend3/src/managers/material.rs:202-227.  `keys[i]` = Material::key()
 *      (R3N_KEY_*), which is host-side state in the reference (not part of the 208-byte record). */
int r3n_materials_write(r3n_ctx *ctx, const uint32_t *slots, const r3n_material208 *records,
                        const uint8_t *keys, uint32_t n);
/*      TextureManager (rend3/src/managers/texture.rs): replaces the whole bindless 2D texture array.  Material records
 *      sample entry `id - 1` (opaque.wgsl:151-160).  Formats other than RGBA8 -> R3N_ERR_UNSUPPORTED. */
int r3n_textures_write(r3n_ctx *ctx, const r3n_texture_desc32 *descs, uint32_t n_textures, const uint32_t *texels,
                       uint64_t n_texels);
/*      Same, for textures in the formats the loader produces (above): `payload` holds every texture's levels back to
 *      back in ITS OWN format (block-compressed levels: ceil(w/4) x ceil(h/4) blocks, 8 or 16 B each), desc.offset =
 *      BYTE offset of level 0 in `payload`.  The library expands / decodes everything on the GPU into its RGBA8 texel
 *      pool (the texture unit's job in the reference: formats go straight to wgpu, rend3/src/managers/texture.rs:59-
 *      150).  Snorm / float / 16-bit / ETC2 / ASTC / BC6H formats -> R3N_ERR_UNSUPPORTED. */
int r3n_textures_write_encoded(r3n_ctx *ctx, const r3n_texture_desc32 *descs, uint32_t n_textures, const void *payload,
                               uint64_t payload_bytes);
/*      Draw order of the blend-key objects for the transparent pass: object slots back to front, as the CPU batcher
 *      sorts them every frame (rend3-routine/src/culling/batching.rs:146-176, Sorting::BLENDING: -distance^2 from the
 *      camera location to the object location).  Call once per frame before r3n_resolve_opaque (n may be 0). */
int r3n_blend_order_write(r3n_ctx *ctx, const uint32_t *objects_back_to_front, uint32_t n);
/*      DirectionalLightManager / PointLightManager buffers, byte-identical:
 *      u32 count @0, array @16 (stride 128 / 32): rend3/src/managers/directional.rs:31-53,135-153, point.rs:14-74 */
int r3n_lights_write(r3n_ctx *ctx, const void *directional_buffer, uint64_t directional_bytes,
                     const void *point_buffer, uint64_t point_bytes);

/* TonemappingRoutine::new's output_format (rend3-routine/src/tonemapping.rs:29-106): *Srgb targets take the exact OETF of
 * the fixed-function store (blit.wgsl fs_main_scene), other targets the shader's srgb_scene_to_display approximation
 * (fs_main_monitor, math/color.wgsl:13-19, exponent 0.4166); Bgra8* targets hold blue first.  Default RGBA8_UNORM_SRGB. */
#define R3N_OUTPUT_RGBA8_UNORM_SRGB 0u
#define R3N_OUTPUT_BGRA8_UNORM_SRGB 1u
#define R3N_OUTPUT_RGBA8_UNORM 2u
#define R3N_OUTPUT_BGRA8_UNORM 3u
int r3n_set_output_format(r3n_ctx *ctx, uint32_t format);
/* Skinning kernel (r3n_skinning).  EXACT (default): vector ALU, every operation rounded once -- bit-identical to the oracle's
 * restatement of skinning.wgsl:37-94.  MFMA (opt-in): the joint-matrix x vertex-block contraction on the matrix cores
 * (v_mfma_f32_16x16x4_f32: four joint matrices x 16 vertices per instruction), for rigs of at most FOUR joints
 * (R3N_ERR_UNSUPPORTED otherwise); its f32 results follow the fused-multiply-add order of the instruction, bit-identical to
 * oracle/r3o.c::r3o_skinning_mfma_order, within 1e-6 relative of EXACT. */
#define R3N_SKIN_EXACT 0u
#define R3N_SKIN_MFMA 1u
int r3n_set_skinning_mode(r3n_ctx *ctx, uint32_t mode);
/* switches the fragment-stage arithmetic (R3N_SHADE_*) for the resolves enqueued from now on */
int r3n_set_shade_mode(r3n_ctx *ctx, uint32_t mode);

/* ---- frame (node order of BaseRenderGraph::add_to_graph, rend3-routine/src/base.rs:135-185)
 * r3n_frame_begin: create_frame_uniforms (uniforms.rs:73-125) + render-target setup (base.rs:224-264) +
 * clear_shadow_buffers (clear.rs:4-20).  Clears colour to `clear_color`, depth and the shadow atlas to 0.0.
 * `samples` is SampleCount::One (1) or ::Four (4): the colour / depth targets of the viewport then hold 4 samples per pixel
 * (forward.rs:358, base.rs:236-258), Hi-Z starts from their depth-min resolve (resolve_depth_min.wgsl) and the HDR target
 * the tonemapper reads is the render pass's box resolve.  Shadow views are always single-sampled (base.rs:230).
 * width / height / samples may change from one frame to the next: the targets are re-created, the cameras' culling history (last
 * frame's result bits and predicted triangles) is KEPT, as the reference keeps a camera's culling buffers whatever the target's
 * size (CullingBufferMap is keyed by the CameraSpecifier alone, culler.rs:53-80). */
int r3n_frame_begin(r3n_ctx *ctx, const r3n_frame_uniforms496 *uniforms, uint32_t width, uint32_t height,
                    uint32_t samples, const float clear_color[4], uint32_t shadow_atlas_width,
                    uint32_t shadow_atlas_height);
/* skinning::add_skinning_to_graph (rend3-routine/src/skinning.rs:211-226): build_gpu_skinning_input_buffers (:54-139) +
 * GpuSkinner::execute_pass (:142-199) + skinning.wgsl.  `inputs`: one record per skeleton, `joint_matrices`: all
 * skeletons' joint matrices back to back (column-major mat4).  ONE launch covers every skeleton (the reference issues
 * one dispatch + one dynamic-offset bind per skeleton).  Must precede the frame's bakes (base.rs:145). */
int r3n_skinning(r3n_ctx *ctx, const r3n_skinning_input40 *inputs, uint32_t n_skeletons, const float *joint_matrices,
                 uint32_t n_joint_matrices);
/* rend3-anim on the GPU.  r3n_animation_write replaces the rig / clip tables.  r3n_pose_skeletons queues
 * pose_animation_frame's per-skin work (rend3-anim/src/lib.rs:213-262) for `n` skeleton instances: the NEXT r3n_skinning
 * evaluates them on the GPU -- after copying its host `joint_matrices` (which may be NULL when every skeleton is posed
 * this way) -- and writes each skeleton's joint matrices (global * inverse bind) at its matrix_base. */
int r3n_animation_write(r3n_ctx *ctx, const r3n_anim_rig16 *rigs, uint32_t n_rigs, const r3n_anim_joint80 *joints,
                        uint32_t n_joints, const r3n_anim_clip16 *clips, uint32_t n_clips, const r3n_anim_track80 *tracks,
                        uint32_t n_tracks, const float *times, uint32_t n_times, const float *values, uint32_t n_values);
int r3n_pose_skeletons(r3n_ctx *ctx, const r3n_pose_request16 *requests, uint32_t n);
/* GpuCuller::object_uniform_upload (culler.rs:427-529) + uniform_prep.wgsl.  Called on its own it bakes every enabled slot, like
 * the reference.  Inside r3n_render_frame the bake is fused into the object pass and covers only the slots a kernel reads: those
 * inside the frustum now or (viewport) in the camera's previous frame -- the others keep what they held. */
int r3n_uniform_bake(r3n_ctx *ctx, r3n_camera camera, const r3n_camera_header240 *header);
/* GpuCuller::add_culling_to_graph (culler.rs:682-713) = batch_objects (batching.rs:120-250, frustum cull +
 * slot assignment, done on the GPU here) + GpuCuller::cull (culler.rs:531-659) + cull.wgsl.
 * Owns the per-camera temporal state (previous-invocation map, ping-pong result bits, predicted lists). */
int r3n_cull(r3n_ctx *ctx, r3n_camera camera);
/* HiZRoutine::add_hi_z_to_graph (hi_z.rs:161-234) + hi_z.wgsl: pyramid from the viewport depth so far */
int r3n_hi_z(r3n_ctx *ctx);
/* Shadow viewport of a shadow camera inside the atlas (base.rs:369: ViewportRect(desc.map.offset, size)) */
int r3n_shadow_viewport(r3n_ctx *ctx, r3n_camera shadow_camera, uint32_t x, uint32_t y, uint32_t size);
/* ForwardRoutine::add_forward_to_graph (forward.rs:192-315) for one (camera, routine type, culling source,
 * material key): rasterises that draw-call range with depth test GreaterEqual + write (forward.rs:347-351).
 * FORWARD colour is resolved per pixel by r3n_resolve_opaque (each pixel shaded once, for its nearest fragment). */
int r3n_forward(r3n_ctx *ctx, r3n_camera camera, uint32_t pass, uint32_t source, uint32_t material_key);
/* Evaluates opaque.wgsl (VS :91-135 + FS :203-551) for the nearest fragment of every pixel -> Rgba16Float HDR.
 * Must follow the last opaque / cutout FORWARD r3n_forward of the frame (base.rs:172) and precede r3n_tonemap.
 * The transparent pass -- r3n_forward(R3N_CAMERA_VIEWPORT, R3N_PASS_FORWARD, R3N_SOURCE_RESIDUAL, R3N_KEY_BLEND),
 * base.rs:181 -- comes after it: this frame's passing triangles of the blend-key objects, in the order given to
 * r3n_blend_order_write, depth-tested against the opaque depth (no depth write) and alpha-blended into the HDR target
 * (pbr/routine.rs:113-118). */
int r3n_resolve_opaque(r3n_ctx *ctx);
/* TonemappingRoutine::add_to_graph (tonemapping.rs:108-147) + blit.wgsl into an Rgba8UnormSrgb target.
 * If `host_rgba8` is non-NULL the image is also copied out (synchronises), `pitch_bytes` per row. */
int r3n_tonemap(r3n_ctx *ctx, void *host_rgba8, uint64_t pitch_bytes);
/* The routine's `src` is any Rgba16Float target (tonemapping.rs:108-116: `src: RenderTargetHandle`), not only the
 * one r3n_resolve_opaque fills: this writes `n_pixels` RGBA half texels starting at `first_pixel` into the frame's HDR
 * target.  A following r3n_tonemap then runs the stand-alone blit kernel over the whole target. */
int r3n_hdr_write(r3n_ctx *ctx, const uint16_t *rgba16f, uint64_t first_pixel, uint64_t n_pixels);
/* Marks the end of the frame: swaps the temporal state (InputOutputBuffer::swap, culling/suballoc.rs:164-214). */
int r3n_frame_end(r3n_ctx *ctx);

/* ---- the whole frame in ONE call.  r3n_render_frame issues every node of BaseRenderGraph::add_to_graph
 * (rend3-routine/src/base.rs:129-185, executed by RenderGraph::execute, rend3/src/graph/graph.rs:265-518) in the reference's
 * order -- exactly the sequence of the per-node entry points above: r3n_frame_begin, r3n_shadow_viewport, r3n_skinning, per shadow
 * view r3n_uniform_bake / r3n_cull / r3n_forward(DEPTH, RESIDUAL, OPAQUE | CUTOUT), the viewport's r3n_uniform_bake,
 * r3n_forward(FORWARD, PREDICTED, ..), r3n_hi_z, r3n_cull, r3n_forward(FORWARD, RESIDUAL, ..), r3n_resolve_opaque, the transparent
 * r3n_forward, r3n_tonemap, r3n_frame_end.  The per-node entry points stay (a Rust integration calls them from its node closures);
 * this one is for hosts that do not need a graph between the nodes: one FFI crossing per frame instead of ~45.
 * `desc` carries what Renderer::evaluate_instructions + the node closures compute on the CPU in the reference: the FrameUniforms
 * (uniforms.rs:28-48), the viewport's PerCameraUniform header (culler.rs:485-502), one header + atlas viewport per shadow map
 * (directional.rs:99-157, base.rs:366-396), the light buffers, optionally this frame's skinning inputs. */
typedef struct r3n_shadow_view272 {
    r3n_camera_header240 header;   /* header.shadow_index == the view's index in the array */
    uint32_t x, y, size;           /* ShadowMap offset / size in the atlas (base.rs:369) */
    uint32_t _pad[5];
} r3n_shadow_view272;
/* multi-GPU hook (not in the reference): called on the calling thread at the points where ranks merge -- after the shadow
 * nodes, after pass 1 (before r3n_hi_z), after pass 2 (before r3n_resolve_opaque); the callee enqueues its collectives on
 * r3n_stream() (r3n_exchange_depth / r3n_exchange_buffers give the buffers), the shadow views' on the stream
 * r3n_exchange_shadow_stream returns.  Non-zero return aborts the frame (R3N_ERR_STATE). */
#define R3N_EXCHANGE_SHADOW 0u
#define R3N_EXCHANGE_PASS1 1u
#define R3N_EXCHANGE_PASS2 2u
typedef int (*r3n_exchange_fn)(void *user, uint32_t site);
#define R3N_FRAME_VIEWPORT_FIRST 1u  /* hand the GPU the viewport's bake + pass 1 before the shadow nodes (same results) */
#define R3N_FRAME_SHADOW_MASK 2u     /* shadow_view_mask is valid: bit v set = this rank renders view v, WHOLE (every object slot,
                                        whatever r3n_set_object_range says); the others are not rendered here (their atlas
                                        rectangles arrive through the exchange) */
typedef struct r3n_frame_desc {
    uint32_t struct_size;            /* sizeof(r3n_frame_desc) */
    uint32_t flags;                  /* R3N_FRAME_* */
    uint32_t width, height, samples; /* as r3n_frame_begin */
    uint32_t shadow_atlas_width, shadow_atlas_height;
    uint32_t n_shadow_views;
    float clear_color[4];
    const r3n_frame_uniforms496 *uniforms;
    const r3n_camera_header240 *viewport_header;
    const r3n_shadow_view272 *shadow_views;  /* n_shadow_views entries */
    uint64_t shadow_view_mask;
    /* light buffers as r3n_lights_write takes them; NULL directional_buffer = keep what the context has */
    const void *directional_buffer;
    uint64_t directional_bytes;
    const void *point_buffer;
    uint64_t point_bytes;
    /* skinning (base.rs:145) as r3n_skinning takes it; n_skeletons == 0 = no skinning node this frame */
    const r3n_skinning_input40 *skin_inputs;
    uint32_t n_skeletons, n_joint_matrices;
    const float *joint_matrices;
    r3n_exchange_fn exchange;        /* NULL = single GPU */
    void *exchange_user;
} r3n_frame_desc;
int r3n_render_frame(r3n_ctx *ctx, const r3n_frame_desc *desc);

/* ---- multi-GPU support: object-range sharding (not in the reference).
 * Only objects with slot in [begin, end) are culled/drawn by this context; buffers stay replicated.  The ranges are the CALLER's
 * partition of the slots that exist when it is made: a slot beyond every rank's `end` (objects added after the object buffer grew) is
 * drawn by NO rank until the ranges are set again -- re-partition after world edits that add slots, or give the last rank
 * end = 0xFFFFFFFF. */
int r3n_set_object_range(r3n_ctx *ctx, uint32_t begin, uint32_t end);
/* The same sharding by OWNER BYTE instead of slot range: this context culls / draws the opaque and cutout objects whose
 * owners[slot] == rank (a spatial partition -- Morton order of the bounding-sphere centres -- gives every rank a compact region
 * of the world and of the screen; its slots are not contiguous).  n >= the object capacity (re-send after the object buffer
 * grows); owners == NULL returns to r3n_set_object_range.  Synchronises (world-edit rate). */
int r3n_set_object_owners(r3n_ctx *ctx, const uint8_t *owners, uint32_t n, uint32_t rank);
/* How the VIEWPORT camera is sharded over ranks (shadow views are sharded by view in both):
 *   R3N_SHARD_OBJECTS (default): a rank draws its objects (r3n_set_object_range / r3n_set_object_owners) over the whole target;
 *     ranks MAX-merge the depth plane after pass 1 and the keys after pass 2;
 *   R3N_SHARD_ROWS (sort-first): a rank culls and draws EVERY object, but rasterises only the rows [row_begin, row_end) of
 *     r3n_set_row_range -- its rows of the depth plane / keys are then final without a reduction: after pass 1 the ranks all-gather
 *     their rows of the depth plane (every rank culls against the whole Hi-Z pyramid and arrives at the unsharded visible sets),
 *     after pass 2 nothing is exchanged.  Culling is replicated, pixel work is divided, and the largest exchange (8 B per pixel of
 *     keys) disappears.  Do not combine with object ranges / owners. */
#define R3N_SHARD_OBJECTS 0u
#define R3N_SHARD_ROWS 1u
int r3n_set_shard_mode(r3n_ctx *ctx, uint32_t mode);
/* The exchanges of the sort-first split issued BY THE LIBRARY over RCCL, inside r3n_render_frame: no callback, no host-language
 * collective calls on the frame path (Python's torch.distributed calls cost 0.33 ms per frame, more than a rank's GPU work at
 * N = 8).  RCCL is bound at run time (the copy the process already holds -- PyTorch's -- else ROCm's librccl.so.1).
 *   r3n_comm_unique_id   one ncclUniqueId (128 B); rank 0 makes R3N_COMM_IDS of them and hands them to every rank through
 *                        whatever launched the ranks (torch.distributed broadcast, MPI, a file);
 *   r3n_comm_init        collective: three communicators -- main stream (depth bands), shadow lane (shadow views), resolve stream
 *                        (image rows) -- so that no exchange queues behind another stream's;  switches the context to
 *                        R3N_SHARD_ROWS; from then on r3n_render_frame owns the shadow views v with v mod world == rank, rasterises
 *                        the band of rows `rank` of r3n_host row ranges (rows split as evenly as possible, the first height mod world
 *                        bands one row taller), broadcasts the shadow rectangles on the shadow lane's stream, all-gathers the
 *                        depth bands in front of r3n_hi_z (keys under MSAA; in place when the bands are equal, else one grouped
 *                        broadcast per band) and the Rgba8 rows behind the resolve.  r3n_frame_desc.exchange must be NULL then.
 *   r3n_comm_set_split   R3N_SHARD_ROWS (what r3n_comm_init selects) or R3N_SHARD_OBJECTS: the object-range split of BASELINE.json's
 *                        north_star -- this rank culls + draws the viewport's objects of r3n_set_object_range / r3n_set_object_owners
 *                        over the WHOLE target; r3n_render_frame then issues a MAX all-reduce of the pass-1 depth plane (f32;
 *                        the u64 keys under MSAA) in front of r3n_hi_z and a MAX reduce-scatter of the u64 visibility keys onto
 *                        the row bands behind pass 2 (an all-reduce when the bands are ragged); shadow views, resolve bands and
 *                        the row gather as above.  Choose by workload: rows replicates the cull, objects moves 12 B per pixel.
 *   r3n_comm_destroy     collective; back to a single-rank context.
 * The three communicators run collectives concurrently on three streams; RCCL requires every rank to issue the collectives of one
 * communicator in the same order, so r3n_frame_desc.flags, samples, the target size and n_shadow_views MUST be identical on
 * every rank (tests/rccl_shim.cpp executes every collective synchronously and times out on an order mismatch). */
#define R3N_COMM_ID_BYTES 128
#define R3N_COMM_IDS 3
int r3n_comm_unique_id(uint8_t *id /* R3N_COMM_ID_BYTES */);
int r3n_comm_init(r3n_ctx *ctx, const uint8_t *ids /* R3N_COMM_IDS x R3N_COMM_ID_BYTES */, uint32_t rank, uint32_t world);
int r3n_comm_set_split(r3n_ctx *ctx, uint32_t mode /* R3N_SHARD_OBJECTS | R3N_SHARD_ROWS */);
int r3n_comm_destroy(r3n_ctx *ctx);
/* Device pointers + sizes of the exchange buffers, for RCCL (all-reduce MAX over ranks):
 * the 64-bit visibility/depth keys (width*height u64, as int64 non-negative) and the f32 shadow atlas. */
/* this camera's own object-slot range, overriding r3n_set_object_range for it (a shadow view owned whole by one rank draws
 * every object there and nothing elsewhere); begin == end == 0xFFFFFFFF returns the camera to the global range */
int r3n_set_camera_object_range(r3n_ctx *ctx, r3n_camera camera, uint32_t begin, uint32_t end);
/* the pass-1 exchange of single-sample targets: only DEPTH has to be global for the Hi-Z cull (33 MB at 4K instead of 66 MB of
 * keys).  Derives mip 0 of the Hi-Z pyramid (f32 depth plane) from the keys and returns its device address; after the callers'
 * element-wise MAX all-reduce of the plane on r3n_stream(), r3n_hi_z builds the pyramid from it. */
int r3n_exchange_depth(r3n_ctx *ctx, void **depth_f32, uint64_t *count);
int r3n_exchange_buffers(r3n_ctx *ctx, void **visibility_keys, uint64_t *visibility_count,
                         void **shadow_atlas, uint64_t *shadow_atlas_count);
/* The shadow atlas for the shadow-view exchange (a view's owner sends its atlas rectangle to the other ranks) WITHOUT ordering the
 * main stream behind the shadow views: *stream = a shadow lane's HIP stream, ordered behind every lane that drew a view this
 * frame (the main stream while R3N_SINGLE_STREAM=1).  Collectives enqueued there run beside the viewport's passes, which never
 * read the atlas; the resolve -- its only reader -- waits for that lane as it does for the views themselves. */
int r3n_exchange_shadow_stream(r3n_ctx *ctx, void **shadow_atlas, uint64_t *shadow_atlas_count, void **stream);
/* Device pointer of the tonemapped Rgba8 image (width*height*4 bytes) and the rows [row_begin,row_end) this
 * rank resolves/tonemaps when screen-space work is split after the exchange. */
int r3n_set_row_range(r3n_ctx *ctx, uint32_t row_begin, uint32_t row_end);
int r3n_output_buffer(r3n_ctx *ctx, void **rgba8, uint64_t *bytes);
/* The same buffer WITHOUT ordering the main stream behind the resolve, for work that only has to follow the resolve (the row
 * all-gather): *stream = the HIP stream this frame's resolve was enqueued on (its own stream while frames are in flight, else
 * the main stream).  Work enqueued there is announced with r3n_output_work_enqueued, so that whatever later waits for the resolve
 * (read-backs, the frame that reuses these targets) waits for that work too.  The next frame's culling and rasterisation then
 * overlap with the resolve AND the gather, as they do without an exchange. */
int r3n_output_buffer_async(r3n_ctx *ctx, void **rgba8, uint64_t *bytes, void **stream);
int r3n_output_work_enqueued(r3n_ctx *ctx);

/* ---- parity / debug taps (not in the reference; synchronise) */
/* L1 set: flags[i] = 1 iff object slot i survived the frustum test of `camera`'s last r3n_cull */
int r3n_readback_visible_objects(r3n_ctx *ctx, r3n_camera camera, uint8_t *flags, uint32_t capacity);
/* L2 set in canonical layout: bit for (object o, triangle t) at index tri_base[o] + t, where tri_base is the
 * exclusive scan of (enabled ? index_count/3 : 0) over object slots.  pass = execute_culling result
 * (cull.wgsl:264-324); residual = newly visible (cull.wgsl:366-371).  `n` = total triangle count. */
int r3n_readback_triangle_sets(r3n_ctx *ctx, r3n_camera camera, uint8_t *pass, uint8_t *residual, uint64_t n);
/* per-region IndirectCall as cull.wgsl leaves it: calls[0..3) predicted, calls[3..6) residual (by material key) */
int r3n_readback_draw_calls(r3n_ctx *ctx, r3n_camera camera, r3n_indirect_call calls[6]);
/* work-queue occupancy of the rasteriser: big_items[i] = number of >8x8 px work items the i-th r3n_forward call of
 * the last frame produced (performance diagnostics only) */
int r3n_readback_raster_stats(r3n_ctx *ctx, uint32_t big_items[64]);
/* The camera's baked matrices, 32 floats per slot.  After r3n_render_frame only the slots inside the frustum this frame or last
 * frame are defined (see r3n_uniform_bake); after the per-node r3n_uniform_bake every enabled slot is. */
int r3n_readback_baked(r3n_ctx *ctx, r3n_camera camera, float *model_view_and_mvp, uint32_t capacity);
int r3n_readback_mesh(r3n_ctx *ctx, uint64_t byte_offset, void *dst, uint64_t bytes); /* e.g. skinned attribute runs */
int r3n_readback_joint_matrices(r3n_ctx *ctx, uint32_t first_matrix, float *dst, uint32_t n_matrices); /* what the last r3n_skinning read */
int r3n_readback_texels(r3n_ctx *ctx, uint64_t first_texel, uint32_t *rgba8, uint64_t n_texels); /* the decoded RGBA8 texel pool (after
                                                                                         r3n_textures_write_encoded: texture i's levels back to back, textures
                                                                                         in array order, each starting on a 4-texel boundary) */
int r3n_readback_visibility(r3n_ctx *ctx, uint64_t *keys);  /* width*height*samples, a pixel's samples contiguous */
int r3n_readback_depth(r3n_ctx *ctx, float *depth);         /* width*height, from the visibility keys (min over samples) */
int r3n_readback_hiz(r3n_ctx *ctx, float *pyramid, uint64_t count); /* all mips, mip0 first */
int r3n_readback_shadow_atlas(r3n_ctx *ctx, float *atlas);  /* atlas_w*atlas_h */
int r3n_readback_hdr(r3n_ctx *ctx, uint16_t *rgba16f);      /* width*height*4 half bits */
int r3n_readback_output(r3n_ctx *ctx, uint8_t *rgba8, float *rgba_f32); /* either may be NULL */

/* ---- timing taps for bench.py: HIP events recorded on the context's stream around every kernel launch of a
 * named stage.  r3n_stage_times returns accumulated milliseconds and launch counts since the last reset. */
#define R3N_STAGE_BAKE 0
#define R3N_STAGE_OBJECT_CULL 1
#define R3N_STAGE_TRIANGLE_CULL 2
#define R3N_STAGE_HIZ 3
#define R3N_STAGE_RASTER 4          /* viewport: per-triangle pass (small triangles + work-item emission) */
#define R3N_STAGE_SHADE 5
#define R3N_STAGE_TONEMAP 6
#define R3N_STAGE_CLEAR 7
#define R3N_STAGE_RASTER_BIG 8      /* viewport: wave-cooperative pass over the large-triangle work items */
#define R3N_STAGE_SHADOW_RASTER 9   /* shadow views: per-triangle pass */
#define R3N_STAGE_SHADOW_RASTER_BIG 10
#define R3N_STAGE_SKINNING 11
#define R3N_STAGE_VERTEX 12         /* resolve pre-pass: flag the visible triangles + one vertex stage per flagged triangle */
#define R3N_STAGE_POSE 13           /* animation poses evaluated in front of the skinning kernel */
#define R3N_STAGE_EXCHANGE_SHADOW 14 /* r3n_comm_init: shadow rectangles packed, broadcast, unpacked (shadow lane) */
#define R3N_STAGE_EXCHANGE_DEPTH 15  /* depth bands (keys under MSAA) gathered in front of Hi-Z (main stream) */
#define R3N_STAGE_EXCHANGE_ROWS 16   /* Rgba8 rows gathered behind the resolve (resolve's stream) */
#define R3N_STAGE_EXCHANGE_KEYS 17   /* object-range split: MAX reduce-scatter of the visibility keys onto the row bands (main stream) */
#define R3N_STAGE_RASTER_CUT 18      /* viewport, CUTOUT key: per-triangle pass (alpha test per fragment; the opaque key's launches stay under RASTER) */
#define R3N_STAGE_RASTER_BIG_CUT 19  /* viewport, CUTOUT key: work-item pass */
#define R3N_STAGE_COUNT 20
int r3n_timing_enable(r3n_ctx *ctx, int enable);
/* What a timed span holds besides its kernels -- two event packets and a launch's dispatch, measured around an empty kernel when
 * timing is first enabled (median of 32) -- and already taken off every span r3n_stage_times reports. */
int r3n_timing_overhead(r3n_ctx *ctx, double *ms_per_span);
/* Shadow views normally run on auxiliary streams concurrently with the viewport chain; per-kernel durations measured
 * while kernels of other streams are resident are inflated, so timing passes can serialise everything on the main
 * stream (enable = 0).  Only between frames. */
int r3n_set_multi_stream(r3n_ctx *ctx, int enable);
int r3n_stage_times(r3n_ctx *ctx, double ms[R3N_STAGE_COUNT], uint64_t launches[R3N_STAGE_COUNT], int reset);
/* Measured HBM streaming rate of this device, the denominator BASELINE.md section 5 asks for ("measured on the target
 * box, not taken from the datasheet"): a float4 grid-stride copy of `bytes` (>= 64 MiB, well past the 256 MiB Infinity
 * Cache when 1 GiB) repeated `repeats` times, HIP-event timed; *gb_per_s = (bytes read + bytes written) / best time.
 * Allocates and frees two scratch buffers; synchronises. */
int r3n_hbm_copy_rate(r3n_ctx *ctx, uint64_t bytes, uint32_t repeats, double *gb_per_s);
/* Self-test of the short correctly-rounded reciprocal / square root / reciprocal square root sequences the shading kernels use
 * (csrc/exact_math.h): all 2^32 f32 bit patterns on the device against the compiler's IEEE expansions.  hist[3][512]: patterns
 * on which the UNGUARDED sequences differ, by function (rcp, sqrt, rsqrt) and sign + biased exponent; guarded[3]: patterns on
 * which the guarded functions -- what the library evaluates -- differ (must be 0).  Needs no context; returns a hipError_t. */
int r3n_selftest_exact_math(int hip_device, unsigned long long *hist, unsigned long long *guarded);
/* exact_math::unorm8 (c / 255 without the division) against the division for all 256 inputs; *n_bad = how many differ (0). */
int r3n_selftest_unorm8(int hip_device, uint32_t *n_bad);

/* ---- host-side mirror of the reference's CPU math on the path (rend3_amd/csrc/host.cpp).
 * In a real integration these stay in Rust (rend3 core); they exist here so the standalone harness, the
 * bench and the parity tests can build the boundary inputs without the oracle.  Matrices are column-major f32[16]. */
/* glam Mat4 mul / inverse as used by camera.rs:64-73, uniforms.rs:41-43 */
void r3n_host_mat4_mul(const float *a, const float *b, float *out);
void r3n_host_mat4_inverse(const float *m, float *out);
/* glam look_at_{lh,rh} (shadow_camera.rs:12-14,28); rh != 0 selects the right-handed form */
void r3n_host_look_at(const float eye[3], const float center[3], const float up[3], int rh, float *out);
/* camera.rs:88-107: kind 0 = Orthographic{size[3]}, 1 = Perspective{vfov_deg = params[0], near = params[1]} */
void r3n_host_projection(int kind, const float *params, int rh, float aspect_ratio, float *out);
/* Frustum::from_matrix, rend3/src/util/frustum.rs:96-145: 5 planes x vec4 */
void r3n_host_frustum_from_matrix(const float *m, float *planes20);
/* Frustum::contains_sphere, rend3/src/util/frustum.rs:148-161: 1 when the sphere touches or is inside all 5 planes */
int r3n_host_frustum_contains_sphere(const float *planes20, const float center[3], float radius);
/* BoundingSphere::{from_mesh, apply_transform}, frustum.rs:15-56 */
void r3n_host_bounding_sphere_from_mesh(const float *positions, uint64_t vertex_count, float out_center[3],
                                        float *out_radius);
void r3n_host_bounding_sphere_apply_transform(const float center[3], float radius, const float *m,
                                              float out_center[3], float *out_radius);
/* ObjectManager::object_add_callback for `n` objects at once (rend3/src/managers/object.rs:236-300): fills the 128-byte
 * records (world bounding sphere = mesh sphere transformed, object.rs:268-269; enabled = 1).
 * mesh_desc[i] = {center.xyz, radius} as f32[4]; mesh_u32[i] = {first_index, index_count, attr_off[6]} as u32[8]. */
void r3n_host_build_object_records(uint32_t n, const float *transforms /* 16 n */, const float *mesh_desc /* 4 n */,
                                   const uint32_t *mesh_u32 /* 8 n */, const uint32_t *material_index /* n */,
                                   r3n_object128 *out_records);
/* Mesh::calculate_normals_for_buffers, rend3-types/src/lib.rs:662-704 */
void r3n_host_calculate_normals(const float *positions, uint64_t vertex_count, const uint32_t *indices,
                                uint64_t index_count, int left_handed, float *normals);
/* shadow_camera, rend3/src/managers/directional/shadow_camera.rs:6-33: outputs the shadow view matrix and its
 * orthographic projection */
void r3n_host_shadow_camera(const float direction[3], float distance, uint32_t resolution,
                            const float camera_location[3], int rh, float *out_view, float *out_proj);
/* allocate_shadow_atlas, rend3/src/managers/directional/shadow_alloc.rs:59-136.  Returns the number of maps
 * written (0 when there is nothing to allocate); out_maps[i] = {offset_x, offset_y, size, handle}. */
uint32_t r3n_host_allocate_shadow_atlas(const uint32_t *handles, const uint16_t *resolutions, uint32_t n,
                                        uint32_t max_dimension, uint32_t out_dimensions[2],
                                        uint32_t *out_maps /* 4*n */);

/* Everything the CPU computes per frame on the path, in one call: CameraState::new (camera.rs:23-85) for the viewport,
 * DirectionalLightManager::evaluate (directional.rs:99-157: shadow atlas allocation, one shadow camera per light, the light
 * buffer), FrameUniforms::new (uniforms.rs:28-48) and the PerCameraUniform headers of the viewport and of every shadow view
 * (culler.rs:477-502).  `lights[i].resolution == 0` marks a free slot of the light manager (skipped; handles keep their index). */
typedef struct r3n_host_camera144 {
    float view[16];
    uint32_t projection_kind;      /* 0 Orthographic{size}, 1 Perspective{vfov, near}, 2 Raw(mat4) (camera.rs:88-107) */
    uint32_t handedness;           /* 0 left, 1 right */
    float aspect_ratio;            /* Option<f32>: <= 0 = None (1.0) */
    uint32_t _pad;
    float projection_params[16];   /* size.xyz | vfov degrees, near | the matrix */
} r3n_host_camera144;
typedef struct r3n_host_directional_light48 {
    float color[3];
    float intensity;
    float direction[3];
    float distance;
    uint32_t resolution;
    uint32_t _pad[3];
} r3n_host_directional_light48;
typedef struct r3n_host_frame {
    r3n_frame_uniforms496 uniforms;
    r3n_camera_header240 viewport_header;
    uint32_t shadow_atlas_width, shadow_atlas_height, n_shadow_views, _pad0;
    float camera_location[3];
    float _pad1;
    float view_proj[16];
    r3n_shadow_view272 shadow_views[R3N_MAX_SHADOW_VIEWS];
    uint32_t shadow_handles[R3N_MAX_SHADOW_VIEWS];  /* light handle of view i */
    uint64_t directional_bytes;                     /* 16 + 128 * n_shadow_views */
    uint8_t directional_buffer[8208];              /* 16 + 128 * R3N_MAX_SHADOW_VIEWS */
} r3n_host_frame;
/* returns 0, or -1 when more than R3N_MAX_SHADOW_VIEWS lights cast shadows */
int r3n_host_evaluate_frame(const r3n_host_camera144 *camera, const r3n_host_directional_light48 *lights, uint32_t n_lights,
                            uint32_t max_atlas_dimension, const float ambient[4], uint32_t width, uint32_t height, uint32_t samples,
                            uint32_t object_capacity, r3n_host_frame *out);

#ifdef __cplusplus
}
#endif
#endif /* R3N_H */
