// rend3-amd-sys: raw bindings of librend3_amd.so -- GENERATED from include/r3n.h by tools/gen_rust_sys.py, do not edit.
// One item per #define, struct and function of the header; the documentation lives there (each entry point cites the
// rend3 interface it replaces).  Source only in this repository: the build image has no Rust toolchain, so this crate is
// kept honest by tests/test_rust_bindings.py (symbol-for-symbol diff against the header and the built library).
#![allow(non_camel_case_types, non_upper_case_globals, clippy::too_many_arguments)]
use std::os::raw::{c_char, c_int, c_void};

pub const R3N_OK: i32 = 0;
pub const R3N_ERR_INVALID_ARG: i32 = -1;
pub const R3N_ERR_HIP: i32 = -2;
pub const R3N_ERR_NO_DEVICE: i32 = -3;
pub const R3N_ERR_STATE: i32 = -4;
pub const R3N_ERR_UNSUPPORTED: i32 = -5;
pub const R3N_ERR_CAPACITY: i32 = -6;
pub const R3N_CAMERA_VIEWPORT: u32 = 0xFFFFFFFF;
pub const R3N_MAX_SHADOW_VIEWS: u32 = 64;
pub const R3N_PASS_DEPTH: u32 = 0;
pub const R3N_PASS_FORWARD: u32 = 1;
pub const R3N_SOURCE_PREDICTED: u32 = 0;
pub const R3N_SOURCE_RESIDUAL: u32 = 1;
pub const R3N_KEY_OPAQUE: u32 = 0;
pub const R3N_KEY_CUTOUT: u32 = 1;
pub const R3N_KEY_BLEND: u32 = 2;
pub const R3N_TEXTURE_RGBA8_UNORM: u32 = 0;
pub const R3N_TEXTURE_RGBA8_UNORM_SRGB: u32 = 1;
pub const R3N_TEXTURE_R8_UNORM: u32 = 2;
pub const R3N_TEXTURE_RG8_UNORM: u32 = 3;
pub const R3N_TEXTURE_BGRA8_UNORM: u32 = 4;
pub const R3N_TEXTURE_BGRA8_UNORM_SRGB: u32 = 5;
pub const R3N_TEXTURE_BC1_RGBA_UNORM: u32 = 6;
pub const R3N_TEXTURE_BC1_RGBA_UNORM_SRGB: u32 = 7;
pub const R3N_TEXTURE_BC2_RGBA_UNORM: u32 = 8;
pub const R3N_TEXTURE_BC2_RGBA_UNORM_SRGB: u32 = 9;
pub const R3N_TEXTURE_BC3_RGBA_UNORM: u32 = 10;
pub const R3N_TEXTURE_BC3_RGBA_UNORM_SRGB: u32 = 11;
pub const R3N_TEXTURE_BC4_R_UNORM: u32 = 12;
pub const R3N_TEXTURE_BC5_RG_UNORM: u32 = 13;
pub const R3N_TEXTURE_BC7_RGBA_UNORM: u32 = 14;
pub const R3N_TEXTURE_BC7_RGBA_UNORM_SRGB: u32 = 15;
pub const R3N_TEXTURE_R8_SNORM: u32 = 16;
pub const R3N_TEXTURE_RG8_SNORM: u32 = 17;
pub const R3N_TEXTURE_RGBA8_SNORM: u32 = 18;
pub const R3N_TEXTURE_R16_FLOAT: u32 = 19;
pub const R3N_TEXTURE_RG16_FLOAT: u32 = 20;
pub const R3N_TEXTURE_RGBA16_FLOAT: u32 = 21;
pub const R3N_TEXTURE_R32_FLOAT: u32 = 22;
pub const R3N_TEXTURE_RG32_FLOAT: u32 = 23;
pub const R3N_TEXTURE_RGBA32_FLOAT: u32 = 24;
pub const R3N_TEXTURE_RGBA16_UNORM: u32 = 25;
pub const R3N_TEXTURE_RGBA16_SNORM: u32 = 26;
pub const R3N_TEXTURE_RGB10A2_UNORM: u32 = 27;
pub const R3N_TEXTURE_RG11B10_FLOAT: u32 = 28;
pub const R3N_TEXTURE_RGB9E5_UFLOAT: u32 = 29;
pub const R3N_TEXTURE_BC4_R_SNORM: u32 = 30;
pub const R3N_TEXTURE_BC5_RG_SNORM: u32 = 31;
pub const R3N_TEXTURE_BC6H_RGB_UFLOAT: u32 = 32;
pub const R3N_TEXTURE_BC6H_RGB_FLOAT: u32 = 33;
pub const R3N_TEXTURE_FORMAT_COUNT: u32 = 34;
pub const R3N_SHADE_EXACT: u32 = 0;
pub const R3N_SHADE_FAST: u32 = 1;
pub const R3N_OUTPUT_RGBA8_UNORM_SRGB: u32 = 0;
pub const R3N_OUTPUT_BGRA8_UNORM_SRGB: u32 = 1;
pub const R3N_OUTPUT_RGBA8_UNORM: u32 = 2;
pub const R3N_OUTPUT_BGRA8_UNORM: u32 = 3;
pub const R3N_SKIN_EXACT: u32 = 0;
pub const R3N_SKIN_MFMA: u32 = 1;
pub const R3N_EXCHANGE_SHADOW: u32 = 0;
pub const R3N_EXCHANGE_PASS1: u32 = 1;
pub const R3N_EXCHANGE_PASS2: u32 = 2;
pub const R3N_FRAME_VIEWPORT_FIRST: u32 = 1;
pub const R3N_FRAME_SHADOW_MASK: u32 = 2;
pub const R3N_SHARD_OBJECTS: u32 = 0;
pub const R3N_SHARD_ROWS: u32 = 1;
pub const R3N_COMM_ID_BYTES: i32 = 128;
pub const R3N_COMM_IDS: i32 = 3;
pub const R3N_STAGE_BAKE: i32 = 0;
pub const R3N_STAGE_OBJECT_CULL: i32 = 1;
pub const R3N_STAGE_TRIANGLE_CULL: i32 = 2;
pub const R3N_STAGE_HIZ: i32 = 3;
pub const R3N_STAGE_RASTER: i32 = 4;
pub const R3N_STAGE_SHADE: i32 = 5;
pub const R3N_STAGE_TONEMAP: i32 = 6;
pub const R3N_STAGE_CLEAR: i32 = 7;
pub const R3N_STAGE_RASTER_BIG: i32 = 8;
pub const R3N_STAGE_SHADOW_RASTER: i32 = 9;
pub const R3N_STAGE_SHADOW_RASTER_BIG: i32 = 10;
pub const R3N_STAGE_SKINNING: i32 = 11;
pub const R3N_STAGE_VERTEX: i32 = 12;
pub const R3N_STAGE_POSE: i32 = 13;
pub const R3N_STAGE_EXCHANGE_SHADOW: i32 = 14;
pub const R3N_STAGE_EXCHANGE_DEPTH: i32 = 15;
pub const R3N_STAGE_EXCHANGE_ROWS: i32 = 16;
pub const R3N_STAGE_EXCHANGE_KEYS: i32 = 17;
pub const R3N_STAGE_RASTER_CUT: i32 = 18;
pub const R3N_STAGE_RASTER_BIG_CUT: i32 = 19;
pub const R3N_STAGE_COUNT: i32 = 20;

#[repr(C)]
pub struct r3n_ctx {
    _private: [u8; 0],
}

pub type r3n_exchange_fn = Option<unsafe extern "C" fn(user: *mut c_void, site: u32) -> c_int>;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_object128 {
    pub transform: [f32; 16],
    pub bounding_sphere_center: [f32; 3],
    pub bounding_sphere_radius: f32,
    pub first_index: u32,
    pub index_count: u32,
    pub material_index: u32,
    pub vertex_attribute_start_offsets: [u32; 6],
    pub enabled: u32,
    pub _pad: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_material208 {
    pub textures: [u32; 10],
    pub _pad: [u32; 2],
    pub uv_transform0: [f32; 12],
    pub uv_transform1: [f32; 12],
    pub albedo: [f32; 4],
    pub emissive: [f32; 3],
    pub roughness: f32,
    pub metallic: f32,
    pub reflectance: f32,
    pub clear_coat: f32,
    pub clear_coat_roughness: f32,
    pub anisotropy: f32,
    pub ambient_occlusion: f32,
    pub alpha_cutout: f32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_texture_desc32 {
    pub offset: u32,
    pub width: u32,
    pub height: u32,
    pub mips: u32,
    pub format: u32,
    pub stored_mips: u32,
    pub _pad: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_camera_header240 {
    pub view: [f32; 16],
    pub view_proj: [f32; 16],
    pub shadow_index: u32,
    pub _pad: [u32; 3],
    pub frustum: [f32; 20],
    pub resolution: [f32; 2],
    pub flags: u32,
    pub object_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_frame_uniforms496 {
    pub view: [f32; 16],
    pub view_proj: [f32; 16],
    pub origin_view_proj: [f32; 16],
    pub inv_view: [f32; 16],
    pub inv_view_proj: [f32; 16],
    pub inv_origin_view_proj: [f32; 16],
    pub frustum: [f32; 20],
    pub ambient: [f32; 4],
    pub resolution: [u32; 2],
    pub _pad: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_indirect_call {
    pub vertex_count: u32,
    pub instance_count: u32,
    pub base_index: u32,
    pub vertex_offset: i32,
    pub base_instance: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_skinning_input40 {
    pub base_position_offset: u32,
    pub base_normal_offset: u32,
    pub base_tangent_offset: u32,
    pub joint_indices_offset: u32,
    pub joint_weight_offset: u32,
    pub updated_position_offset: u32,
    pub updated_normal_offset: u32,
    pub updated_tangent_offset: u32,
    pub joint_matrix_base_offset: u32,
    pub vertex_count: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_anim_rig16 {
    pub first_joint: u32,
    pub n_joints: u32,
    pub max_depth: u32,
    pub _pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_anim_joint80 {
    pub parent: i32,
    pub depth: u32,
    pub _pad: [u32; 2],
    pub inverse_bind: [f32; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_anim_clip16 {
    pub rig: u32,
    pub first_track: u32,
    pub duration: f32,
    pub _pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_anim_track80 {
    pub animated: u32,
    pub key_first: [u32; 3],
    pub key_count: [u32; 3],
    pub value_first: [u32; 3],
    pub bind_t: [f32; 3],
    pub bind_r: [f32; 4],
    pub bind_s: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_pose_request16 {
    pub clip: u32,
    pub time: f32,
    pub matrix_base: u32,
    pub _pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_config {
    pub struct_size: u32,
    pub max_big_items: u32,
    pub shade_mode: u32,
    pub _pad: u32,
    pub reserved: [u64; 2],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_shadow_view272 {
    pub header: r3n_camera_header240,
    pub x: u32,
    pub y: u32,
    pub size: u32,
    pub _pad: [u32; 5],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_frame_desc {
    pub struct_size: u32,
    pub flags: u32,
    pub width: u32,
    pub height: u32,
    pub samples: u32,
    pub shadow_atlas_width: u32,
    pub shadow_atlas_height: u32,
    pub n_shadow_views: u32,
    pub clear_color: [f32; 4],
    pub uniforms: *const r3n_frame_uniforms496,
    pub viewport_header: *const r3n_camera_header240,
    pub shadow_views: *const r3n_shadow_view272,
    pub shadow_view_mask: u64,
    pub directional_buffer: *const c_void,
    pub directional_bytes: u64,
    pub point_buffer: *const c_void,
    pub point_bytes: u64,
    pub skin_inputs: *const r3n_skinning_input40,
    pub n_skeletons: u32,
    pub n_joint_matrices: u32,
    pub joint_matrices: *const f32,
    pub exchange: r3n_exchange_fn,
    pub exchange_user: *mut c_void,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_host_camera144 {
    pub view: [f32; 16],
    pub projection_kind: u32,
    pub handedness: u32,
    pub aspect_ratio: f32,
    pub _pad: u32,
    pub projection_params: [f32; 16],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_host_directional_light48 {
    pub color: [f32; 3],
    pub intensity: f32,
    pub direction: [f32; 3],
    pub distance: f32,
    pub resolution: u32,
    pub _pad: [u32; 3],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct r3n_host_frame {
    pub uniforms: r3n_frame_uniforms496,
    pub viewport_header: r3n_camera_header240,
    pub shadow_atlas_width: u32,
    pub shadow_atlas_height: u32,
    pub n_shadow_views: u32,
    pub _pad0: u32,
    pub camera_location: [f32; 3],
    pub _pad1: f32,
    pub view_proj: [f32; 16],
    pub shadow_views: [r3n_shadow_view272; R3N_MAX_SHADOW_VIEWS as usize],
    pub shadow_handles: [u32; R3N_MAX_SHADOW_VIEWS as usize],
    pub directional_bytes: u64,
    pub directional_buffer: [u8; 8208],
}

#[link(name = "rend3_amd")]
extern "C" {
    pub fn r3n_create(hip_device: c_int, config: *const r3n_config) -> *mut r3n_ctx;
    pub fn r3n_destroy(ctx: *mut r3n_ctx);
    pub fn r3n_last_error(ctx: *const r3n_ctx) -> *const c_char;
    pub fn r3n_create_error() -> *const c_char;
    pub fn r3n_sync(ctx: *mut r3n_ctx) -> c_int;
    pub fn r3n_stream(ctx: *mut r3n_ctx) -> *mut c_void;
    pub fn r3n_mesh_buffer_write(ctx: *mut r3n_ctx, byte_offset: u64, data: *const c_void, bytes: u64) -> c_int;
    pub fn r3n_objects_write(ctx: *mut r3n_ctx, slots: *const u32, records: *const r3n_object128, n: u32, capacity: u32) -> c_int;
    pub fn r3n_materials_write(ctx: *mut r3n_ctx, slots: *const u32, records: *const r3n_material208, keys: *const u8, n: u32) -> c_int;
    pub fn r3n_textures_write(ctx: *mut r3n_ctx, descs: *const r3n_texture_desc32, n_textures: u32, texels: *const u32, n_texels: u64) -> c_int;
    pub fn r3n_textures_write_encoded(ctx: *mut r3n_ctx, descs: *const r3n_texture_desc32, n_textures: u32, payload: *const c_void, payload_bytes: u64) -> c_int;
    pub fn r3n_blend_order_write(ctx: *mut r3n_ctx, objects_back_to_front: *const u32, n: u32) -> c_int;
    pub fn r3n_lights_write(ctx: *mut r3n_ctx, directional_buffer: *const c_void, directional_bytes: u64, point_buffer: *const c_void, point_bytes: u64) -> c_int;
    pub fn r3n_set_output_format(ctx: *mut r3n_ctx, format: u32) -> c_int;
    pub fn r3n_set_skinning_mode(ctx: *mut r3n_ctx, mode: u32) -> c_int;
    pub fn r3n_set_shade_mode(ctx: *mut r3n_ctx, mode: u32) -> c_int;
    pub fn r3n_frame_begin(ctx: *mut r3n_ctx, uniforms: *const r3n_frame_uniforms496, width: u32, height: u32, samples: u32, clear_color: *const f32, shadow_atlas_width: u32, shadow_atlas_height: u32) -> c_int;
    pub fn r3n_skinning(ctx: *mut r3n_ctx, inputs: *const r3n_skinning_input40, n_skeletons: u32, joint_matrices: *const f32, n_joint_matrices: u32) -> c_int;
    pub fn r3n_animation_write(ctx: *mut r3n_ctx, rigs: *const r3n_anim_rig16, n_rigs: u32, joints: *const r3n_anim_joint80, n_joints: u32, clips: *const r3n_anim_clip16, n_clips: u32, tracks: *const r3n_anim_track80, n_tracks: u32, times: *const f32, n_times: u32, values: *const f32, n_values: u32) -> c_int;
    pub fn r3n_pose_skeletons(ctx: *mut r3n_ctx, requests: *const r3n_pose_request16, n: u32) -> c_int;
    pub fn r3n_uniform_bake(ctx: *mut r3n_ctx, camera: u32, header: *const r3n_camera_header240) -> c_int;
    pub fn r3n_cull(ctx: *mut r3n_ctx, camera: u32) -> c_int;
    pub fn r3n_hi_z(ctx: *mut r3n_ctx) -> c_int;
    pub fn r3n_shadow_viewport(ctx: *mut r3n_ctx, shadow_camera: u32, x: u32, y: u32, size: u32) -> c_int;
    pub fn r3n_forward(ctx: *mut r3n_ctx, camera: u32, pass: u32, source: u32, material_key: u32) -> c_int;
    pub fn r3n_resolve_opaque(ctx: *mut r3n_ctx) -> c_int;
    pub fn r3n_tonemap(ctx: *mut r3n_ctx, host_rgba8: *mut c_void, pitch_bytes: u64) -> c_int;
    pub fn r3n_hdr_write(ctx: *mut r3n_ctx, rgba16f: *const u16, first_pixel: u64, n_pixels: u64) -> c_int;
    pub fn r3n_frame_end(ctx: *mut r3n_ctx) -> c_int;
    pub fn r3n_render_frame(ctx: *mut r3n_ctx, desc: *const r3n_frame_desc) -> c_int;
    pub fn r3n_set_object_range(ctx: *mut r3n_ctx, begin: u32, end: u32) -> c_int;
    pub fn r3n_set_object_owners(ctx: *mut r3n_ctx, owners: *const u8, n: u32, rank: u32) -> c_int;
    pub fn r3n_set_shard_mode(ctx: *mut r3n_ctx, mode: u32) -> c_int;
    pub fn r3n_comm_unique_id(id: *mut u8) -> c_int;
    pub fn r3n_comm_init(ctx: *mut r3n_ctx, ids: *const u8, rank: u32, world: u32) -> c_int;
    pub fn r3n_comm_set_split(ctx: *mut r3n_ctx, mode: u32) -> c_int;
    pub fn r3n_comm_destroy(ctx: *mut r3n_ctx) -> c_int;
    pub fn r3n_set_camera_object_range(ctx: *mut r3n_ctx, camera: u32, begin: u32, end: u32) -> c_int;
    pub fn r3n_exchange_depth(ctx: *mut r3n_ctx, depth_f32: *mut *mut c_void, count: *mut u64) -> c_int;
    pub fn r3n_exchange_buffers(ctx: *mut r3n_ctx, visibility_keys: *mut *mut c_void, visibility_count: *mut u64, shadow_atlas: *mut *mut c_void, shadow_atlas_count: *mut u64) -> c_int;
    pub fn r3n_exchange_shadow_stream(ctx: *mut r3n_ctx, shadow_atlas: *mut *mut c_void, shadow_atlas_count: *mut u64, stream: *mut *mut c_void) -> c_int;
    pub fn r3n_set_row_range(ctx: *mut r3n_ctx, row_begin: u32, row_end: u32) -> c_int;
    pub fn r3n_output_buffer(ctx: *mut r3n_ctx, rgba8: *mut *mut c_void, bytes: *mut u64) -> c_int;
    pub fn r3n_output_buffer_async(ctx: *mut r3n_ctx, rgba8: *mut *mut c_void, bytes: *mut u64, stream: *mut *mut c_void) -> c_int;
    pub fn r3n_output_work_enqueued(ctx: *mut r3n_ctx) -> c_int;
    pub fn r3n_readback_visible_objects(ctx: *mut r3n_ctx, camera: u32, flags: *mut u8, capacity: u32) -> c_int;
    pub fn r3n_readback_triangle_sets(ctx: *mut r3n_ctx, camera: u32, pass: *mut u8, residual: *mut u8, n: u64) -> c_int;
    pub fn r3n_readback_draw_calls(ctx: *mut r3n_ctx, camera: u32, calls: *mut r3n_indirect_call) -> c_int;
    pub fn r3n_readback_raster_stats(ctx: *mut r3n_ctx, big_items: *mut u32) -> c_int;
    pub fn r3n_readback_baked(ctx: *mut r3n_ctx, camera: u32, model_view_and_mvp: *mut f32, capacity: u32) -> c_int;
    pub fn r3n_readback_mesh(ctx: *mut r3n_ctx, byte_offset: u64, dst: *mut c_void, bytes: u64) -> c_int;
    pub fn r3n_readback_joint_matrices(ctx: *mut r3n_ctx, first_matrix: u32, dst: *mut f32, n_matrices: u32) -> c_int;
    pub fn r3n_readback_texels(ctx: *mut r3n_ctx, first_texel: u64, rgba8: *mut u32, n_texels: u64) -> c_int;
    pub fn r3n_readback_visibility(ctx: *mut r3n_ctx, keys: *mut u64) -> c_int;
    pub fn r3n_readback_depth(ctx: *mut r3n_ctx, depth: *mut f32) -> c_int;
    pub fn r3n_readback_hiz(ctx: *mut r3n_ctx, pyramid: *mut f32, count: u64) -> c_int;
    pub fn r3n_readback_shadow_atlas(ctx: *mut r3n_ctx, atlas: *mut f32) -> c_int;
    pub fn r3n_readback_hdr(ctx: *mut r3n_ctx, rgba16f: *mut u16) -> c_int;
    pub fn r3n_readback_output(ctx: *mut r3n_ctx, rgba8: *mut u8, rgba_f32: *mut f32) -> c_int;
    pub fn r3n_timing_enable(ctx: *mut r3n_ctx, enable: c_int) -> c_int;
    pub fn r3n_timing_overhead(ctx: *mut r3n_ctx, ms_per_span: *mut f64) -> c_int;
    pub fn r3n_set_multi_stream(ctx: *mut r3n_ctx, enable: c_int) -> c_int;
    pub fn r3n_stage_times(ctx: *mut r3n_ctx, ms: *mut f64, launches: *mut u64, reset: c_int) -> c_int;
    pub fn r3n_hbm_copy_rate(ctx: *mut r3n_ctx, bytes: u64, repeats: u32, gb_per_s: *mut f64) -> c_int;
    pub fn r3n_selftest_exact_math(hip_device: c_int, hist: *mut u64, guarded: *mut u64) -> c_int;
    pub fn r3n_selftest_unorm8(hip_device: c_int, n_bad: *mut u32) -> c_int;
    pub fn r3n_host_mat4_mul(a: *const f32, b: *const f32, out: *mut f32);
    pub fn r3n_host_mat4_inverse(m: *const f32, out: *mut f32);
    pub fn r3n_host_look_at(eye: *const f32, center: *const f32, up: *const f32, rh: c_int, out: *mut f32);
    pub fn r3n_host_projection(kind: c_int, params: *const f32, rh: c_int, aspect_ratio: f32, out: *mut f32);
    pub fn r3n_host_frustum_from_matrix(m: *const f32, planes20: *mut f32);
    pub fn r3n_host_frustum_contains_sphere(planes20: *const f32, center: *const f32, radius: f32) -> c_int;
    pub fn r3n_host_bounding_sphere_from_mesh(positions: *const f32, vertex_count: u64, out_center: *mut f32, out_radius: *mut f32);
    pub fn r3n_host_bounding_sphere_apply_transform(center: *const f32, radius: f32, m: *const f32, out_center: *mut f32, out_radius: *mut f32);
    pub fn r3n_host_build_object_records(n: u32, transforms: *const f32, mesh_desc: *const f32, mesh_u32: *const u32, material_index: *const u32, out_records: *mut r3n_object128);
    pub fn r3n_host_calculate_normals(positions: *const f32, vertex_count: u64, indices: *const u32, index_count: u64, left_handed: c_int, normals: *mut f32);
    pub fn r3n_host_shadow_camera(direction: *const f32, distance: f32, resolution: u32, camera_location: *const f32, rh: c_int, out_view: *mut f32, out_proj: *mut f32);
    pub fn r3n_host_allocate_shadow_atlas(handles: *const u32, resolutions: *const u16, n: u32, max_dimension: u32, out_dimensions: *mut u32, out_maps: *mut u32) -> u32;
    pub fn r3n_host_evaluate_frame(camera: *const r3n_host_camera144, lights: *const r3n_host_directional_light48, n_lights: u32, max_atlas_dimension: u32, ambient: *const f32, width: u32, height: u32, samples: u32, object_capacity: u32, out: *mut r3n_host_frame) -> c_int;
}
