//! rend3 <-> `librend3_amd.so` (include/r3n.h): the context and the world's mirror.
//!
//! Installed as `rend3/src/util/amd.rs` by `bindings/rend3-hooks.patch`.  `Renderer::new` creates ONE `AmdContext` (one `r3n_ctx`: a
//! HIP device and its streams) beside the wgpu device (`rend3/src/renderer/setup.rs`), and the managers hand it every byte
//! they upload, at the sites named on each method below (the patch's hunks): mesh attribute runs and indices, `ShaderObject<M>`
//! and material records as encase lays them out, the 2D textures in the formats the loader produced, the two light buffers.
//! rend3-routine-amd's node bodies (`BaseRenderGraph`, `GpuCuller`, `ForwardRoutine`, ...) render from that copy through the same
//! context (`AmdContext::of(renderer)` = `renderer.amd`).  The wgpu buffers keep being written as before: every other routine of
//! the render graph (skybox, egui, user nodes) still finds its data.
//!
//! Source only in this repository (no Rust toolchain in the build image); the same uploads run from Python through ctypes
//! (`rend3_amd/renderer.py`), which the GPU tests drive.
use std::{ffi::CStr, os::raw::c_void, sync::Arc};

use encase::{internal::WriteInto, ShaderSize, StorageBuffer};
use parking_lot::Mutex;
use rend3_amd_sys as sys;
use rend3_types::{MipmapCount, MipmapSource, Texture};

use crate::Renderer;

/// Owner of the `r3n_ctx` and the error convention.  Lives in `Renderer::amd` (rend3/src/renderer/mod.rs), destroyed with it.
pub struct AmdContext {
    pub ctx: *mut sys::r3n_ctx,
    /// The 2D textures as the loader produced them: `r3n_textures_write_encoded` replaces the whole bindless array, so the
    /// context keeps the encoded payloads (handle index -> texture) and re-sends them when the set changed.
    textures: Mutex<TextureMirror>,
    /// The two light buffers as last written (directional, point): `r3n_lights_write` takes both at once.
    lights: Mutex<(Vec<u8>, Vec<u8>)>,
}

#[derive(Default)]
struct TextureMirror {
    entries: Vec<Option<MirroredTexture>>,
    dirty: bool,
}

struct MirroredTexture {
    format: u32,
    width: u32,
    height: u32,
    mips: u32,
    stored_mips: u32,
    data: Vec<u8>,
}

// The reference serialises graph execution and instruction evaluation behind the data_core mutex (rend3/src/graph/graph.rs:265,
// rend3/src/renderer/eval.rs); the C ABI asks for the same: one thread at a time.  `MeshManager::add` runs outside that lock
// (rend3/src/renderer/mod.rs:148-150) and takes its own buffer-state mutex; r3n_mesh_buffer_write is called under it.
unsafe impl Send for AmdContext {}
unsafe impl Sync for AmdContext {}

impl AmdContext {
    /// `Renderer::new` (rend3/src/renderer/setup.rs): HIP device `R3N_HIP_DEVICE` (default 0); `R3N_SHADE_FAST=1` opts into the
    /// fast fragment arithmetic (framebuffer within 1e-3 after tonemap instead of bit-identical).
    pub fn from_env() -> Result<Arc<Self>, String> {
        let device = std::env::var("R3N_HIP_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
        let fast = std::env::var("R3N_SHADE_FAST").map_or(false, |v| v == "1");
        Self::new(device, fast).map(Arc::new)
    }

    pub fn new(hip_device: i32, shade_fast: bool) -> Result<Self, String> {
        let config = sys::r3n_config {
            struct_size: std::mem::size_of::<sys::r3n_config>() as u32,
            max_big_items: 0,
            shade_mode: if shade_fast { sys::R3N_SHADE_FAST } else { sys::R3N_SHADE_EXACT },
            _pad: 0,
            reserved: [0; 2],
        };
        let ctx = unsafe { sys::r3n_create(hip_device, &config) };
        if ctx.is_null() {
            return Err(unsafe { CStr::from_ptr(sys::r3n_create_error()) }.to_string_lossy().into_owned());
        }
        Ok(Self { ctx, textures: Mutex::new(TextureMirror::default()), lights: Mutex::new((Vec::new(), Vec::new())) })
    }

    /// The context of `renderer`: the routine constructors keep the reference's signatures (`BaseRenderGraph::new(&renderer,
    /// &spp)`, `PbrRoutine::new(&renderer, ...)`), so no call site hands a context around.
    pub fn of(renderer: &Arc<Renderer>) -> Arc<AmdContext> {
        Arc::clone(&renderer.amd)
    }

    /// Same from a node body (`NodeExecutionContext::renderer` is a plain reference).
    pub fn of_ref(renderer: &Renderer) -> Arc<AmdContext> {
        Arc::clone(&renderer.amd)
    }

    pub fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(sys::r3n_last_error(self.ctx)) }.to_string_lossy().into_owned()
    }

    /// Every entry point returns 0 or a negative code and never unwinds.  The reference's node bodies `unwrap` / `panic!` on the
    /// states these codes describe (culler.rs:439,572), so the hooks do the same, with the library's message.
    #[track_caller]
    pub fn check(&self, code: i32, what: &str) {
        assert!(code == sys::R3N_OK, "{what}: {} ({code})", self.last_error());
    }

    /// `RenderGraph::execute` end (graph.rs:510 `queue.submit`).
    pub fn frame_end(&self) {
        self.check(unsafe { sys::r3n_frame_end(self.ctx) }, "r3n_frame_end");
    }

    /// Blocks until the device is idle (tests, screenshots).
    pub fn sync(&self) {
        self.check(unsafe { sys::r3n_sync(self.ctx) }, "r3n_sync");
    }

    // ---------------------------------------------------------------------------------------------- the world's mirror

    /// `MeshManager::add` (rend3/src/managers/mesh.rs:147-156): one call per attribute run and one for the indices, at the
    /// byte offsets the range allocator handed out -- beside `upload.add(range.start, ..)`.
    pub fn mesh_buffer_write(&self, byte_offset: u64, bytes: &[u8]) {
        self.check(unsafe { sys::r3n_mesh_buffer_write(self.ctx, byte_offset, bytes.as_ptr().cast(), bytes.len() as u64) }, "r3n_mesh_buffer_write");
    }

    /// `FreelistDerivedBuffer::apply` for the object archetypes (rend3/src/managers/object.rs:344-364 over
    /// util/freelist/buffer.rs:56-98): the stale `ShaderObject<M>` records, encase-serialised exactly as the scatter copy
    /// writes them (128 B each), and the buffer's reserved count.
    pub fn objects_write<T: ShaderSize + WriteInto>(&self, slots: &[u32], records: impl Iterator<Item = T>, reserved_count: u32) {
        assert_eq!(T::SHADER_SIZE.get(), 128, "ShaderObject<M> is 128 bytes (object.rs:23-36)");
        let mut bytes = vec![0u8; slots.len() * 128];
        for (chunk, record) in bytes.chunks_exact_mut(128).zip(records) {
            StorageBuffer::new(&mut *chunk).write(&record).unwrap();
        }
        self.check(
            unsafe { sys::r3n_objects_write(self.ctx, slots.as_ptr(), bytes.as_ptr().cast::<sys::r3n_object128>(), slots.len() as u32, reserved_count) },
            "r3n_objects_write",
        );
    }

    /// `apply_buffer_gpu::<M>` (rend3/src/managers/material.rs:315-340): the stale `GpuPoweredShaderWrapper<M>` records and
    /// `Material::key()` of each.  Only the 208-byte layout of `PbrMaterial` (rend3-routine/src/pbr/material.rs:526-583) is a
    /// material of the AMD routines; archetypes of other sizes are not mirrored (their objects are not drawn by them either).
    pub fn materials_write<T: ShaderSize + WriteInto>(&self, slots: &[u32], records: impl Iterator<Item = T>, keys: &[u8]) {
        if T::SHADER_SIZE.get() != 208 {
            return;
        }
        assert_eq!(slots.len(), keys.len());
        let mut bytes = vec![0u8; slots.len() * 208];
        for (chunk, record) in bytes.chunks_exact_mut(208).zip(records) {
            StorageBuffer::new(&mut *chunk).write(&record).unwrap();
        }
        self.check(
            unsafe { sys::r3n_materials_write(self.ctx, slots.as_ptr(), bytes.as_ptr().cast::<sys::r3n_material208>(), keys.as_ptr(), slots.len() as u32) },
            "r3n_materials_write",
        );
    }

    /// `Renderer::add_texture_2d` (rend3/src/renderer/mod.rs:183-197): the texture as the caller gave it, under the index its
    /// handle will translate to (`TextureManager::translation_fn`: index + 1 is the id material records carry).  Formats a
    /// `texture_2d<f32>` binding cannot hold (`texture_format_id` = None) are reported like the reference reports them.
    pub fn texture_fill(&self, index: usize, texture: &Texture) {
        let Some(format) = texture_format_id(texture.format) else {
            log::error!("rend3-amd: texture format {:?} is not supported by the AMD routines; the texture reads as unbound", texture.format);
            return;
        };
        let full_chain = 32 - texture.size.x.max(texture.size.y).max(1).leading_zeros();
        let mips = match texture.mip_count {
            MipmapCount::Specific(v) => v.get(),
            MipmapCount::Maximum => full_chain,
        };
        let stored_mips = match texture.mip_source {
            MipmapSource::Uploaded => 0,   // every level is in `data`
            MipmapSource::Generated => 1,  // level 0 only: the library runs the blit chain of util/mipmap.rs on the GPU
        };
        let mut mirror = self.textures.lock();
        if mirror.entries.len() <= index {
            mirror.entries.resize_with(index + 1, || None);
        }
        mirror.entries[index] =
            Some(MirroredTexture { format, width: texture.size.x, height: texture.size.y, mips, stored_mips, data: texture.data.clone() });
        mirror.dirty = true;
    }

    /// `TextureManager::remove` for the 2D manager (rend3/src/renderer/eval.rs, `DeleteTexture2D`).
    pub fn texture_remove(&self, index: usize) {
        let mut mirror = self.textures.lock();
        if let Some(entry) = mirror.entries.get_mut(index) {
            *entry = None;
            mirror.dirty = true;
        }
    }

    /// `TextureManager::evaluate` (rend3/src/managers/texture.rs:259-276: the bindless array is rebuilt when it is dirty): the
    /// whole 2D array in one call.  A removed or unsupported entry stays in the array as a 1 x 1 transparent texel so that the
    /// ids of the others do not move.
    pub fn textures_flush(&self) {
        let mut mirror = self.textures.lock();
        if !mirror.dirty {
            return;
        }
        mirror.dirty = false;
        let mut descs = Vec::with_capacity(mirror.entries.len());
        let mut payload: Vec<u8> = Vec::new();
        for entry in &mirror.entries {
            while payload.len() % 16 != 0 {
                payload.push(0);
            }
            let offset = payload.len() as u32;
            match entry {
                Some(t) => {
                    payload.extend_from_slice(&t.data);
                    descs.push(sys::r3n_texture_desc32 { offset, width: t.width, height: t.height, mips: t.mips, format: t.format, stored_mips: t.stored_mips, _pad: [0; 2] });
                }
                None => {
                    payload.extend_from_slice(&[0u8; 4]);
                    descs.push(sys::r3n_texture_desc32 { offset, width: 1, height: 1, mips: 1, format: sys::R3N_TEXTURE_RGBA8_UNORM, stored_mips: 0, _pad: [0; 2] });
                }
            }
        }
        self.check(
            unsafe { sys::r3n_textures_write_encoded(self.ctx, descs.as_ptr(), descs.len() as u32, payload.as_ptr().cast::<c_void>(), payload.len() as u64) },
            "r3n_textures_write_encoded",
        );
    }

    /// `DirectionalLightManager::evaluate` (rend3/src/managers/directional.rs:135-155): the storage buffer as `write_to_buffer`
    /// fills it (encase: 16-byte header with the count, then 128-byte records).  Shadow cameras and the atlas allocation stay in
    /// Rust.  The library takes both light buffers in one call, so the context keeps the other half as it was last written.
    pub fn lights_write_directional<T: encase::ShaderType + WriteInto>(&self, buffer: &T) {
        let mut lights = self.lights.lock();
        lights.0.clear();
        StorageBuffer::new(&mut lights.0).write(buffer).unwrap();
        self.lights_send(&lights);
    }

    /// `PointLightManager::evaluate` (rend3/src/managers/point.rs:58-74): 16-byte header, 32-byte records.
    pub fn lights_write_point<T: encase::ShaderType + WriteInto>(&self, buffer: &T) {
        let mut lights = self.lights.lock();
        lights.1.clear();
        StorageBuffer::new(&mut lights.1).write(buffer).unwrap();
        self.lights_send(&lights);
    }

    fn lights_send(&self, lights: &(Vec<u8>, Vec<u8>)) {
        let ptr = |v: &Vec<u8>| if v.is_empty() { std::ptr::null() } else { v.as_ptr().cast::<c_void>() };
        self.check(
            unsafe { sys::r3n_lights_write(self.ctx, ptr(&lights.0), lights.0.len() as u64, ptr(&lights.1), lights.1.len() as u64) },
            "r3n_lights_write",
        );
    }

    /// `batch_objects` with `Sorting::BLENDING` (rend3-routine/src/culling/batching.rs:146-176): the blend-key objects back to
    /// front, once per frame before the resolve (called by rend3-routine-amd's transparent-pass node).
    pub fn blend_order_write(&self, objects_back_to_front: &[u32]) {
        self.check(unsafe { sys::r3n_blend_order_write(self.ctx, objects_back_to_front.as_ptr(), objects_back_to_front.len() as u32) }, "r3n_blend_order_write");
    }
}

impl Drop for AmdContext {
    fn drop(&mut self) {
        unsafe { sys::r3n_destroy(self.ctx) }
    }
}

/// `Texture::format` -> the library's format id for `r3n_texture_desc32::format` (`include/r3n.h` R3N_TEXTURE_*): every format
/// rend3-gltf's maps produce (rend3-gltf/src/lib.rs:1157-1610) that a `texture_2d<f32>` binding can hold.  `None`: integer, depth,
/// ETC2 / EAC and ASTC formats -- `add_texture_2d` reports them instead of uploading.
pub fn texture_format_id(format: rend3_types::TextureFormat) -> Option<u32> {
    use rend3_types::TextureFormat as F;
    Some(match format {
        F::Rgba8Unorm => sys::R3N_TEXTURE_RGBA8_UNORM,
        F::Rgba8UnormSrgb => sys::R3N_TEXTURE_RGBA8_UNORM_SRGB,
        F::R8Unorm => sys::R3N_TEXTURE_R8_UNORM,
        F::Rg8Unorm => sys::R3N_TEXTURE_RG8_UNORM,
        F::Bgra8Unorm => sys::R3N_TEXTURE_BGRA8_UNORM,
        F::Bgra8UnormSrgb => sys::R3N_TEXTURE_BGRA8_UNORM_SRGB,
        F::Bc1RgbaUnorm => sys::R3N_TEXTURE_BC1_RGBA_UNORM,
        F::Bc1RgbaUnormSrgb => sys::R3N_TEXTURE_BC1_RGBA_UNORM_SRGB,
        F::Bc2RgbaUnorm => sys::R3N_TEXTURE_BC2_RGBA_UNORM,
        F::Bc2RgbaUnormSrgb => sys::R3N_TEXTURE_BC2_RGBA_UNORM_SRGB,
        F::Bc3RgbaUnorm => sys::R3N_TEXTURE_BC3_RGBA_UNORM,
        F::Bc3RgbaUnormSrgb => sys::R3N_TEXTURE_BC3_RGBA_UNORM_SRGB,
        F::Bc4RUnorm => sys::R3N_TEXTURE_BC4_R_UNORM,
        F::Bc5RgUnorm => sys::R3N_TEXTURE_BC5_RG_UNORM,
        F::Bc7RgbaUnorm => sys::R3N_TEXTURE_BC7_RGBA_UNORM,
        F::Bc7RgbaUnormSrgb => sys::R3N_TEXTURE_BC7_RGBA_UNORM_SRGB,
        // decoded to four f32 per texel
        F::R8Snorm => sys::R3N_TEXTURE_R8_SNORM,
        F::Rg8Snorm => sys::R3N_TEXTURE_RG8_SNORM,
        F::Rgba8Snorm => sys::R3N_TEXTURE_RGBA8_SNORM,
        F::R16Float => sys::R3N_TEXTURE_R16_FLOAT,
        F::Rg16Float => sys::R3N_TEXTURE_RG16_FLOAT,
        F::Rgba16Float => sys::R3N_TEXTURE_RGBA16_FLOAT,
        F::R32Float => sys::R3N_TEXTURE_R32_FLOAT,
        F::Rg32Float => sys::R3N_TEXTURE_RG32_FLOAT,
        F::Rgba32Float => sys::R3N_TEXTURE_RGBA32_FLOAT,
        F::Rgba16Unorm => sys::R3N_TEXTURE_RGBA16_UNORM,
        F::Rgba16Snorm => sys::R3N_TEXTURE_RGBA16_SNORM,
        F::Rgb10a2Unorm => sys::R3N_TEXTURE_RGB10A2_UNORM,
        F::Rg11b10Float => sys::R3N_TEXTURE_RG11B10_FLOAT,
        F::Rgb9e5Ufloat => sys::R3N_TEXTURE_RGB9E5_UFLOAT,
        F::Bc4RSnorm => sys::R3N_TEXTURE_BC4_R_SNORM,
        F::Bc5RgSnorm => sys::R3N_TEXTURE_BC5_RG_SNORM,
        F::Bc6hRgbUfloat => sys::R3N_TEXTURE_BC6H_RGB_UFLOAT,
        F::Bc6hRgbFloat => sys::R3N_TEXTURE_BC6H_RGB_FLOAT,
        _ => return None,
    })
}
