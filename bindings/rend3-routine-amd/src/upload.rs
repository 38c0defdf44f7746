//! World uploads: the byte layouts the managers already build (encase std430) go to the library as they are.
//! Replaces the wgpu `ScatterCopy` / `write_buffer` calls of rend3/src/managers/*.rs at the sites named per function.
use crate::amd::AmdContext;
use rend3_amd_sys as sys;
use std::os::raw::c_void;

impl AmdContext {
    /// `MeshManager::add` (rend3/src/managers/mesh.rs:123-184): one call per attribute run and one for the indices, at the
    /// byte offsets the range allocator handed out.
    pub fn mesh_buffer_write(&self, byte_offset: u64, bytes: &[u8]) {
        self.check(unsafe { sys::r3n_mesh_buffer_write(self.ctx, byte_offset, bytes.as_ptr().cast(), bytes.len() as u64) }, "r3n_mesh_buffer_write");
    }

    /// `ObjectManager::evaluate` (rend3/src/managers/object.rs:344-364 over util/freelist/buffer.rs:56-98): the dirty
    /// `ShaderObject<M>` records (128 B each) and the buffer's reserved count.
    pub fn objects_write(&self, slots: &[u32], records: &[sys::r3n_object128], reserved_count: u32) {
        assert_eq!(slots.len(), records.len());
        self.check(unsafe { sys::r3n_objects_write(self.ctx, slots.as_ptr(), records.as_ptr(), slots.len() as u32, reserved_count) }, "r3n_objects_write");
    }

    /// `MaterialManager::evaluate` (rend3/src/managers/material.rs:202-227): 208-B records + `M::key()` per record.
    pub fn materials_write(&self, slots: &[u32], records: &[sys::r3n_material208], keys: &[u8]) {
        assert!(slots.len() == records.len() && slots.len() == keys.len());
        self.check(unsafe { sys::r3n_materials_write(self.ctx, slots.as_ptr(), records.as_ptr(), keys.as_ptr(), slots.len() as u32) }, "r3n_materials_write");
    }

    /// `TextureManager::evaluate` (rend3/src/managers/texture.rs): the whole bindless 2D array in the formats the loader
    /// produced; block formats are decoded and missing mips generated on the GPU (`MipmapSource::Generated`).
    pub fn textures_write_encoded(&self, descs: &[sys::r3n_texture_desc32], payload: &[u8]) {
        self.check(
            unsafe { sys::r3n_textures_write_encoded(self.ctx, descs.as_ptr(), descs.len() as u32, payload.as_ptr().cast::<c_void>(), payload.len() as u64) },
            "r3n_textures_write_encoded",
        );
    }

    /// `DirectionalLightManager::evaluate` / `PointLightManager::evaluate` (directional.rs:135-155, point.rs:58-74): the two
    /// storage buffers as `write_to_buffer` fills them.  Shadow cameras and the atlas allocation stay in Rust.
    pub fn lights_write(&self, directional: &[u8], point: &[u8]) {
        self.check(
            unsafe { sys::r3n_lights_write(self.ctx, directional.as_ptr().cast(), directional.len() as u64, point.as_ptr().cast(), point.len() as u64) },
            "r3n_lights_write",
        );
    }

    /// `batch_objects` with `Sorting::BLENDING` (rend3-routine/src/culling/batching.rs:146-176): the blend-key objects back to
    /// front, once per frame before the resolve.
    pub fn blend_order_write(&self, objects_back_to_front: &[u32]) {
        self.check(unsafe { sys::r3n_blend_order_write(self.ctx, objects_back_to_front.as_ptr(), objects_back_to_front.len() as u32) }, "r3n_blend_order_write");
    }
}

/// `Texture::format` -> the library's format id for `r3n_texture_desc32::format` (`include/r3n.h` R3N_TEXTURE_*): every format
/// rend3-gltf's maps produce (rend3-gltf/src/lib.rs:1157-1610) that a `texture_2d<f32>` binding can hold.  `None`: integer, depth,
/// ETC2 / EAC and ASTC formats -- `add_texture_2d` reports them instead of uploading.
pub fn texture_format_id(format: rend3::types::TextureFormat) -> Option<u32> {
    use rend3::types::TextureFormat as F;
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
