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
