#!/usr/bin/env python3
"""Writes bindings/rend3-hooks.patch: the reference-side hooks of the MI355X-native object pipeline as an APPLICABLE patch against
the rend3 tree.  It makes `Renderer::new` create the `AmdContext` (one `r3n_ctx`) and every manager hand its
uploads to it, at the sites INTEGRATION.md section 2 names:

    rend3/src/util/amd.rs (new)            the context + the world's mirror (= bindings/rend3-hooks/amd.rs)
    rend3/src/renderer/setup.rs, mod.rs    Renderer::new creates the context; add_mesh / add_texture_2d mirror their data
    rend3/src/managers/mesh.rs             MeshManager::add            -> r3n_mesh_buffer_write per attribute run + indices
    rend3/src/managers/object.rs           evaluate::<M>               -> r3n_objects_write (stale ShaderObject<M> records)
    rend3/src/managers/material.rs         apply_buffer_gpu::<M>       -> r3n_materials_write (stale records + Material::key())
    rend3/src/util/freelist/buffer.rs      accessors for the stale list / reserved count the two mirrors above read
    rend3/src/renderer/eval.rs             TextureManager::evaluate    -> r3n_textures_write_encoded; texture removal
    rend3/src/managers/directional.rs, point.rs                        -> r3n_lights_write
    Cargo.toml, rend3/Cargo.toml, rend3/src/lib.rs                     the two crates of bindings/ as workspace members; `pub mod amd`

The edits are exact string replacements on the reference's files (each must match once: the script fails loudly when the
reference moves), the result is diffed against the originals.  Nothing of the reference is copied into this repository: the patch
holds the changed lines and their context only.

usage: python tools/make_hooks_patch.py [/root/reference]        (tests/test_reference_hooks.py applies the committed patch to a
copy of the reference with `git apply --check` and re-derives it)"""
import difflib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "bindings", "rend3-hooks.patch")
AMD_RS = os.path.join(ROOT, "bindings", "rend3-hooks", "amd.rs")

# (file, [(old, new), ...])
EDITS = [
    ("Cargo.toml", [('''    "rend3",
    "rend3-anim",
''', '''    "rend3",
    "rend3-amd-sys",      # bindings/rend3-amd-sys of the rend3_amd repository, copied beside the other crates
    "rend3-anim",
'''), ('''    "rend3-routine",
    "rend3-test",
''', '''    "rend3-routine",
    "rend3-routine-amd",  # bindings/rend3-routine-amd: the node bodies over librend3_amd.so
    "rend3-test",
''')]),
    ("rend3/Cargo.toml", [('''rend3-types = { version = "^0.3.0", path = "../rend3-types" }
''', '''rend3-amd-sys = { version = "0.1.0", path = "../rend3-amd-sys" }
rend3-types = { version = "^0.3.0", path = "../rend3-types" }
''')]),
    ("rend3/src/lib.rs", [('''pub mod util {
    pub mod bind_merge;
''', '''pub mod util {
    pub mod amd;
    pub mod bind_merge;
''')]),
    ("rend3/src/util/freelist/buffer.rs", [('''    pub fn use_index(&mut self, index: usize) {
''', '''    /// Indices written since the last `apply` (rend3-amd: the mirror of the same records, util/amd.rs).
    pub fn stale_indices(&self) -> &[usize] {
        &self.stale
    }

    /// Records the buffer will hold after the next `apply`.
    pub fn reserved_count(&self) -> usize {
        self.reserved_count
    }

    pub fn use_index(&mut self, index: usize) {
''')]),
    ("rend3/src/renderer/mod.rs", [('''    /// Tool which allows scatter uploads to happen.
    pub scatter: ScatterCopy,
}
''', '''    /// Tool which allows scatter uploads to happen.
    pub scatter: ScatterCopy,

    /// The MI355X-native object pipeline's context and its copy of the world (librend3_amd.so).
    pub amd: Arc<crate::util::amd::AmdContext>,
}
'''), ('''        let internal_mesh = self.mesh_manager.add(&self.device, mesh)?;
''', '''        let internal_mesh = self.mesh_manager.add(&self.device, &self.amd, mesh)?;
'''), ('''        let (cmd_buf, internal_texture) = TextureManager::<Texture2DTag>::add(self, texture, false)?;

        // Handle allocation must be done _after_ any validation to prevent deletion of a handle that never gets fully added.
        let handle = self.resource_handle_allocators.d2_texture.allocate(self);
''', '''        // rend3-amd: the CPU-side data is consumed by the wgpu upload below; the mirror takes its copy first
        let amd_copy = texture.clone();
        let (cmd_buf, internal_texture) = TextureManager::<Texture2DTag>::add(self, texture, false)?;

        // Handle allocation must be done _after_ any validation to prevent deletion of a handle that never gets fully added.
        let handle = self.resource_handle_allocators.d2_texture.allocate(self);
        self.amd.texture_fill(handle.idx, &amd_copy);
''')]),
    ("rend3/src/renderer/setup.rs", [('''    let scatter = ScatterCopy::new(&iad.device);

''', '''    let scatter = ScatterCopy::new(&iad.device);

    // rend3-amd: one r3n_ctx (a HIP device and its streams) beside the wgpu device, for the lifetime of the renderer
    let amd = crate::util::amd::AmdContext::from_env().unwrap_or_else(|e| panic!("r3n_create: {e}"));

'''), ('''        mipmap_generator,
        scatter,
    }))
''', '''        mipmap_generator,
        scatter,
        amd,
    }))
''')]),
    ("rend3/src/renderer/eval.rs", [('''                InstructionKind::DeleteTexture2D { handle } => {
                    renderer.resource_handle_allocators.d2_texture.deallocate(handle);
                    data_core.d2_texture_manager.remove(handle)
''', '''                InstructionKind::DeleteTexture2D { handle } => {
                    renderer.resource_handle_allocators.d2_texture.deallocate(handle);
                    renderer.amd.texture_remove(handle.idx);
                    data_core.d2_texture_manager.remove(handle)
'''), ('''    data_core.object_manager.evaluate(&renderer.device, &mut encoder, &renderer.scatter, &delayed_object_handles);

    // Level 2
    let d2_texture = data_core.d2_texture_manager.evaluate(&renderer.device);
''', '''    data_core.object_manager.evaluate(
        &renderer.device,
        &mut encoder,
        &renderer.scatter,
        &renderer.amd,
        &delayed_object_handles,
    );

    // Level 2
    let d2_texture = data_core.d2_texture_manager.evaluate(&renderer.device);
    // rend3-amd: the bindless 2D array, when it changed (before the materials that carry its indices)
    renderer.amd.textures_flush();
'''), ('''        &renderer.scatter,
        renderer.profile,
        &data_core.d2_texture_manager,
    );
''', '''        &renderer.scatter,
        &renderer.amd,
        renderer.profile,
        &data_core.d2_texture_manager,
    );
''')]),
    ("rend3/src/managers/mesh.rs", [('''    pub fn add(&self, device: &Device, mesh: Mesh) -> Result<InternalMesh, MeshCreationError> {
''', '''    pub fn add(
        &self,
        device: &Device,
        amd: &crate::util::amd::AmdContext,
        mesh: Mesh,
    ) -> Result<InternalMesh, MeshCreationError> {
'''), ('''            upload.add(range.start, attribute.untyped_data());
''', '''            upload.add(range.start, attribute.untyped_data());
            amd.mesh_buffer_write(range.start, attribute.untyped_data());
'''), ('''        upload.add(index_range.start, bytemuck::cast_slice(&mesh.indices));
''', '''        upload.add(index_range.start, bytemuck::cast_slice(&mesh.indices));
        amd.mesh_buffer_write(index_range.start, bytemuck::cast_slice(&mesh.indices));
''')]),
    ("rend3/src/managers/object.rs", [('''    evaluate: fn(&mut ObjectArchetype, &Device, &mut CommandEncoder, &ScatterCopy, &[RawObjectHandle]),
''', '''    evaluate: fn(&mut ObjectArchetype, &Device, &mut CommandEncoder, &ScatterCopy, &AmdContext, &[RawObjectHandle]),
'''), ('''        scatter: &ScatterCopy,
        deferred_removals: &[RawObjectHandle],
    ) {
        for archetype in self.archetype.values_mut() {
            (archetype.evaluate)(archetype, device, encoder, scatter, deferred_removals);
''', '''        scatter: &ScatterCopy,
        amd: &AmdContext,
        deferred_removals: &[RawObjectHandle],
    ) {
        for archetype in self.archetype.values_mut() {
            (archetype.evaluate)(archetype, device, encoder, scatter, amd, deferred_removals);
'''), ('''    scatter: &ScatterCopy,
    deferred_removals: &[RawObjectHandle],
) {
    let data_vec = archetype.data_vec.downcast_slice_mut::<Option<InternalObject<M>>>().unwrap();
''', '''    scatter: &ScatterCopy,
    amd: &AmdContext,
    deferred_removals: &[RawObjectHandle],
) {
    let data_vec = archetype.data_vec.downcast_slice_mut::<Option<InternalObject<M>>>().unwrap();
'''), ('''    archetype.buffer.apply(device, encoder, scatter, |idx| data_vec[idx].as_ref().map(|o| o.inner).unwrap_or_default())
''', '''    // rend3-amd: the same records, in the same byte layout, into the library's world (r3n_objects_write)
    let stale: Vec<u32> = archetype.buffer.stale_indices().iter().map(|&idx| idx as u32).collect();
    amd.objects_write(
        &stale,
        stale.iter().map(|&idx| data_vec[idx as usize].as_ref().map(|o| o.inner).unwrap_or_default()),
        archetype.buffer.reserved_count() as u32,
    );

    archetype.buffer.apply(device, encoder, scatter, |idx| data_vec[idx].as_ref().map(|o| o.inner).unwrap_or_default())
'''), ('''    util::{
        freelist::FreelistDerivedBuffer, frustum::BoundingSphere, iter::ExactSizerIterator, scatter_copy::ScatterCopy,
        typedefs::FastHashMap,
    },
''', '''    util::{
        amd::AmdContext, freelist::FreelistDerivedBuffer, frustum::BoundingSphere, iter::ExactSizerIterator,
        scatter_copy::ScatterCopy, typedefs::FastHashMap,
    },
''')]),
    ("rend3/src/managers/material.rs", [('''    util::{
        bind_merge::BindGroupLayoutBuilder, freelist::FreelistDerivedBuffer, math::round_up, scatter_copy::ScatterCopy,
        typedefs::FastHashMap,
    },
''', '''    util::{
        amd::AmdContext, bind_merge::BindGroupLayoutBuilder, freelist::FreelistDerivedBuffer, math::round_up,
        scatter_copy::ScatterCopy, typedefs::FastHashMap,
    },
'''), ('''        &mut CommandEncoder,
        &ScatterCopy,
        &mut WasmVecAny,
        &TextureManager<crate::types::Texture2DTag>,
    ),
''', '''        &mut CommandEncoder,
        &ScatterCopy,
        &AmdContext,
        &mut WasmVecAny,
        &TextureManager<crate::types::Texture2DTag>,
    ),
'''), ('''        scatter: &ScatterCopy,
        profile: RendererProfile,
        texture_manager: &TextureManager<crate::types::Texture2DTag>,
    ) {
''', '''        scatter: &ScatterCopy,
        amd: &AmdContext,
        profile: RendererProfile,
        texture_manager: &TextureManager<crate::types::Texture2DTag>,
    ) {
'''), ('''                    encoder,
                    scatter,
                    &mut archetype.data_vec,
                    texture_manager,
                ),
''', '''                    encoder,
                    scatter,
                    amd,
                    &mut archetype.data_vec,
                    texture_manager,
                ),
'''), ('''    scatter: &ScatterCopy,
    data_vec: &mut WasmVecAny,
    texture_manager: &TextureManager<crate::types::Texture2DTag>,
) {
    let data_vec = data_vec.downcast_slice::<Option<InternalMaterial<M>>>().unwrap();

    let translation_fn = texture_manager.translation_fn();

''', '''    scatter: &ScatterCopy,
    amd: &AmdContext,
    data_vec: &mut WasmVecAny,
    texture_manager: &TextureManager<crate::types::Texture2DTag>,
) {
    let data_vec = data_vec.downcast_slice::<Option<InternalMaterial<M>>>().unwrap();

    let translation_fn = texture_manager.translation_fn();

    // rend3-amd: the same records into the library's world, with Material::key() of each (host-side state in rend3, a byte per
    // record there: r3n_materials_write)
    let stale: Vec<u32> = buffer.stale_indices().iter().map(|&idx| idx as u32).collect();
    let keys: Vec<u8> = stale.iter().map(|&idx| data_vec[idx as usize].as_ref().unwrap().inner.key() as u8).collect();
    amd.materials_write(
        &stale,
        stale.iter().map(|&idx| {
            let material = &data_vec[idx as usize].as_ref().unwrap().inner;
            GpuPoweredShaderWrapper::<M> {
                textures: material
                    .to_textures()
                    .map_to_u32(|handle_opt| handle_opt.map(translation_fn).map_or(0, NonZeroU32::get)),
                data: material.to_data(),
            }
        }),
        &keys,
    );

''')]),
    ("rend3/src/managers/directional.rs", [('''        self.data_buffer.write_to_buffer(&renderer.device, &renderer.queue, &buffer);

        (new_shadow_map_size, shadow_data)
''', '''        self.data_buffer.write_to_buffer(&renderer.device, &renderer.queue, &buffer);
        renderer.amd.lights_write_directional(&buffer);  // rend3-amd: the same bytes

        (new_shadow_map_size, shadow_data)
''')]),
    ("rend3/src/managers/point.rs", [('''        self.data_buffer.write_to_buffer(&renderer.device, &renderer.queue, &buffer);
    }
''', '''        self.data_buffer.write_to_buffer(&renderer.device, &renderer.queue, &buffer);
        renderer.amd.lights_write_point(&buffer);  // rend3-amd: the same bytes
    }
''')]),
]
NEW_FILES = [("rend3/src/util/amd.rs", AMD_RS)]


def patched_files(ref):
    """{path: (old text or None, new text)}"""
    out = {}
    for rel, edits in EDITS:
        old = open(os.path.join(ref, rel)).read()
        new = old
        for a, b in edits:
            if new.count(a) != 1:
                raise SystemExit(f"{rel}: the anchor below matches {new.count(a)} times (the reference moved?)\n{a}")
            new = new.replace(a, b)
        out[rel] = (old, new)
    for rel, src in NEW_FILES:
        out[rel] = (None, open(src).read())
    return out


def make(ref):
    chunks = []
    for rel, (old, new) in patched_files(ref).items():
        if old is None:
            lines = new.splitlines(keepends=True)
            chunks.append(f"diff --git a/{rel} b/{rel}\nnew file mode 100644\n--- /dev/null\n+++ b/{rel}\n@@ -0,0 +1,{len(lines)} @@\n" + "".join("+" + l for l in lines))
        else:
            diff = "".join(difflib.unified_diff(old.splitlines(keepends=True), new.splitlines(keepends=True), f"a/{rel}", f"b/{rel}", n=3))
            chunks.append(f"diff --git a/{rel} b/{rel}\n" + diff)
    return "".join(chunks)


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    text = make(ref)
    open(PATCH, "w").write(text)
    print(PATCH, len(text.splitlines()), "lines,", text.count("\ndiff --git") + 1, "files")
