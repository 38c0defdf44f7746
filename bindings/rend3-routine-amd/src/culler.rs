//! rend3-routine/src/culling/culler.rs:185-714 -- `GpuCuller` with the same two graph entry points.
use crate::amd::AmdContext;
use glam::UVec2;
use rend3::graph::{DataHandle, RenderGraph, RenderTargetHandle};
use rend3::types::{Material, SampleCount};
use rend3_amd_sys as sys;
use rend3_routine::common::CameraSpecifier;
use rend3_routine::culling::{DrawCallSet, PerCameraUniform};
use std::sync::Arc;

/// The reference's culler owns the K1 / K2 pipelines, the per-camera ping-pong buffers and `PerCameraPreviousInvocationsMap`
/// (culler.rs:185-197); all of that state lives inside the `r3n_ctx` now, so this is a handle.
pub struct GpuCuller<'a> {
    pub amd: &'a AmdContext,
    /// culler.rs:133-141: front-face / cull-mode folded into the header's flags by the caller exactly as before
    pub winding: rend3::types::Handedness,
}

impl<'a> GpuCuller<'a> {
    /// culler.rs:198-425 compiles the two compute pipelines; nothing to build here.
    pub fn new<M: Material>(amd: &'a AmdContext, winding: rend3::types::Handedness) -> Self {
        Self { amd, winding }
    }

    /// culler.rs:661-695 (`object_uniform_upload`, :427-529): header upload + K1 (`uniform_prep.wgsl`).
    pub fn add_object_uniform_upload_to_graph<'node, M: Material>(
        &'node self,
        graph: &mut RenderGraph<'node>,
        camera_specifier: CameraSpecifier,
        resolution: UVec2,
        samples: SampleCount,
        name: &str,
    ) {
        let mut node = graph.add_node(name);
        node.add_side_effect();
        node.build(move |ctx| {
            let camera = match camera_specifier {
                CameraSpecifier::Shadow(index) => &ctx.eval_output.shadows[index as usize].camera,
                CameraSpecifier::Viewport => &ctx.data_core.viewport_camera_state,
            };
            // identical to culler.rs:485-502: view, view_proj, frustum, resolution, flags, object count, shadow index -- 240 B
            let header = PerCameraUniform::header(camera, camera_specifier, resolution, samples, self.winding, ctx.data_core.object_manager.buffer::<M>().map_or(0, |b| b.reserved_count()));
            let mut bytes = [0u8; 240];
            encase::StorageBuffer::new(&mut bytes[..]).write(&header).unwrap();
            self.amd.check(unsafe { sys::r3n_uniform_bake(self.amd.ctx, camera_specifier.to_shader_index(), bytes.as_ptr().cast()) }, "r3n_uniform_bake");
        });
    }

    /// culler.rs:697-712: `batch_objects` (CPU frustum cull + sort + 256-object batches, batching.rs:120-250) followed by one
    /// K2 dispatch per batch (:531-659).  Both halves run on the GPU behind one call; the handles stay in the signature so call
    /// sites compile unchanged, the draw-call set is owned by the context.
    pub fn add_culling_to_graph<'node, M: Material>(
        &'node self,
        graph: &mut RenderGraph<'node>,
        _draw_calls_hdl: DataHandle<Arc<DrawCallSet>>,
        _depth_handle: RenderTargetHandle,
        camera_specifier: CameraSpecifier,
        name: &str,
    ) {
        let mut node = graph.add_node(name);
        node.add_side_effect();
        node.build(move |_ctx| {
            self.amd.check(unsafe { sys::r3n_cull(self.amd.ctx, camera_specifier.to_shader_index()) }, "r3n_cull");
        });
    }
}
