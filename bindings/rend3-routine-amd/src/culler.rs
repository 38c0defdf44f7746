//! rend3-routine/src/culling/culler.rs:185-714 -- `GpuCuller` with the reference's constructor and graph entry points.
use crate::amd::AmdContext;
use glam::UVec2;
use rend3::graph::{DataHandle, RenderGraph, RenderTargetHandle};
use rend3::types::{GraphDataHandle, Material, SampleCount};
use rend3::{Renderer, ShaderPreProcessor};
use rend3_amd_sys as sys;
use rend3_routine::common::CameraSpecifier;
use rend3_routine::culling::{CullingBufferMap, DrawCallSet, PerCameraUniform};
use std::sync::Arc;

/// The reference's culler owns the K1 / K2 pipelines, the per-camera ping-pong buffers and `PerCameraPreviousInvocationsMap`
/// (culler.rs:185-197); all of that state lives inside the `r3n_ctx` now.  `culling_buffer_map_handle` stays a public field
/// because `PbrRoutine::new` takes it (pbr/routine.rs:40); nothing reads the map.
pub struct GpuCuller {
    pub amd: Arc<AmdContext>,
    pub culling_buffer_map_handle: GraphDataHandle<CullingBufferMap>,
    /// culler.rs:133-141: front face / cull mode folded into the header's flags exactly as before
    winding: rend3::types::Handedness,
}

impl GpuCuller {
    /// culler.rs:199 -- same signature; the two compute pipelines it compiles there do not exist here.
    pub fn new<M>(renderer: &Arc<Renderer>, _spp: &ShaderPreProcessor) -> Self
    where
        M: Material,
    {
        Self { amd: AmdContext::of(renderer), culling_buffer_map_handle: renderer.add_graph_data(CullingBufferMap::default()), winding: renderer.handedness }
    }

    /// culler.rs:661-680 (`object_uniform_upload`, :427-529): header upload + K1 (`uniform_prep.wgsl`).
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

    /// culler.rs:682-713: `batch_objects` (CPU frustum cull + sort + 256-object batches, batching.rs:120-250) followed by one
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
