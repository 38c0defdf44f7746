//! rend3-routine/src/culling/culler.rs:185-714 -- `GpuCuller` with the reference's constructor and graph entry points.
use crate::amd::AmdContext;
use glam::UVec2;
use rend3::graph::{DataHandle, RenderGraph, RenderTargetHandle};
use rend3::types::{GraphDataHandle, Material, SampleCount};
use rend3::{Renderer, ShaderPreProcessor};
use rend3_amd_sys as sys;
use rend3_routine::common::CameraSpecifier;
use rend3_routine::culling::{CullingBufferMap, DrawCallSet};
use std::sync::Arc;

/// The 240-byte head of the reference's per-camera uniform buffer (`PerCameraUniform`, culler.rs:157-174, a PRIVATE struct there):
/// restated field for field so that `encase` lays it out identically -- view @0, view_proj @64, shadow_index @128, frustum @144
/// (five planes), resolution @224, flags @232, object_count @236 (`static_assert`s in csrc/layouts.h).
/// Every input is public reference API: `CameraState::{view, view_proj, world_frustum}` (rend3/src/managers/camera.rs:63-81),
/// `rend3::util::frustum::Frustum` (a `ShaderType`), `CameraSpecifier::to_shader_index` (rend3-routine/src/common/camera.rs:27).
#[derive(encase::ShaderType)]
struct PerCameraHeader {
    view: glam::Mat4,
    view_proj: glam::Mat4,
    shadow_index: u32,
    frustum: rend3::util::frustum::Frustum,
    resolution: glam::Vec2,
    flags: u32,
    object_count: u32,
}
const POSITIVE_AREA_VISIBLE: u32 = 1 << 0; // culler.rs:150-155 (`PerCameraUniformFlags`, private there)
const MULTISAMPLED: u32 = 1 << 1;

/// culler.rs:133-149 (`TriangleVisibility::from_winding_and_face(..).is_positive()`, private there), restated.
fn positive_area_visible(winding: wgpu::FrontFace, culling: wgpu::Face) -> bool {
    matches!((winding, culling), (wgpu::FrontFace::Ccw, wgpu::Face::Back) | (wgpu::FrontFace::Cw, wgpu::Face::Front))
}

/// The reference's culler owns the K1 / K2 pipelines, the per-camera ping-pong buffers and `PerCameraPreviousInvocationsMap`
/// (culler.rs:185-197); all of that state lives inside the `r3n_ctx` now.  `culling_buffer_map_handle` stays a public field
/// because `PbrRoutine::new` takes it (pbr/routine.rs:40); nothing reads the map.
pub struct GpuCuller {
    pub amd: Arc<AmdContext>,
    pub culling_buffer_map_handle: GraphDataHandle<CullingBufferMap>,
    /// culler.rs:189,318: `renderer.handedness.into()`; folded into the header's flags exactly as before
    winding: wgpu::FrontFace,
}

impl GpuCuller {
    /// culler.rs:199 -- same signature; the two compute pipelines it compiles there do not exist here.
    pub fn new<M>(renderer: &Arc<Renderer>, _spp: &ShaderPreProcessor) -> Self
    where
        M: Material,
    {
        Self { amd: AmdContext::of(renderer), culling_buffer_map_handle: renderer.add_graph_data(CullingBufferMap::default()), winding: renderer.handedness.into() }
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
            // culler.rs:446-451: the object capacity is the object buffer's size over the record size (public API on both sides);
            // nothing to draw is a silent return, as there
            let max_object_count = ctx.data_core.object_manager.buffer::<M>().map(wgpu::Buffer::size).unwrap_or(0)
                / <rend3::managers::ShaderObject<M> as encase::ShaderSize>::SHADER_SIZE.get();
            if max_object_count == 0 {
                return;
            }
            // culling face per camera kind: culler.rs:478-481; the header's fields: culler.rs:485-502
            let culling = match camera_specifier {
                CameraSpecifier::Shadow(_) => wgpu::Face::Front,
                CameraSpecifier::Viewport => wgpu::Face::Back,
            };
            let header = PerCameraHeader {
                view: camera.view(),
                view_proj: camera.view_proj(),
                shadow_index: camera_specifier.to_shader_index(),
                frustum: camera.world_frustum(),
                resolution: resolution.as_vec2(),
                flags: (if positive_area_visible(self.winding, culling) { POSITIVE_AREA_VISIBLE } else { 0 })
                    | (if samples != SampleCount::One { MULTISAMPLED } else { 0 }),
                object_count: max_object_count as u32,
            };
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
        node.build(move |ctx| {
            // culler.rs:705-707: the same early return as the bake node above -- with no object buffer the bake did not run this
            // frame, and r3n_cull refuses (R3N_ERR_STATE) a camera whose header belongs to an earlier frame
            let max_object_count = ctx.data_core.object_manager.buffer::<M>().map(wgpu::Buffer::size).unwrap_or(0)
                / <rend3::managers::ShaderObject<M> as encase::ShaderSize>::SHADER_SIZE.get();
            if max_object_count == 0 {
                return;
            }
            self.amd.check(unsafe { sys::r3n_cull(self.amd.ctx, camera_specifier.to_shader_index()) }, "r3n_cull");
        });
    }
}
