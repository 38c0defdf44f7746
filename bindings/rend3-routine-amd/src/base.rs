//! rend3-routine/src/base.rs:72-186 and pbr/routine.rs:17-133 -- `BaseRenderGraph` / `PbrRoutine` with the reference's
//! constructors and `add_to_graph`, node for node in the reference's order.
use crate::amd::AmdContext;
use crate::culler::GpuCuller;
use crate::forward::{CullingSource, ForwardRoutine, ForwardRoutineArgs, ForwardRoutineBindingData, ForwardRoutineCreateArgs, RoutineType};
use crate::hi_z::HiZRoutine;
use crate::skinning::{self, GpuSkinner};
use crate::tonemapping::TonemappingRoutine;
use crate::uniforms;
use glam::{UVec2, Vec4};
use rend3::graph::{DataHandle, RenderGraph, RenderPassTargets, RenderTargetHandle};
use rend3::types::{GraphDataHandle, SampleCount};
use rend3::graph::InstructionEvaluationOutput;
use rend3::{Renderer, RendererDataCore, ShaderPreProcessor};
use rend3_amd_sys as sys;
use rend3_routine::common::{self, CameraSpecifier, PerMaterialArchetypeInterface, WholeFrameInterfaces};
use rend3_routine::culling::CullingBufferMap;
use rend3_routine::forward::ShaderModulePair;
use rend3_routine::pbr::{PbrMaterial, TransparencyType};
use std::sync::Arc;

pub use rend3_routine::base::DepthTargets;

/// rend3-routine/src/pbr/routine.rs:17-27
pub struct PbrRoutine {
    pub opaque_depth: ForwardRoutine<PbrMaterial>,
    pub cutout_depth: ForwardRoutine<PbrMaterial>,
    pub opaque_routine: ForwardRoutine<PbrMaterial>,
    pub cutout_routine: ForwardRoutine<PbrMaterial>,
    pub blend_routine: ForwardRoutine<PbrMaterial>,
    pub hi_z: HiZRoutine,
    pub per_material: PerMaterialArchetypeInterface<PbrMaterial>,
}

impl PbrRoutine {
    /// pbr/routine.rs:35-41 -- same signature.  The five routines differ by (routine type, `TransparencyType` key) only; the
    /// shader modules the reference compiles here (:47-70) are what the library's rasteriser + resolve implement.
    pub fn new(
        renderer: &Arc<Renderer>,
        data_core: &mut RendererDataCore,
        spp: &ShaderPreProcessor,
        interfaces: &WholeFrameInterfaces,
        culling_buffer_map_handle: &GraphDataHandle<CullingBufferMap>,
    ) -> Self {
        data_core.material_manager.ensure_archetype::<PbrMaterial>(&renderer.device, renderer.profile);
        let per_material = PerMaterialArchetypeInterface::<PbrMaterial>::new(&renderer.device);
        let mut inner = |routine_type, transparency: TransparencyType, name| {
            ForwardRoutine::new(ForwardRoutineCreateArgs {
                name,
                renderer,
                data_core,
                spp,
                interfaces,
                per_material: &per_material,
                material_key: transparency as u64,
                routine_type,
                shaders: ShaderModulePair::none(),
                culling_buffer_map_handle: culling_buffer_map_handle.clone(),
                extra_bgls: &[],
                descriptor_callback: None,
            })
        };
        Self {
            opaque_depth: inner(RoutineType::Depth, TransparencyType::Opaque, "Shadow Depth Opaque"),
            cutout_depth: inner(RoutineType::Depth, TransparencyType::Cutout, "Shadow Depth Cutout"),
            opaque_routine: inner(RoutineType::Forward, TransparencyType::Opaque, "Opaque"),
            cutout_routine: inner(RoutineType::Forward, TransparencyType::Cutout, "Cutout"),
            blend_routine: inner(RoutineType::Forward, TransparencyType::Blend, "Forward Blend"),
            hi_z: HiZRoutine::new(renderer, spp),
            per_material,
        }
    }
}

/// base.rs:75-79
pub struct OutputRenderTarget {
    pub handle: RenderTargetHandle,
    pub resolution: UVec2,
    pub samples: SampleCount,
}

/// base.rs:81-85 (the skybox routine is rend3-routine's own: not on this path, its node is skipped)
pub struct BaseRenderGraphRoutines<'node> {
    pub pbr: &'node PbrRoutine,
    pub skybox: Option<&'node rend3_routine::skybox::SkyboxRoutine>,
    pub tonemapping: &'node TonemappingRoutine,
}

/// base.rs:87-91
pub struct BaseRenderGraphInputs<'a, 'node> {
    pub eval_output: &'a InstructionEvaluationOutput,
    pub routines: BaseRenderGraphRoutines<'node>,
    pub target: OutputRenderTarget,
}

/// base.rs:93-97
#[derive(Debug, Default)]
pub struct BaseRenderGraphSettings {
    pub ambient_color: Vec4,
    pub clear_color: Vec4,
}

/// base.rs:103-108
pub struct BaseRenderGraph {
    pub interfaces: common::WholeFrameInterfaces,
    pub samplers: common::Samplers,
    pub gpu_culler: GpuCuller,
    pub gpu_skinner: GpuSkinner,
}

impl BaseRenderGraph {
    /// base.rs:111-124 -- same signature.  The `AmdContext` is created here (first `AmdContext::of(renderer)`) and lives in the
    /// renderer's graph storage; `interfaces` / `samplers` stay because `PbrRoutine::new` / `TonemappingRoutine::new` take them.
    pub fn new(renderer: &Arc<Renderer>, spp: &ShaderPreProcessor) -> Self {
        let interfaces = common::WholeFrameInterfaces::new(&renderer.device);
        let samplers = common::Samplers::new(&renderer.device);
        let gpu_culler = GpuCuller::new::<PbrMaterial>(renderer, spp);
        let gpu_skinner = GpuSkinner::new(renderer, spp);
        Self { interfaces, samplers, gpu_culler, gpu_skinner }
    }

    /// base.rs:129-185 -- same signature, same nodes, same order; the one addition is "Resolve Opaque": the opaque passes write
    /// visibility keys, and their fragment shaders run once per pixel after pass 2.  (A host that does
    /// not need a graph between the nodes can issue the whole list with ONE call, `r3n_render_frame`.)
    #[allow(clippy::too_many_arguments)]
    pub fn add_to_graph<'node>(
        &'node self,
        graph: &mut RenderGraph<'node>,
        inputs: BaseRenderGraphInputs<'_, 'node>,
        settings: BaseRenderGraphSettings,
    ) {
        let amd = &self.gpu_culler.amd;
        let pbr = inputs.routines.pbr;
        let (resolution, samples) = (inputs.target.resolution, inputs.target.samples);
        // the handles the reference threads between its nodes: declared so the routine signatures are the reference's; the data
        // behind them lives in the context
        let depth = DepthTargets::new(graph, resolution, samples);
        let cull = graph.add_data();
        let uniform_bg: DataHandle<wgpu::BindGroup> = graph.add_data();
        let forward = |graph: &mut RenderGraph<'node>, routine: &'node ForwardRoutine<PbrMaterial>, label: &str, camera, culling_source, samples| {
            routine.add_forward_to_graph(ForwardRoutineArgs {
                graph,
                label,
                camera,
                binding_data: ForwardRoutineBindingData { whole_frame_uniform_bg: uniform_bg, per_material_bgl: &pbr.per_material, extra_bgs: None },
                culling_source,
                samples,
                renderpass: RenderPassTargets { targets: vec![], depth_stencil: None },
            })
        };
        // clear_shadow_buffers + create_frame_uniforms (base.rs:139,142)
        uniforms::add_to_graph(graph, amd, uniforms::UniformInformation { ambient: settings.ambient_color, resolution, samples, clear_color: settings.clear_color });
        // skinning (base.rs:145)
        skinning::add_skinning_to_graph(graph, &self.gpu_skinner);
        // shadow_object_uniform_upload (base.rs:148)
        for (i, shadow) in inputs.eval_output.shadows.iter().enumerate() {
            self.gpu_culler.add_object_uniform_upload_to_graph::<PbrMaterial>(graph, CameraSpecifier::Shadow(i as u32), UVec2::splat(shadow.map.size), SampleCount::One, &format!("Shadow Culling S{i}"));
        }
        // pbr_shadow_culling (base.rs:150)
        let shadow_cull: Vec<_> = (0..inputs.eval_output.shadows.len()).map(|_| graph.add_data()).collect();
        for (i, &handle) in shadow_cull.iter().enumerate() {
            self.gpu_culler.add_culling_to_graph::<PbrMaterial>(graph, handle, depth.rendering_target(), CameraSpecifier::Shadow(i as u32), &format!("Shadow Culling S{i}"));
        }
        // pbr_shadow_rendering (base.rs:153,366-396)
        for (i, &handle) in shadow_cull.iter().enumerate() {
            for routine in [&pbr.opaque_depth, &pbr.cutout_depth] {
                forward(graph, routine, &format!("pbr shadow renderering S{i}"), CameraSpecifier::Shadow(i as u32), CullingSource::Residual(handle), SampleCount::One);
            }
        }
        // object_uniform_upload (base.rs:156)
        self.gpu_culler.add_object_uniform_upload_to_graph::<PbrMaterial>(graph, CameraSpecifier::Viewport, resolution, samples, "Uniform Bake");
        // pbr_render_opaque_predicted_triangles (base.rs:159)
        for routine in [&pbr.opaque_routine, &pbr.cutout_routine] {
            forward(graph, routine, "PBR Forward Pass 1", CameraSpecifier::Viewport, CullingSource::Predicted, samples);
        }
        // hi_z (base.rs:162)
        pbr.hi_z.add_hi_z_to_graph(graph, depth, resolution);
        // pbr_culling (base.rs:169)
        self.gpu_culler.add_culling_to_graph::<PbrMaterial>(graph, cull, depth.rendering_target(), CameraSpecifier::Viewport, "Primary Culling");
        // pbr_render_opaque_residual_triangles (base.rs:172)
        for routine in [&pbr.opaque_routine, &pbr.cutout_routine] {
            forward(graph, routine, "PBR Forward Pass 2", CameraSpecifier::Viewport, CullingSource::Residual(cull), samples);
        }
        // the deferred evaluation of the opaque passes' fragments
        let mut resolve = graph.add_node("Resolve Opaque");
        resolve.add_side_effect();
        resolve.build(move |_ctx| amd.check(unsafe { sys::r3n_resolve_opaque(amd.ctx) }, "r3n_resolve_opaque"));
        // skybox (base.rs:175): not on this path.  pbr_forward_rendering_transparent (base.rs:181,451-465)
        forward(graph, &pbr.blend_routine, "PBR Forward Transparent", CameraSpecifier::Viewport, CullingSource::Residual(cull), samples);
        // tonemapping (base.rs:184)
        inputs.routines.tonemapping.add_to_graph(graph, depth.rendering_target(), inputs.target.handle, uniform_bg);
        let mut end = graph.add_node("Frame End");
        end.add_side_effect();
        end.build(move |_ctx| amd.frame_end());
    }
}
