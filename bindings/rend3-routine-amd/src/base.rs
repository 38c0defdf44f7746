//! rend3-routine/src/base.rs:103-186 -- `BaseRenderGraph::add_to_graph` with the reference's node order.
use crate::amd::AmdContext;
use crate::culler::GpuCuller;
use crate::forward::{CullingSource, ForwardRoutine, ForwardRoutineArgs, RoutineType};
use crate::hi_z::HiZRoutine;
use crate::skinning::GpuSkinner;
use crate::tonemapping::TonemappingRoutine;
use crate::uniforms;
use glam::{UVec2, Vec4};
use rend3::graph::RenderGraph;
use rend3::types::SampleCount;
use rend3::InstructionEvaluationOutput;
use rend3_amd_sys as sys;
use rend3_routine::common::CameraSpecifier;
use rend3_routine::pbr::{PbrMaterial, TransparencyType};

/// rend3-routine/src/pbr/routine.rs:17-133
pub struct PbrRoutine<'a> {
    pub opaque_depth: ForwardRoutine<'a, PbrMaterial>,
    pub cutout_depth: ForwardRoutine<'a, PbrMaterial>,
    pub opaque_routine: ForwardRoutine<'a, PbrMaterial>,
    pub cutout_routine: ForwardRoutine<'a, PbrMaterial>,
    pub blend_routine: ForwardRoutine<'a, PbrMaterial>,
    pub hi_z: HiZRoutine<'a>,
}

impl<'a> PbrRoutine<'a> {
    pub fn new(amd: &'a AmdContext) -> Self {
        let key = |t: TransparencyType| t as u64;
        Self {
            opaque_depth: ForwardRoutine::new(amd, RoutineType::Depth, key(TransparencyType::Opaque)),
            cutout_depth: ForwardRoutine::new(amd, RoutineType::Depth, key(TransparencyType::Cutout)),
            opaque_routine: ForwardRoutine::new(amd, RoutineType::Forward, key(TransparencyType::Opaque)),
            cutout_routine: ForwardRoutine::new(amd, RoutineType::Forward, key(TransparencyType::Cutout)),
            blend_routine: ForwardRoutine::new(amd, RoutineType::Forward, key(TransparencyType::Blend)),
            hi_z: HiZRoutine { amd },
        }
    }
}

/// base.rs:76-80
pub struct OutputRenderTarget {
    pub resolution: UVec2,
    pub samples: SampleCount,
}

/// base.rs:82-86 (skybox: out of scope of this path)
pub struct BaseRenderGraphRoutines<'node> {
    pub pbr: &'node PbrRoutine<'node>,
    pub tonemapping: &'node TonemappingRoutine<'node>,
}

/// base.rs:88-92
pub struct BaseRenderGraphInputs<'a, 'node> {
    pub eval_output: &'a InstructionEvaluationOutput,
    pub routines: BaseRenderGraphRoutines<'node>,
    pub target: OutputRenderTarget,
}

/// base.rs:94-98
#[derive(Debug, Default)]
pub struct BaseRenderGraphSettings {
    pub ambient_color: Vec4,
    pub clear_color: Vec4,
}

/// base.rs:103-108
pub struct BaseRenderGraph<'a> {
    pub amd: &'a AmdContext,
    pub gpu_culler: GpuCuller<'a>,
    pub gpu_skinner: GpuSkinner<'a>,
}

impl<'a> BaseRenderGraph<'a> {
    /// base.rs:111-124
    pub fn new(amd: &'a AmdContext, handedness: rend3::types::Handedness) -> Self {
        Self { amd, gpu_culler: GpuCuller::new::<PbrMaterial>(amd, handedness), gpu_skinner: GpuSkinner { amd } }
    }

    /// base.rs:129-185.  Same nodes, same order; the one addition is "Resolve Opaque": the opaque passes write visibility keys,
    /// and their fragment shaders run once per pixel after pass 2.
    pub fn add_to_graph<'node>(&'node self, graph: &mut RenderGraph<'node>, inputs: BaseRenderGraphInputs<'_, 'node>, settings: BaseRenderGraphSettings) {
        let amd = self.amd;
        let pbr = inputs.routines.pbr;
        let (resolution, samples) = (inputs.target.resolution, inputs.target.samples);
        // clear_shadow_buffers + create_frame_uniforms (base.rs:139,142)
        uniforms::add_to_graph(graph, amd, uniforms::UniformInformation { ambient: settings.ambient_color, resolution, samples, clear_color: settings.clear_color });
        // skinning (base.rs:145)
        self.gpu_skinner.add_skinning_to_graph(graph);
        // shadow_object_uniform_upload (base.rs:148)
        for (i, shadow) in inputs.eval_output.shadows.iter().enumerate() {
            self.gpu_culler.add_object_uniform_upload_to_graph::<PbrMaterial>(graph, CameraSpecifier::Shadow(i as u32), UVec2::splat(shadow.map.size), SampleCount::One, &format!("Shadow Culling S{i}"));
        }
        // pbr_shadow_culling (base.rs:150)
        for i in 0..inputs.eval_output.shadows.len() {
            self.gpu_culler.add_culling_to_graph::<PbrMaterial>(graph, Default::default(), Default::default(), CameraSpecifier::Shadow(i as u32), &format!("Shadow Culling S{i}"));
        }
        // pbr_shadow_rendering (base.rs:153,366-396)
        for i in 0..inputs.eval_output.shadows.len() {
            for routine in [&pbr.opaque_depth, &pbr.cutout_depth] {
                routine.add_forward_to_graph(ForwardRoutineArgs { graph, label: &format!("pbr shadow renderering S{i}"), camera: CameraSpecifier::Shadow(i as u32), culling_source: CullingSource::Residual, samples: SampleCount::One });
            }
        }
        // object_uniform_upload (base.rs:156)
        self.gpu_culler.add_object_uniform_upload_to_graph::<PbrMaterial>(graph, CameraSpecifier::Viewport, resolution, samples, "Uniform Bake");
        // pbr_render_opaque_predicted_triangles (base.rs:159)
        for routine in [&pbr.opaque_routine, &pbr.cutout_routine] {
            routine.add_forward_to_graph(ForwardRoutineArgs { graph, label: "PBR Forward Pass 1", camera: CameraSpecifier::Viewport, culling_source: CullingSource::Predicted, samples });
        }
        // hi_z (base.rs:162)
        pbr.hi_z.add_hi_z_to_graph(graph);
        // pbr_culling (base.rs:169)
        self.gpu_culler.add_culling_to_graph::<PbrMaterial>(graph, Default::default(), Default::default(), CameraSpecifier::Viewport, "Primary Culling");
        // pbr_render_opaque_residual_triangles (base.rs:172)
        for routine in [&pbr.opaque_routine, &pbr.cutout_routine] {
            routine.add_forward_to_graph(ForwardRoutineArgs { graph, label: "PBR Forward Pass 2", camera: CameraSpecifier::Viewport, culling_source: CullingSource::Residual, samples });
        }
        // the deferred evaluation of the opaque passes' fragments
        let mut resolve = graph.add_node("Resolve Opaque");
        resolve.add_side_effect();
        resolve.build(move |_ctx| amd.check(unsafe { sys::r3n_resolve_opaque(amd.ctx) }, "r3n_resolve_opaque"));
        // skybox (base.rs:175): not on this path.  pbr_forward_rendering_transparent (base.rs:181,451-465)
        pbr.blend_routine.add_forward_to_graph(ForwardRoutineArgs { graph, label: "PBR Forward Transparent", camera: CameraSpecifier::Viewport, culling_source: CullingSource::Residual, samples });
        // tonemapping (base.rs:184)
        inputs.routines.tonemapping.add_to_graph(graph, None);
        let mut end = graph.add_node("Frame End");
        end.add_side_effect();
        end.build(move |_ctx| amd.frame_end());
    }
}
