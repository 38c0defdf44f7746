//! rend3-routine/src/forward.rs:135-316 -- one (routine type, material key) pipeline behind the reference's
//! `ForwardRoutineCreateArgs` / `ForwardRoutineArgs` (forward.rs:72-133): same fields, so call sites build them unchanged; the
//! wgpu-only members (shader modules, bind-group layouts, render-pass targets, bind groups) are carried and not read.
use crate::amd::AmdContext;
use rend3::graph::{DataHandle, RenderGraph, RenderPassTargets};
use rend3::types::{GraphDataHandle, Material, SampleCount};
use rend3::{Renderer, RendererDataCore, ShaderPreProcessor};
use rend3_amd_sys as sys;
use rend3_routine::common::{CameraSpecifier, PerMaterialArchetypeInterface, WholeFrameInterfaces};
use rend3_routine::culling::{CullingBufferMap, DrawCallSet};
use rend3_routine::forward::ShaderModulePair;
use std::marker::PhantomData;
use std::sync::Arc;
use wgpu::{BindGroup, BindGroupLayout, ColorTargetState, RenderPipelineDescriptor};

/// forward.rs:40-44
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum RoutineType {
    Depth = sys::R3N_PASS_DEPTH as isize,
    Forward = sys::R3N_PASS_FORWARD as isize,
}

/// forward.rs:64-70
#[derive(Clone, Copy)]
pub enum CullingSource {
    Predicted,
    Residual(DataHandle<Arc<DrawCallSet>>),
}

/// forward.rs:84-105
pub struct ForwardRoutineCreateArgs<'a, M> {
    pub name: &'a str,

    pub renderer: &'a Arc<Renderer>,
    pub data_core: &'a mut RendererDataCore,
    pub spp: &'a ShaderPreProcessor,

    pub interfaces: &'a WholeFrameInterfaces,
    pub per_material: &'a PerMaterialArchetypeInterface<M>,
    pub material_key: u64,

    pub routine_type: RoutineType,
    pub shaders: ShaderModulePair<'a>,

    pub culling_buffer_map_handle: GraphDataHandle<CullingBufferMap>,

    pub extra_bgls: &'a [&'a BindGroupLayout],
    #[allow(clippy::type_complexity)]
    pub descriptor_callback: Option<&'a dyn Fn(&mut RenderPipelineDescriptor<'_>, &mut [Option<ColorTargetState>])>,
}

/// forward.rs:107-120
pub struct ForwardRoutineBindingData<'node, M> {
    pub whole_frame_uniform_bg: DataHandle<BindGroup>,
    pub per_material_bgl: &'node PerMaterialArchetypeInterface<M>,
    pub extra_bgs: Option<&'node [BindGroup]>,
}

/// forward.rs:122-133
pub struct ForwardRoutineArgs<'a, 'node, M> {
    pub graph: &'a mut RenderGraph<'node>,

    pub label: &'a str,

    pub camera: CameraSpecifier,
    pub binding_data: ForwardRoutineBindingData<'node, M>,

    pub culling_source: CullingSource,
    pub samples: SampleCount,
    pub renderpass: RenderPassTargets,
}

pub struct ForwardRoutine<M: Material> {
    amd: Arc<AmdContext>,
    routine_type: RoutineType,
    /// `Material::key()` of the archetype this routine draws (pbr/material.rs:383-392: TransparencyType as u64)
    material_key: u64,
    _phantom: PhantomData<M>,
}

impl<M: Material> ForwardRoutine<M> {
    /// forward.rs:159-190 builds the render pipelines (cull mode, depth compare GreaterEqual, blend state); that fixed-function
    /// state is what the library's rasteriser implements.
    pub fn new(args: ForwardRoutineCreateArgs<'_, M>) -> Self {
        Self { amd: AmdContext::of(args.renderer), routine_type: args.routine_type, material_key: args.material_key, _phantom: PhantomData }
    }

    /// forward.rs:192-315: per material-key region one `draw_indexed_indirect` (or nothing when the region is empty, :285-288).
    pub fn add_forward_to_graph<'node>(&'node self, args: ForwardRoutineArgs<'_, 'node, M>) {
        let mut builder = args.graph.add_node(args.label);
        builder.add_side_effect();
        let pass = self.routine_type as u32;
        let source = match args.culling_source {
            CullingSource::Predicted => sys::R3N_SOURCE_PREDICTED,
            CullingSource::Residual(_) => sys::R3N_SOURCE_RESIDUAL,
        };
        let camera = args.camera.to_shader_index();
        let key = self.material_key as u32;
        builder.build(move |_ctx| {
            self.amd.check(unsafe { sys::r3n_forward(self.amd.ctx, camera, pass, source, key) }, "r3n_forward");
        });
    }
}
