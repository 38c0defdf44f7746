//! rend3-routine/src/forward.rs:135-316 -- one (routine type, material key) pipeline; `add_forward_to_graph` keeps its
//! `ForwardRoutineArgs` shape (forward.rs:72-108) minus the wgpu-only members.
use crate::amd::AmdContext;
use rend3::graph::RenderGraph;
use rend3::types::{Material, SampleCount};
use rend3_amd_sys as sys;
use rend3_routine::common::CameraSpecifier;
use std::marker::PhantomData;

/// forward.rs:40-44
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum RoutineType {
    Depth = sys::R3N_PASS_DEPTH as isize,
    Forward = sys::R3N_PASS_FORWARD as isize,
}

/// forward.rs:64-70; the `DrawCallSet` handle of `Residual` is owned by the context
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum CullingSource {
    Predicted,
    Residual,
}

pub struct ForwardRoutineArgs<'a, 'node> {
    pub graph: &'a mut RenderGraph<'node>,
    pub label: &'a str,
    pub camera: CameraSpecifier,
    pub culling_source: CullingSource,
    pub samples: SampleCount,
}

pub struct ForwardRoutine<'a, M: Material> {
    pub amd: &'a AmdContext,
    pub routine_type: RoutineType,
    /// `Material::key()` of the archetype this routine draws (pbr/material.rs:383-392: TransparencyType as u64)
    pub material_key: u64,
    _phantom: PhantomData<M>,
}

impl<'a, M: Material> ForwardRoutine<'a, M> {
    /// forward.rs:159-190 builds the render pipeline (cull mode, depth compare GreaterEqual, blend state); that fixed-function
    /// state is what the library's rasteriser implements.
    pub fn new(amd: &'a AmdContext, routine_type: RoutineType, material_key: u64) -> Self {
        Self { amd, routine_type, material_key, _phantom: PhantomData }
    }

    /// forward.rs:192-315: per material-key region one `draw_indexed_indirect` (or nothing when the region is empty, :285-288).
    pub fn add_forward_to_graph<'node>(&'node self, args: ForwardRoutineArgs<'_, 'node>) {
        let mut builder = args.graph.add_node(args.label);
        builder.add_side_effect();
        let pass = self.routine_type as u32;
        let source = match args.culling_source {
            CullingSource::Predicted => sys::R3N_SOURCE_PREDICTED,
            CullingSource::Residual => sys::R3N_SOURCE_RESIDUAL,
        };
        let camera = args.camera.to_shader_index();
        let key = self.material_key as u32;
        builder.build(move |_ctx| {
            self.amd.check(unsafe { sys::r3n_forward(self.amd.ctx, camera, pass, source, key) }, "r3n_forward");
        });
    }
}
