//! rend3-routine/src/hi_z.rs:18-235
use crate::amd::AmdContext;
use crate::base::DepthTargets;
use glam::UVec2;
use rend3::graph::RenderGraph;
use rend3::{Renderer, ShaderPreProcessor};
use rend3_amd_sys as sys;
use std::sync::Arc;

pub struct HiZRoutine {
    amd: Arc<AmdContext>,
}

impl HiZRoutine {
    /// hi_z.rs:29 -- same signature (a plain `&Renderer` there: the context is looked up, `BaseRenderGraph::new` created it).
    pub fn new(renderer: &Renderer, _spp: &ShaderPreProcessor) -> Self {
        Self { amd: AmdContext::of_ref(renderer) }
    }

    /// hi_z.rs:161-234: one raster pass per mip (`hi_z.wgsl:19-32`, and `resolve_depth_min.wgsl` first under MSAA); here two
    /// launches build the whole pyramid from the pass-1 depth the context holds (`depth_targets` / `resolution` were given to
    /// `r3n_frame_begin`).
    pub fn add_hi_z_to_graph<'node>(
        &'node self,
        graph: &mut RenderGraph<'node>,
        _depth_targets: DepthTargets,
        _resolution: UVec2,
    ) {
        let mut node = graph.add_node("HiZ");
        node.add_side_effect();
        node.build(move |_ctx| {
            self.amd.check(unsafe { sys::r3n_hi_z(self.amd.ctx) }, "r3n_hi_z");
        });
    }
}
