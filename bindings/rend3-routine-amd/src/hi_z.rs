//! rend3-routine/src/hi_z.rs:18-235
use crate::amd::AmdContext;
use rend3::graph::RenderGraph;
use rend3_amd_sys as sys;

pub struct HiZRoutine<'a> {
    pub amd: &'a AmdContext,
}

impl<'a> HiZRoutine<'a> {
    /// hi_z.rs:161-234: one raster pass per mip (`hi_z.wgsl:19-32`, and `resolve_depth_min.wgsl` first under MSAA); here two
    /// launches build the whole pyramid from the pass-1 depth.
    pub fn add_hi_z_to_graph<'node>(&'node self, graph: &mut RenderGraph<'node>) {
        let mut node = graph.add_node("HiZ");
        node.add_side_effect();
        node.build(move |_ctx| {
            self.amd.check(unsafe { sys::r3n_hi_z(self.amd.ctx) }, "r3n_hi_z");
        });
    }
}
