//! rend3-routine/src/skinning.rs:54-226: `build_gpu_skinning_input_buffers` + `add_skinning_to_graph`.
use crate::amd::AmdContext;
use rend3::graph::RenderGraph;
use rend3::{Renderer, ShaderPreProcessor};
use rend3_amd_sys as sys;
use std::sync::Arc;

pub struct GpuSkinner {
    amd: Arc<AmdContext>,
}

impl GpuSkinner {
    /// skinning.rs:150 takes `&wgpu::Device`; the context hangs off the renderer, so this one takes the renderer
    /// (`BaseRenderGraph::new` is its only caller, base.rs:121).
    pub fn new(renderer: &Arc<Renderer>, _spp: &ShaderPreProcessor) -> GpuSkinner {
        GpuSkinner { amd: AmdContext::of(renderer) }
    }
}

/// skinning.rs:211-226 -- same signature: nothing to do without skeletons (:216-218); otherwise the 40-byte `GpuSkinningInput`
/// records and the joint matrices exactly as skinning.rs:54-139 collects them -- ONE launch covers every skeleton (the
/// reference dispatches per skeleton with a dynamic offset, :181-198).
pub fn add_skinning_to_graph<'node>(graph: &mut RenderGraph<'node>, gpu_skinner: &'node GpuSkinner) {
    let mut node = graph.add_node("skinning");
    node.add_side_effect();
    node.build(move |ctx| {
        let (inputs, matrices) = collect_skinning_inputs(&ctx.data_core.skeleton_manager);
        if inputs.is_empty() {
            return;
        }
        let amd = &gpu_skinner.amd;
        amd.check(
            unsafe { sys::r3n_skinning(amd.ctx, inputs.as_ptr(), inputs.len() as u32, matrices.as_ptr().cast(), (matrices.len() / 16) as u32) },
            "r3n_skinning",
        );
    });
}

/// skinning.rs:54-139 without the buffer creation: per skeleton the source ranges of its mesh's attributes, its private output
/// ranges and the base index of its joint matrices in the flat matrix array -- from `InternalSkeleton`'s public fields
/// (rend3/src/managers/skeleton.rs:19-33), exactly the loop of skinning.rs:82-132.
fn collect_skinning_inputs(skeletons: &rend3::managers::SkeletonManager) -> (Vec<sys::r3n_skinning_input40>, Vec<f32>) {
    use rend3::types::{
        VERTEX_ATTRIBUTE_JOINT_INDICES, VERTEX_ATTRIBUTE_JOINT_WEIGHTS, VERTEX_ATTRIBUTE_NORMAL, VERTEX_ATTRIBUTE_POSITION, VERTEX_ATTRIBUTE_TANGENT,
    };
    let mut inputs = Vec::new();
    let mut matrices = Vec::new();
    for skeleton in skeletons.skeletons() {
        let mut input = sys::r3n_skinning_input40 {
            base_position_offset: u32::MAX,
            base_normal_offset: u32::MAX,
            base_tangent_offset: u32::MAX,
            joint_indices_offset: u32::MAX,
            joint_weight_offset: u32::MAX,
            updated_position_offset: u32::MAX,
            updated_normal_offset: u32::MAX,
            updated_tangent_offset: u32::MAX,
            joint_matrix_base_offset: (matrices.len() / 16) as u32,
            vertex_count: skeleton.vertex_count,
        };
        for (attribute, range) in &skeleton.source_attribute_ranges {
            match attribute {
                a if *a == *VERTEX_ATTRIBUTE_POSITION => input.base_position_offset = range.start as u32,
                a if *a == *VERTEX_ATTRIBUTE_NORMAL => input.base_normal_offset = range.start as u32,
                a if *a == *VERTEX_ATTRIBUTE_TANGENT => input.base_tangent_offset = range.start as u32,
                a if *a == *VERTEX_ATTRIBUTE_JOINT_INDICES => input.joint_indices_offset = range.start as u32,
                a if *a == *VERTEX_ATTRIBUTE_JOINT_WEIGHTS => input.joint_weight_offset = range.start as u32,
                a => unreachable!("Unknown skinning input attribute {a:?}"),
            }
        }
        for (attribute, range) in &skeleton.overridden_attribute_ranges {
            match attribute {
                a if *a == *VERTEX_ATTRIBUTE_POSITION => input.updated_position_offset = range.start as u32,
                a if *a == *VERTEX_ATTRIBUTE_NORMAL => input.updated_normal_offset = range.start as u32,
                a if *a == *VERTEX_ATTRIBUTE_TANGENT => input.updated_tangent_offset = range.start as u32,
                a => unreachable!("Unknown skinning output attribute {a:?}"),
            }
        }
        inputs.push(input);
        for m in &skeleton.joint_matrices {
            matrices.extend_from_slice(&m.to_cols_array());
        }
    }
    (inputs, matrices)
}
