//! rend3-routine/src/uniforms.rs:73-125 (`add_to_graph`) + clear.rs:4-20 (`add_depth_clear_to_graph`) in one node.
use crate::amd::AmdContext;
use glam::{UVec2, Vec4};
use rend3::graph::RenderGraph;
use rend3::types::SampleCount;
use rend3_amd_sys as sys;
use rend3_routine::uniforms::FrameUniforms;

pub struct UniformInformation {
    pub ambient: Vec4,
    pub resolution: UVec2,
    pub samples: SampleCount,
    pub clear_color: Vec4,
}

/// Uploads the 496-byte `FrameUniforms` (built exactly as uniforms.rs:41-56 does), clears the frame's targets (depth 0.0 /
/// colour, base.rs:245-263) and the shadow atlas (0.0), and places the shadow viewports (`ShadowDesc::map`).
pub fn add_to_graph<'node>(graph: &mut RenderGraph<'node>, amd: &'node std::sync::Arc<AmdContext>, info: UniformInformation) {
    let mut builder = graph.add_node("build uniform data");
    builder.add_side_effect();
    builder.build(move |ctx| {
        let uniforms = FrameUniforms::new(&ctx.data_core.viewport_camera_state, &info_as_reference(&info));
        let mut bytes = [0u8; 496];
        encase::UniformBuffer::new(&mut bytes[..]).write(&uniforms).unwrap();
        let size = ctx.eval_output.shadow_target_size;
        let clear = info.clear_color.to_array();
        amd.check(
            unsafe {
                sys::r3n_frame_begin(amd.ctx, bytes.as_ptr().cast(), info.resolution.x, info.resolution.y, info.samples as u32, clear.as_ptr(), size.x, size.y)
            },
            "r3n_frame_begin",
        );
        for (i, shadow) in ctx.eval_output.shadows.iter().enumerate() {
            amd.check(unsafe { sys::r3n_shadow_viewport(amd.ctx, i as u32, shadow.map.offset.x, shadow.map.offset.y, shadow.map.size) }, "r3n_shadow_viewport");
        }
    });
}

fn info_as_reference(info: &UniformInformation) -> rend3_routine::uniforms::UniformInformation<'static> {
    // samplers are not read by FrameUniforms::new; the reference struct wants the field
    rend3_routine::uniforms::UniformInformation { samplers: rend3_routine::common::Samplers::placeholder(), ambient: info.ambient, resolution: info.resolution }
}
