//! rend3-routine/src/tonemapping.rs:29-148
use crate::amd::AmdContext;
use rend3::graph::RenderGraph;
use rend3::types::TextureFormat;
use rend3_amd_sys as sys;

pub struct TonemappingRoutine<'a> {
    pub amd: &'a AmdContext,
}

impl<'a> TonemappingRoutine<'a> {
    /// tonemapping.rs:29-106: `fs_main_scene` for *Srgb targets (the store applies the exact OETF), `fs_main_monitor`
    /// (srgb_scene_to_display, exponent 0.4166) for the others; Rgba / Bgra byte order.
    pub fn new(amd: &'a AmdContext, output_format: TextureFormat) -> Self {
        let format = match output_format {
            TextureFormat::Rgba8UnormSrgb => sys::R3N_OUTPUT_RGBA8_UNORM_SRGB,
            TextureFormat::Bgra8UnormSrgb => sys::R3N_OUTPUT_BGRA8_UNORM_SRGB,
            TextureFormat::Rgba8Unorm => sys::R3N_OUTPUT_RGBA8_UNORM,
            TextureFormat::Bgra8Unorm => sys::R3N_OUTPUT_BGRA8_UNORM,
            other => panic!("unsupported surface format {other:?}"),
        };
        amd.check(unsafe { sys::r3n_set_output_format(amd.ctx, format) }, "r3n_set_output_format");
        Self { amd }
    }

    /// tonemapping.rs:108-147.  `readback`: where the host wants the frame (rend3-test/src/runner.rs:189-225 reads the target
    /// back); `None` leaves it on the device (`r3n_output_buffer` shares it with the presentation layer).
    pub fn add_to_graph<'node>(&'node self, graph: &mut RenderGraph<'node>, readback: Option<(&'node mut [u8], u64)>) {
        let mut builder = graph.add_node("Tonemapping");
        builder.add_side_effect();
        builder.build(move |_ctx| {
            let (ptr, pitch) = match readback {
                Some((buf, pitch)) => (buf.as_mut_ptr().cast(), pitch),
                None => (std::ptr::null_mut(), 0),
            };
            self.amd.check(unsafe { sys::r3n_tonemap(self.amd.ctx, ptr, pitch) }, "r3n_tonemap");
        });
    }
}
