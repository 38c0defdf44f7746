//! rend3-routine/src/tonemapping.rs:79-148
use crate::amd::AmdContext;
use rend3::graph::{DataHandle, RenderGraph, RenderTargetHandle};
use rend3::types::TextureFormat;
use rend3::{Renderer, ShaderPreProcessor};
use rend3_amd_sys as sys;
use rend3_routine::common::WholeFrameInterfaces;
use std::sync::Arc;
use wgpu::BindGroup;

pub struct TonemappingRoutine {
    amd: Arc<AmdContext>,
}

impl TonemappingRoutine {
    /// tonemapping.rs:85-106 -- same signature: `fs_main_scene` for *Srgb targets (the store applies the exact OETF),
    /// `fs_main_monitor` (srgb_scene_to_display, exponent 0.4166) for the others; Rgba / Bgra byte order.
    pub fn new(
        renderer: &Renderer,
        _spp: &ShaderPreProcessor,
        _interfaces: &WholeFrameInterfaces,
        output_format: TextureFormat,
    ) -> Self {
        let amd = AmdContext::of_ref(renderer);
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

    /// tonemapping.rs:108-147 -- same signature.  `src` is the frame's HDR target, which the context owns; `dst` is the surface
    /// texture: the presentation layer copies `r3n_output_buffer` into it (or maps it), tests read it back with
    /// `r3n_readback_output`.
    pub fn add_to_graph<'node>(
        &'node self,
        graph: &mut RenderGraph<'node>,
        _src: RenderTargetHandle,
        _dst: RenderTargetHandle,
        _forward_uniform_bg: DataHandle<BindGroup>,
    ) {
        let mut builder = graph.add_node("Tonemapping");
        builder.add_side_effect();
        builder.build(move |_ctx| {
            self.amd.check(unsafe { sys::r3n_tonemap(self.amd.ctx, std::ptr::null_mut(), 0) }, "r3n_tonemap");
        });
    }
}
