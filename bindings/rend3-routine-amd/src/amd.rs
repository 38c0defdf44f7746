//! The `r3n_ctx` owner lives in rend3 itself: `bindings/rend3-hooks.patch` adds `rend3::util::amd::AmdContext` (the context, the
//! error convention and the world's mirror the managers feed) and makes `Renderer::new` create it (`Renderer::amd`).  The node
//! bodies of this crate reach it through `AmdContext::of(renderer)` / `AmdContext::of_ref(renderer)`; nothing here owns state.
pub use rend3::util::amd::{texture_format_id, AmdContext};
