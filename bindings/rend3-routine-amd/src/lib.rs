//! Node bodies of rend3-routine's hot path over the C ABI of `librend3_amd.so`.
//!
//! Every public item keeps the name and signature of the rend3-routine item it replaces (file:line cited on each;
//! tests/test_rust_bindings.py diffs the constructor and `add_*_to_graph` signatures against the reference's, pinned in
//! tests/golden/rust_signatures.json): `BaseRenderGraph::new(&renderer, &spp)`, `PbrRoutine::new(...)`,
//! `TonemappingRoutine::new(...)` are called exactly as in the reference's examples -- the `AmdContext` is found from the
//! renderer (`AmdContext::of` = `Renderer::amd`: bindings/rend3-hooks.patch makes `Renderer::new` create it and the managers feed it).  Only the closures registered with the render
//! graph differ: where the reference records wgpu passes, these call `r3n_*`.  The graph
//! machinery, the managers and the user-facing `Renderer` API stay rend3's.  Source only in this repository (no Rust toolchain
//! in the build image): the same call sequence runs from Python in `rend3_amd/renderer.py`, which the GPU tests drive.
//!
//! rend3 has no FFI of its own; the C ABI (`include/r3n.h`) is the boundary its node bodies call instead of wgpu.
pub mod amd;
pub mod base;
pub mod culler;
pub mod forward;
pub mod hi_z;
pub mod skinning;
pub mod tonemapping;
pub mod uniforms;

pub use amd::AmdContext;
