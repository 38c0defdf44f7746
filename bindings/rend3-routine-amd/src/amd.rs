//! Owner of the `r3n_ctx` (one HIP device + its streams) and the error convention.
use rend3::Renderer;
use rend3_amd_sys as sys;
use std::collections::HashMap;
use std::ffi::CStr;
use std::sync::{Arc, Mutex, OnceLock};

/// Lives next to `Renderer::data_core` (rend3/src/renderer/mod.rs:54-106); created where `Renderer::new` creates the wgpu
/// device (rend3/src/renderer/setup.rs:20-107), destroyed with the renderer.
pub struct AmdContext {
    pub(crate) ctx: *mut sys::r3n_ctx,
}

// The reference serialises graph execution behind the data_core mutex (rend3/src/graph/graph.rs:265); the C ABI asks for the
// same: one thread at a time.
unsafe impl Send for AmdContext {}
unsafe impl Sync for AmdContext {}

/// One context per `Renderer`, found FROM the renderer: the routine constructors keep the reference's signatures
/// (`BaseRenderGraph::new(&renderer, &spp)`, `PbrRoutine::new(&renderer, ...)`, ...), so no call site hands a context
/// around.  The first constructor that asks creates it (HIP device `R3N_HIP_DEVICE`, default 0; `R3N_SHADE_FAST=1` opts into
/// the fast fragment arithmetic) and parks a handle in the renderer's graph storage (`Renderer::add_graph_data`,
/// rend3/src/renderer/mod.rs:385-393), where rend3 keeps cross-frame routine state; the map below is only the lookup.
static CONTEXTS: OnceLock<Mutex<HashMap<usize, Arc<AmdContext>>>> = OnceLock::new();

impl AmdContext {
    /// The context of `renderer` (created on first use).
    pub fn of(renderer: &Arc<Renderer>) -> Arc<AmdContext> {
        let key = Arc::as_ptr(renderer) as usize;
        let mut map = CONTEXTS.get_or_init(Default::default).lock().unwrap();
        if let Some(ctx) = map.get(&key) {
            return Arc::clone(ctx);
        }
        let device = std::env::var("R3N_HIP_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
        let fast = std::env::var("R3N_SHADE_FAST").map_or(false, |v| v == "1");
        let ctx = Arc::new(AmdContext::new(device, fast).unwrap_or_else(|e| panic!("r3n_create: {e}")));
        // keeps the context alive as long as the renderer's graph storage (dropped with the renderer)
        std::mem::forget(renderer.add_graph_data(Arc::clone(&ctx)));
        map.insert(key, Arc::clone(&ctx));
        ctx
    }

    /// Same lookup from a node body (`NodeExecutionContext::renderer` is a plain reference).
    pub fn of_ref(renderer: &Renderer) -> Arc<AmdContext> {
        let map = CONTEXTS.get_or_init(Default::default).lock().unwrap();
        Arc::clone(map.get(&(renderer as *const Renderer as usize)).expect("no AmdContext for this renderer: construct BaseRenderGraph first"))
    }

    /// `shade_fast`: opt into `R3N_SHADE_FAST` (fused multiply-add / hardware reciprocals in the fragment stage; framebuffer
    /// within 1e-3 after tonemap instead of bit-identical).
    pub fn new(hip_device: i32, shade_fast: bool) -> Result<Self, String> {
        let config = sys::r3n_config {
            struct_size: std::mem::size_of::<sys::r3n_config>() as u32,
            max_big_items: 0,
            shade_mode: if shade_fast { sys::R3N_SHADE_FAST } else { sys::R3N_SHADE_EXACT },
            _pad: 0,
            reserved: [0; 2],
        };
        let ctx = unsafe { sys::r3n_create(hip_device, &config) };
        if ctx.is_null() {
            return Err(unsafe { CStr::from_ptr(sys::r3n_create_error()) }.to_string_lossy().into_owned());
        }
        Ok(Self { ctx })
    }

    pub fn last_error(&self) -> String {
        unsafe { CStr::from_ptr(sys::r3n_last_error(self.ctx)) }.to_string_lossy().into_owned()
    }

    /// Every entry point returns 0 or a negative code and never unwinds.  The reference's node bodies `unwrap` / `panic!` on the
    /// states these codes describe (culler.rs:439,572), so the adaptor does the same, with the library's message.
    #[track_caller]
    pub fn check(&self, code: i32, what: &str) {
        assert!(code == sys::R3N_OK, "{what}: {} ({code})", self.last_error());
    }

    /// `Renderer::evaluate_instructions` end: nothing to do; `RenderGraph::execute` end (graph.rs:510 `queue.submit`).
    pub fn frame_end(&self) {
        self.check(unsafe { sys::r3n_frame_end(self.ctx) }, "r3n_frame_end");
    }

    /// Blocks until the device is idle (tests, screenshots).
    pub fn sync(&self) {
        self.check(unsafe { sys::r3n_sync(self.ctx) }, "r3n_sync");
    }
}

impl Drop for AmdContext {
    fn drop(&mut self) {
        unsafe { sys::r3n_destroy(self.ctx) }
    }
}
