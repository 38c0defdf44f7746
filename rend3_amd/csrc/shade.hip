// shade.hip -- translation unit of the resolve / blend / tonemap kernels (kernels_shade.h) and their launchers.
// Split from r3n.hip so that the largest kernels of the library compile in parallel with the rest.
#include <hip/hip_runtime.h>

#include "kernels_shade.h"

namespace {
int launch_status() { return (int)hipGetLastError(); }
}  // namespace

extern "C" int r3n_internal_resolve_ms(const ShadeArgs *ap, int tex, int rec, int split, hipStream_t stream);

extern "C" {

int r3n_internal_build_srgb_lut(unsigned char *lut, hipStream_t stream) {
    hipLaunchKernelGGL(k_build_srgb_lut, dim3((R3N_SRGB_LUT_SIZE + 255u) / 256u), dim3(256), 0, stream, lut);
    return launch_status();
}

int r3n_internal_shade_prepass(const ShadeArgs *ap, int tex, size_t first_key, size_t n_keys, hipStream_t stream) {
    const ShadeArgs &a = *ap;
    // a.seen is all zero here: allocated zeroed, and k_vertex_stage clears every flag it consumes
    hipLaunchKernelGGL(k_mark_visible, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, stream, a.vis, a.seen, first_key, n_keys);
    const dim3 vgrid((unsigned)(((size_t)a.total_tris + 255) / 256));
    if (tex) hipLaunchKernelGGL(k_vertex_stage<true>, vgrid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(k_vertex_stage<false>, vgrid, dim3(256), 0, stream, a);
    return launch_status();
}

int r3n_internal_resolve(const ShadeArgs *ap, uint32_t samples, int tex, int rec, int split, int fast, hipStream_t stream) {
    if (samples == 4) return r3n_internal_resolve_ms(ap, tex, rec, split, stream);  // shade_ms.hip
    const ShadeArgs &a = *ap;
    const dim3 rgrid((a.width + 15u) / 16u, (a.row_end - a.row_begin + 15u) / 16u);
    if (rec) {  // the record-based single-sample kernels read the lights from a.view_lights
        if (!a.view_lights) return (int)hipErrorInvalidValue;
        hipLaunchKernelGGL(k_stage_view_lights, dim3(1), dim3(256), 0, stream, a, const_cast<ViewLights *>(a.view_lights));
    }
    if (rec && tex && a.variants != 0u) {
        // one launch per material class in flight (kernels_shade.h R3N_CLS_*); the general kernel is the chain's top
        for (uint32_t v = 0; v < R3N_VARIANTS; ++v) {
            if (!((a.variants >> v) & 1u)) continue;
            if (v < R3N_VARIANTS - 1u) {
                const int rc = r3n_internal_resolve_class(ap, v, fast, stream);
                if (rc) return rc;
            } else if (fast) {
                hipLaunchKernelGGL((k_resolve_opaque<1, true, true, false, true>), rgrid, dim3(256), a.resolve_lds, stream, a);
            } else {
                hipLaunchKernelGGL((k_resolve_opaque<1, true, true>), rgrid, dim3(256), a.resolve_lds, stream, a);
            }
        }
    } else if (rec && fast) {
        if (tex) hipLaunchKernelGGL((k_resolve_opaque<1, true, true, false, true>), rgrid, dim3(256), a.resolve_lds, stream, a);
        else hipLaunchKernelGGL((k_resolve_opaque<1, false, true, false, true>), rgrid, dim3(256), a.resolve_lds, stream, a);
    } else if (rec) {
        if (tex) hipLaunchKernelGGL((k_resolve_opaque<1, true, true>), rgrid, dim3(256), a.resolve_lds, stream, a);
        else hipLaunchKernelGGL((k_resolve_opaque<1, false, true>), rgrid, dim3(256), a.resolve_lds, stream, a);
    } else {
        if (tex) hipLaunchKernelGGL((k_resolve_opaque<1, true, false>), rgrid, dim3(256), a.resolve_lds, stream, a);
        else hipLaunchKernelGGL((k_resolve_opaque<1, false, false>), rgrid, dim3(256), a.resolve_lds, stream, a);
    }
    return launch_status();
}

int r3n_internal_tonemap(const ushort4 *hdr, uchar4 *out, float4 *out_f32, size_t first_pixel, size_t n_pixels, const unsigned char *srgb_lut,
                         uint32_t output_format, hipStream_t stream) {
    const size_t pairs = (n_pixels + 1) / 2;
    hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, stream, hdr, out, out_f32, first_pixel, n_pixels,
                       srgb_lut, output_format);
    return launch_status();
}

}  // extern "C"
