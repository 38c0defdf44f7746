// shade_blend.hip -- the transparent pass's ordered blend (row N3), its own translation unit (build time).
#include <hip/hip_runtime.h>

#include "kernels_shade.h"

extern "C" {

int r3n_internal_blend_apply(const ShadeArgs *ap, const BlendApplyArgs *bp, uint32_t samples, int tex, hipStream_t stream) {
    const ShadeArgs &sa = *ap;
    const BlendApplyArgs &ba = *bp;
    const dim3 g((ba.n_samples + 255u) / 256u);
    if (samples == 4) {
        if (tex) hipLaunchKernelGGL((k_blend_apply<4, true>), g, dim3(256), 0, stream, sa, ba);
        else hipLaunchKernelGGL((k_blend_apply<4, false>), g, dim3(256), 0, stream, sa, ba);
    } else {
        if (tex) hipLaunchKernelGGL((k_blend_apply<1, true>), g, dim3(256), 0, stream, sa, ba);
        else hipLaunchKernelGGL((k_blend_apply<1, false>), g, dim3(256), 0, stream, sa, ba);
    }
    return (int)hipGetLastError();
}

int r3n_internal_resolve_samples(const ushort4 *samples, ushort4 *hdr_out, size_t first_pixel, size_t n_pixels, hipStream_t stream) {
    hipLaunchKernelGGL(k_resolve_samples, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream, samples, hdr_out, first_pixel, n_pixels);
    return (int)hipGetLastError();
}

}  // extern "C"
