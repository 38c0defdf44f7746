// shade_ms.hip -- the multisampled (SampleCount::Four) instantiations of the resolve, their own translation unit (build time).
#include <hip/hip_runtime.h>

#include "kernels_shade.h"

extern "C" int r3n_internal_resolve_ms(const ShadeArgs *ap, int tex, int rec, int split, hipStream_t stream) {
    const ShadeArgs &a = *ap;
    const dim3 rgrid((a.width + 15u) / 16u, (a.row_end - a.row_begin + 15u) / 16u);
    if (split) {
        // split resolve: first triangle of every pixel, then the extra triangles of edge pixels in a dense second pass,
        // then the edge pixels' box average
        const dim3 egrid(R3N_EDGEQ * 64u);
        if (tex) {
            hipLaunchKernelGGL((k_resolve_opaque<4, true, true, true>), rgrid, dim3(256), 0, stream, a);
            hipLaunchKernelGGL((k_resolve_edges<true, true>), egrid, dim3(256), 0, stream, a);
        } else {
            hipLaunchKernelGGL((k_resolve_opaque<4, false, true, true>), rgrid, dim3(256), 0, stream, a);
            hipLaunchKernelGGL((k_resolve_edges<false, true>), egrid, dim3(256), 0, stream, a);
        }
        hipLaunchKernelGGL(k_resolve_edge_pixels, egrid, dim3(256), 0, stream, a);
    } else if (rec) {
        if (tex) hipLaunchKernelGGL((k_resolve_opaque<4, true, true>), rgrid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_resolve_opaque<4, false, true>), rgrid, dim3(256), 0, stream, a);
    } else {
        if (tex) hipLaunchKernelGGL((k_resolve_opaque<4, true>), rgrid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((k_resolve_opaque<4, false>), rgrid, dim3(256), 0, stream, a);
    }
    return (int)hipGetLastError();
}
