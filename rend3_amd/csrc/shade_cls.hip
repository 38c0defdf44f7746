// shade_cls.hip -- the material-class instantiations of the single-sample record-based resolve (kernels_shade.h, R3N_CLS_*),
// their own translation unit (build time): textured worlds, variants PLAIN / ALBEDO / PBR3.  The general kernel (variant ALL)
// stays in shade.hip.
#include <hip/hip_runtime.h>

#include "kernels_shade.h"

extern "C" int r3n_internal_resolve_class(const ShadeArgs *ap, uint32_t variant, int fast, hipStream_t stream) {
    const ShadeArgs &a = *ap;
    const dim3 rgrid((a.width + 15u) / 16u, (a.row_end - a.row_begin + 15u) / 16u);
    switch (variant) {
    case 0u:
        if (fast) hipLaunchKernelGGL((k_resolve_opaque<1, true, true, false, true, R3N_CLS_PLAIN, 0u>), rgrid, dim3(256), a.resolve_lds, stream, a);
        else hipLaunchKernelGGL((k_resolve_opaque<1, true, true, false, false, R3N_CLS_PLAIN, 0u>), rgrid, dim3(256), a.resolve_lds, stream, a);
        break;
    case 1u:
        if (fast) hipLaunchKernelGGL((k_resolve_opaque<1, true, true, false, true, R3N_CLS_ALBEDO, 1u>), rgrid, dim3(256), a.resolve_lds, stream, a);
        else hipLaunchKernelGGL((k_resolve_opaque<1, true, true, false, false, R3N_CLS_ALBEDO, 1u>), rgrid, dim3(256), a.resolve_lds, stream, a);
        break;
    case 2u:
        if (fast) hipLaunchKernelGGL((k_resolve_opaque<1, true, true, false, true, R3N_CLS_PBR3, 2u>), rgrid, dim3(256), a.resolve_lds, stream, a);
        else hipLaunchKernelGGL((k_resolve_opaque<1, true, true, false, false, R3N_CLS_PBR3, 2u>), rgrid, dim3(256), a.resolve_lds, stream, a);
        break;
    default:
        return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}
