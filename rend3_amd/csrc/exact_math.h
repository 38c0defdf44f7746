// exact_math.h -- correctly rounded f32 reciprocal / square root / reciprocal square root in fewer instructions than the
// compiler's general expansions, for the arguments a shader actually sees.
//
// The arithmetic contract makes 1 / x, sqrt(x) and 1 / sqrt(x) the correctly rounded IEEE results.  hipcc
// expands them into sequences that also handle subnormal operands and results, infinities and NaN (division: 11 vector
// instructions, square root: 16); the resolve evaluates ~7 normalisations, ~6 reciprocals and ~6 square roots per pixel, a tenth of
// its instruction stream.  Inside a guarded exponent range the scaling and fix-up steps are the identity, and ONE Newton step on
// the hardware estimate (v_rcp_f32 / v_sqrt_f32, 1 ulp) followed by the rounding test is enough.  These are functions of ONE f32:
// `r3n_selftest_exact_math` (selftest.hip) compares every one of the 2^32 bit patterns against the compiler's expansion on the
// device, and the GPU test suite asserts that no pattern inside a function's guard differs -- a proof by exhaustion for the
// hardware the library is built for (gfx950).  Outside the guard the functions take the compiler's expansion.
#pragma once
#include <hip/hip_runtime.h>

namespace exact_math {

// the guard: x is a positive normal number with exponent in [lo, hi) (biased exponent field), as ONE unsigned compare of the bits
template <uint32_t LO, uint32_t HI> __device__ __forceinline__ bool in_range(float x) {
    return (__float_as_uint(x) - (LO << 23)) < ((HI - LO) << 23);
}

// RN(1 / x): y0 = rcp(x) (1 ulp), one Newton step with the residual in an fma.
__device__ __forceinline__ float rcp_core(float x) {
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
// RN(sqrt(x)): s0 = sqrt(x) (1 ulp); the correctly rounded root is s0 or one of its neighbours: the signs of the residuals
// x - s0 * (s0 -+ 1 ulp), exact in an fma, decide (the compiler's own correction step, without the subnormal scaling around it).
__device__ __forceinline__ float sqrt_core(float x) {
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float dn = __uint_as_float(__float_as_uint(s0) - 1u), up = __uint_as_float(__float_as_uint(s0) + 1u);
    const float vp = __builtin_fmaf(-dn, s0, x), vs = __builtin_fmaf(-up, s0, x);
    float s = vp <= 0.0f ? dn : s0;
    s = vs > 0.0f ? up : s;
    return s;
}

// guards (biased exponents), chosen from the exhaustive run's per-exponent mismatch histogram (profiles/r04_exact_math.txt)
// Measured on MI355X (r3n_selftest_exact_math, all 2^32 patterns): rcp_core differs from 1 / x only for biased exponents 0
// (subnormal x) and 253-255 (subnormal quotient, inf, NaN); sqrt_core only below exponent 23 (the residual x - s * s' underflows)
// and never above; their composition only below 22 and at 255.  The guards keep one exponent of margin.
#define R3N_RCP_LO 2u
#define R3N_RCP_HI 252u
#define R3N_SQRT_LO 24u
#define R3N_SQRT_HI 254u

__device__ __forceinline__ float rcp(float x) {  // 1.0f / x
    if (in_range<R3N_RCP_LO, R3N_RCP_HI>(x)) return rcp_core(x);
    return 1.0f / x;
}
__device__ __forceinline__ float sqrt(float x) {  // sqrtf(x)
    if (in_range<R3N_SQRT_LO, R3N_SQRT_HI>(x)) return sqrt_core(x);
    return sqrtf(x);
}
__device__ __forceinline__ float rsqrt(float x) {  // 1.0f / sqrtf(x): two roundings, as the contract writes it
    if (in_range<R3N_SQRT_LO, R3N_SQRT_HI>(x)) return rcp_core(sqrt_core(x));  // the root of a guarded x is far inside rcp's guard
    return 1.0f / sqrtf(x);
}

__device__ __forceinline__ float half_rcp(float x) {  // 0.5f / x: inside rcp's guard the quotient is normal, and halving it is exact
    if (in_range<R3N_RCP_LO, R3N_RCP_HI>(x)) return 0.5f * rcp_core(x);
    return 0.5f / x;
}

// RN((float)c / 255.0f) for an 8-bit c, without the division: q = c * RN(1 / 255) is off by at most an ulp, and one correction with
// the residual c - 255 q (exact in an fma) lands on the correctly rounded quotient for EVERY c in 0 .. 255 -- a function of 256
// inputs, compared one by one with the compiler's division on the device (r3n_selftest_unorm8, tests/test_exact_math.py).
// Three instructions behind the conversion instead of the division's eleven; the rasterisers' cutout test converts eight alphas
// per fragment.
__device__ __forceinline__ float unorm8(uint32_t c) {
    const float x = (float)c, r = 1.0f / 255.0f;
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-255.0f, q, x), r, q);
}

}  // namespace exact_math
