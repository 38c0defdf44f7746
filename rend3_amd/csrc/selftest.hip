// selftest.hip -- exhaustive check of exact_math.h on the device: every f32 bit pattern through the short sequences and through
// the compiler's correctly rounded expansions.  r3n_selftest_exact_math fills, per function, a histogram of DIFFERING patterns by
// sign + biased exponent (512 bins) for the unguarded cores -- what the guards in exact_math.h were chosen from -- and counts
// the differences of the guarded functions (must be zero: tests/test_exact_math.py).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "exact_math.h"

namespace {
__device__ __forceinline__ bool same(float a, float b) {
    return __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);  // any NaN equals any NaN (payloads are not part of the contract)
}
// hist: [fn 0 rcp core | 1 sqrt core | 2 rsqrt core][512]; guarded: [3] differences of rcp / sqrt / rsqrt as the library uses them
__global__ __launch_bounds__(256) void k_exact_probe(unsigned long long *__restrict__ hist, unsigned long long *__restrict__ guarded) {
    __shared__ unsigned int s_hist[3][512];
    for (uint32_t i = threadIdx.x; i < 3u * 512u; i += 256u) (&s_hist[0][0])[i] = 0u;
    __syncthreads();
    unsigned int bad[3] = {0u, 0u, 0u};
    // block b covers the patterns [b << 20, (b + 1) << 20): one exponent bin per block at most two
    const uint32_t base = blockIdx.x << 20;
    for (uint32_t k = threadIdx.x; k < (1u << 20); k += 256u) {
        const uint32_t bits = base + k;
        const float x = __uint_as_float(bits);
        const uint32_t bin = bits >> 23;
        const float r_ref = 1.0f / x, s_ref = sqrtf(x), q_ref = 1.0f / sqrtf(x);
        if (!same(exact_math::rcp_core(x), r_ref)) atomicAdd(&s_hist[0][bin], 1u);
        if (!same(exact_math::sqrt_core(x), s_ref)) atomicAdd(&s_hist[1][bin], 1u);
        if (!same(exact_math::rcp_core(exact_math::sqrt_core(x)), q_ref)) atomicAdd(&s_hist[2][bin], 1u);
        bad[0] += same(exact_math::rcp(x), r_ref) ? 0u : 1u;
        bad[1] += same(exact_math::sqrt(x), s_ref) ? 0u : 1u;
        bad[2] += same(exact_math::rsqrt(x), q_ref) ? 0u : 1u;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 3u * 512u; i += 256u) {
        const unsigned int v = (&s_hist[0][0])[i];
        if (v) atomicAdd(&hist[i], (unsigned long long)v);
    }
    for (int f = 0; f < 3; ++f)
        if (bad[f]) atomicAdd(&guarded[f], (unsigned long long)bad[f]);
}
}  // namespace

namespace {
__global__ void k_unorm8_probe(uint32_t *bad) {
    const uint32_t c = threadIdx.x;  // 256 threads: every 8-bit value
    const float ref = (float)c / 255.0f;
    if (__float_as_uint(exact_math::unorm8(c)) != __float_as_uint(ref)) atomicAdd(bad, 1u);
}
}  // namespace

// exact_math::unorm8 against the compiler's division for all 256 inputs; *n_bad = how many differ (must be 0).
extern "C" int r3n_selftest_unorm8(int device, uint32_t *n_bad) {
    if (!n_bad) return (int)hipErrorInvalidValue;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    uint32_t *d = nullptr;
    if ((e = hipMalloc(&d, 4)) != hipSuccess) return (int)e;
    e = hipMemset(d, 0, 4);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_unorm8_probe, dim3(1), dim3(256), 0, 0, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(n_bad, d, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return (int)e;
}

// hist: 3 x 512 counters, guarded: 3 counters (host memory).  Returns a hipError_t as int.  Stand-alone: needs no context.
extern "C" int r3n_selftest_exact_math(int device, unsigned long long *hist, unsigned long long *guarded) {
    if (!hist || !guarded) return (int)hipErrorInvalidValue;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return (int)e;
    unsigned long long *d = nullptr;
    const size_t n = 3u * 512u + 3u;
    if ((e = hipMalloc(&d, n * 8)) != hipSuccess) return (int)e;
    e = hipMemset(d, 0, n * 8);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_exact_probe, dim3(4096), dim3(256), 0, 0, d, d + 3u * 512u);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned long long host[3u * 512u + 3u];
    if (e == hipSuccess) e = hipMemcpy(host, d, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return (int)e;
    for (size_t i = 0; i < 3u * 512u; ++i) hist[i] = host[i];
    for (size_t i = 0; i < 3; ++i) guarded[i] = host[3u * 512u + i];
    return 0;
}
