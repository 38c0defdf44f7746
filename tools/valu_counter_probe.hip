// valu_counter_probe.hip -- calibration of the SQ_INSTS_VALU_* counters bench.py's "useful f32 flops" are made of.
//
// Each kernel below is ONE wavefront executing a known number of ONE vector instruction (inline asm, so the compiler can neither
// fuse, pack nor drop them).  Run under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32
// SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32` the per-kernel counter values divided by the instruction
// count give each counter's weight for that instruction: whether a packed `v_pk_mul_f32` (two multiplications per lane) counts once
// or twice in MUL_F32 / FLOPS_FP32, what an fma weighs, whether transcendentals are in FLOPS_FP32.  tools/make_traffic.py's
// formula follows the table this prints (profiles/r05_valu_counter_probe.txt).
//
//   hipcc --offload-arch=gfx950 -O2 tools/valu_counter_probe.hip -o tests/_build/valu_counter_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kUnroll = 16;

#define PROBE1(name, text)                                                          \
    __global__ void probe_##name(float *out, int iters) {                           \
        float a = 1.0f + 1e-3f * (float)threadIdx.x, b = 1.0001f;                    \
        for (int i = 0; i < iters; ++i) {                                           \
            _Pragma("unroll") for (int u = 0; u < kUnroll; ++u) asm volatile(text : "+v"(a) : "v"(b)); \
        }                                                                           \
        out[threadIdx.x] = a;                                                       \
    }
#define PROBE2(name, text)                                                          \
    __global__ void probe_##name(float *out, int iters) {                           \
        f2 a = {1.0f + 1e-3f * (float)threadIdx.x, 2.0f}, b = {1.0001f, 0.9999f};    \
        for (int i = 0; i < iters; ++i) {                                           \
            _Pragma("unroll") for (int u = 0; u < kUnroll; ++u) asm volatile(text : "+v"(a) : "v"(b)); \
        }                                                                           \
        out[threadIdx.x] = a.x + a.y;                                               \
    }

PROBE1(v_mul_f32, "v_mul_f32 %0, %0, %1")
PROBE1(v_add_f32, "v_add_f32 %0, %0, %1")
PROBE1(v_sub_f32, "v_sub_f32 %0, %0, %1")
PROBE1(v_fma_f32, "v_fma_f32 %0, %0, %1, %1")
PROBE1(v_fmac_f32, "v_fmac_f32 %0, %1, %1")
PROBE1(v_max_f32, "v_max_f32 %0, %0, %1")
PROBE1(v_rcp_f32, "v_rcp_f32 %0, %1")
PROBE1(v_sqrt_f32, "v_sqrt_f32 %0, %1")
PROBE1(v_floor_f32, "v_floor_f32 %0, %1")
PROBE1(v_mov_b32, "v_mov_b32 %0, %1")
PROBE1(v_cvt_f32_u32, "v_cvt_f32_u32 %0, %1")
PROBE1(v_div_fixup_f32, "v_div_fixup_f32 %0, %0, %1, %1")
PROBE2(v_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
PROBE2(v_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
PROBE2(v_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %1")

int main() {
    float *out = nullptr;
    if (hipMalloc(&out, 64 * sizeof(float)) != hipSuccess) { std::printf("no device\n"); return 1; }
    const int iters = 4096;  // x 16 = 65 536 wave-instructions per kernel
#define RUN(name) hipLaunchKernelGGL(probe_##name, dim3(1), dim3(64), 0, 0, out, iters)
    RUN(v_mul_f32); RUN(v_add_f32); RUN(v_sub_f32); RUN(v_fma_f32); RUN(v_fmac_f32); RUN(v_max_f32); RUN(v_rcp_f32); RUN(v_sqrt_f32);
    RUN(v_floor_f32); RUN(v_mov_b32); RUN(v_cvt_f32_u32); RUN(v_div_fixup_f32); RUN(v_pk_mul_f32); RUN(v_pk_add_f32); RUN(v_pk_fma_f32);
    if (hipDeviceSynchronize() != hipSuccess) { std::printf("launch failed\n"); return 1; }
    std::printf("instructions per kernel: %d\n", iters * kUnroll);
    (void)hipFree(out);
    return 0;
}
