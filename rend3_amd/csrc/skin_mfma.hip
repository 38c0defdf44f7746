// skin_mfma.hip -- the batched 4x4 skin-matrix x vertex-block contraction of BASELINE.json configs[4] on the matrix cores
// (v_mfma_f32_16x16x4_f32), as an opt-in alternative to k_skinning (kernels_cull.h) -- r3n_set_skinning_mode(R3N_SKIN_MFMA).
//
// Reference behaviour restated: skinning.wgsl:37-94 (rend3-routine/src/skinning.rs:54-226 builds its inputs).
//
// Shape.  One MFMA computes D(16 x 16) = A(16 x 4) * B(4 x 16): rows = (joint j, output component r) for up to FOUR joint
// matrices, columns = 16 vertices, K = (x, y, z, 1).  So a rig of at most four joints (the configs[4] rig has two) transforms
// 16 vertices by ALL its joints with one instruction; the blend weights are applied afterwards on the vector ALU, per joint
// slot, and summed across the four 16-lane groups.  A wavefront covers 64 vertices of one skeleton: 4 vertex blocks x
// {position, normal, tangent} = 12 MFMAs.
//
// Arithmetic.  The f32 MFMA is bitwise an fmaf chain over k = 0..3 (MI355X_MICROARCH.md), so this kernel is NOT bit-identical
// to k_skinning, whose contract rounds every multiply and add; it is bit-identical to the FMA-ordered restatement
// oracle/r3o.c::r3o_skinning_mfma_order, which is what tests/test_skinning.py holds it to:
//   q_j   = fma(m_j[.][3], 1, fma(m_j[.][2], z, fma(m_j[.][1], y, fma(m_j[.][0], x, 0))))          (per joint slot j < 4)
//   W_j   = sum over the vertex's influences i (in order) with joint index j and weight > 0 of weight_i
//   p'    = ((q_0 W_0 + q_1 W_1) + q_2 W_2) + q_3 W_3                                                 (unfused)
//   normals / tangents: the same with the columns of m_j scaled by 1 / |column|^2 (folded into the A operand) and w = 0,
//   normalised afterwards (v * (1 / sqrt(v.v)), IEEE) like skinning.wgsl:89-90.
// It is HBM-bound like the vector kernel (96 B per vertex against 384 matrix-core flops): the A/B of the two is what
// profiles/r02_summary.md records.
#include <hip/hip_runtime.h>

#include "device_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// a * b summed over the four 16-lane groups in group order: ((t0 + t1) + t2) + t3, result valid in lanes 0..15
__device__ __forceinline__ float sum_groups(float t, uint32_t lane) {
    const float t1 = __shfl(t, (int)(lane & 15u) + 16, 64), t2 = __shfl(t, (int)(lane & 15u) + 32, 64), t3 = __shfl(t, (int)(lane & 15u) + 48, 64);
    return ((t + t1) + t2) + t3;
}

__global__ __launch_bounds__(256) void k_skinning_mfma(uint32_t *__restrict__ mesh, const r3n_skinning_input40 *__restrict__ inputs,
                                                       const float *__restrict__ joint_matrices, const uint32_t *__restrict__ wave_skeleton,
                                                       const uint32_t *__restrict__ wave_first, const uint32_t *__restrict__ skeleton_joints,
                                                       uint32_t total_waves) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4u + (threadIdx.x >> 6));
    if (w >= total_waves) return;
    const uint32_t sk = __builtin_amdgcn_readfirstlane(wave_skeleton[w]);
    const r3n_skinning_input40 in = inputs[sk];
    const uint32_t nj = skeleton_joints[sk];  // <= 4 (checked on the host)
    const uint32_t v0 = (w - wave_first[sk]) * 64u;
    // A operands: lane l holds A[m = l & 15][k = l >> 4] = row m = (joint j = m / 4, component r = m % 4), column k of joint j
    const uint32_t m = lane & 15u, k = lane >> 4, j = m >> 2, r = m & 3u;
    float a_pos = 0.0f, a_dir = 0.0f;
    if (j < nj) {
        const float *jm = joint_matrices + 16u * (size_t)(in.joint_matrix_base_offset + j);
        a_pos = jm[4u * k + r];
        if (k < 3u && r < 3u) {  // mat3 with its columns scaled by 1 / |column|^2 (skinning.wgsl:46-52 inverse-scale trick)
            const float c[3] = {jm[4u * k], jm[4u * k + 1u], jm[4u * k + 2u]};
            a_dir = a_pos * (1.0f / dot3(c, c));
        }
    }
#pragma unroll 1
    for (uint32_t b = 0; b < 4u; ++b) {
        const uint32_t vb = v0 + b * 16u;
        if (vb >= in.vertex_count) break;  // wave-uniform
        const uint32_t v = vb + (lane & 15u);          // this lane's column: vertex v, K index k = lane >> 4
        const bool live = v < in.vertex_count;
        // B operands: lane l holds B[k = l >> 4][n = l & 15] = component k of vertex n (1 / 0 for k = 3)
        auto comp = [&](uint32_t off, float w3) {
            if (off == R3N_INVALID || !live) return 0.0f;
            return k < 3u ? __uint_as_float(mesh[off / 4u + v * 3u + k]) : w3;
        };
        const float b_pos = comp(in.base_position_offset, 1.0f), b_nrm = comp(in.base_normal_offset, 0.0f), b_tan = comp(in.base_tangent_offset, 0.0f);
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        // D: lane l, register i = row 4 * (l >> 4) + i = joint slot g = l >> 4, component i; column l & 15 = the vertex
        const f32x4 qp = __builtin_amdgcn_mfma_f32_16x16x4f32(a_pos, b_pos, zero, 0, 0, 0);
        const f32x4 qn = __builtin_amdgcn_mfma_f32_16x16x4f32(a_dir, b_nrm, zero, 0, 0, 0);
        const f32x4 qt = __builtin_amdgcn_mfma_f32_16x16x4f32(a_dir, b_tan, zero, 0, 0, 0);
        // this lane's joint slot g = lane >> 4: total weight of the vertex's influences that name it
        float W = 0.0f;
        if (live) {
            const r3n_words2 jj = *reinterpret_cast<const r3n_words2 *>(mesh + in.joint_indices_offset / 4u + v * 2u);
            const r3n_words4 jw = *reinterpret_cast<const r3n_words4 *>(mesh + in.joint_weight_offset / 4u + v * 4u);
            const uint32_t ji[4] = {jj.x & 0xFFFFu, jj.x >> 16, jj.y & 0xFFFFu, jj.y >> 16};
            const float wt[4] = {__uint_as_float(jw.x), __uint_as_float(jw.y), __uint_as_float(jw.z), __uint_as_float(jw.w)};
#pragma unroll
            for (int i = 0; i < 4; ++i) W += (ji[i] == k && wt[i] > 0.0f) ? wt[i] : 0.0f;  // k == lane >> 4 == the joint slot
        }
        float pa[3], na[3], ta[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            pa[c] = sum_groups(qp[c] * W, lane);
            na[c] = sum_groups(qn[c] * W, lane);
            ta[c] = sum_groups(qt[c] * W, lane);
        }
        if (lane < 16u && live) {
            normalize3(na);
            normalize3(ta);
            if (in.updated_position_offset != R3N_INVALID)
                *reinterpret_cast<r3n_words3 *>(mesh + in.updated_position_offset / 4u + v * 3u) = r3n_words3{__float_as_uint(pa[0]), __float_as_uint(pa[1]), __float_as_uint(pa[2])};
            if (in.updated_normal_offset != R3N_INVALID)
                *reinterpret_cast<r3n_words3 *>(mesh + in.updated_normal_offset / 4u + v * 3u) = r3n_words3{__float_as_uint(na[0]), __float_as_uint(na[1]), __float_as_uint(na[2])};
            if (in.updated_tangent_offset != R3N_INVALID)
                *reinterpret_cast<r3n_words3 *>(mesh + in.updated_tangent_offset / 4u + v * 3u) = r3n_words3{__float_as_uint(ta[0]), __float_as_uint(ta[1]), __float_as_uint(ta[2])};
        }
    }
}

extern "C" int r3n_internal_skinning_mfma(uint32_t *mesh, const void *inputs, const float *joint_matrices, const uint32_t *wave_skeleton,
                                          const uint32_t *wave_first, const uint32_t *skeleton_joints, uint32_t total_waves, hipStream_t stream) {
    hipLaunchKernelGGL(k_skinning_mfma, dim3((total_waves + 3u) / 4u), dim3(256), 0, stream, mesh, static_cast<const r3n_skinning_input40 *>(inputs),
                       joint_matrices, wave_skeleton, wave_first, skeleton_joints, total_waves);
    return (int)hipGetLastError();
}
