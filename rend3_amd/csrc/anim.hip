// anim.hip -- row N4, rend3-anim on the GPU: joint matrices of many skeleton instances per frame from keyframe clips.
//
// Reference behaviour restated (rend3-anim/src/lib.rs): sample_at_time (:163-175: first key later than t, linear
// blend with the key before it, factor clamped to [0, 1]; x / 0 when both keys coincide: +-inf clamps, 0 / 0 stays NaN), Vec3 lerp and quaternion nlerp
// with a second normalize (:148-160), pose_animation_frame's per-skin part (:213-262): local matrices from the sampled
// (or bind) scale / rotation / translation of the joints the clip animates -- identity for the others --, global
// matrices down the joint hierarchy, joint matrix = global * inverse bind (Renderer::set_skeleton_joint_transforms).
// The reference does this on the CPU, one skin instance at a time, and uploads the matrices; config 5 (50 000 skinned
// instances) makes it a data-parallel path: one wavefront per (instance, clip, time) request, lane = joint, hierarchy
// levels in LDS, results written straight into the matrix buffer k_skinning reads.
//
// Arithmetic: f32, one rounding per operation, glam 0.25's scalar formulas (oracle/anim.py restates the same).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/r3n.h"

namespace {

#define R3N_ANIM_MAX_JOINTS 512u  // per rig (r3n_animation_write checks it): at most 32 KB of LDS for the matrices

__device__ inline float dot4(const float a[4], const float b[4]) { return ((a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]) + a[3] * b[3]; }
__device__ inline void normalize4(float q[4]) {
    const float r = 1.0f / sqrtf(dot4(q, q));
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = q[k] * r;
}
// sample_at_time's key pair and blend factor
// `sorted`: the key times are non-decreasing (checked once by r3n_animation_write), so "the first key later than t" is an
// upper bound found by bisection -- a real clip has hundreds of keys per channel and the reference's linear scan is a
// chain of dependent loads (measured on the animation example's character: 227 us per pose, whatever the instance
// count); unsorted channels keep the scan.
__device__ inline void sample_keys(const float *__restrict__ times, uint32_t n, bool sorted, float t, uint32_t &prv, uint32_t &nxt, float &x) {
    nxt = n - 1u;
    if (sorted) {
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (times[mid] > t) hi = mid; else lo = mid + 1u;
        }
        if (lo < n) nxt = lo;
    } else {
        for (uint32_t i = 0; i < n; ++i)
            if (times[i] > t) { nxt = i; break; }
    }
    prv = nxt ? nxt - 1u : 0u;
    x = (t - times[prv]) / (times[nxt] - times[prv]);
    if (x < 0.0f) x = 0.0f;
    if (x > 1.0f) x = 1.0f;  // NaN passes both tests and stays NaN, like f32::clamp
}
__device__ inline void mat4_from_srt(const float s[3], const float q[4], const float t[3], float m[16]) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float x2 = x + x, y2 = y + y, z2 = z + z;
    const float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
    const float xa[4] = {1.0f - (yy + zz), xy + wz, xz - wy, 0.0f};
    const float ya[4] = {xy - wz, 1.0f - (xx + zz), yz + wx, 0.0f};
    const float za[4] = {xz + wy, yz - wx, 1.0f - (xx + yy), 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[k] = xa[k] * s[0];
        m[4 + k] = ya[k] * s[1];
        m[8 + k] = za[k] * s[2];
    }
    m[12] = t[0]; m[13] = t[1]; m[14] = t[2]; m[15] = 1.0f;
}
__device__ inline void mat4_mul(const float *a, const float *b, float *o) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            o[4 * c + r] = ((a[r] * b[4 * c] + a[4 + r] * b[4 * c + 1]) + a[8 + r] * b[4 * c + 2]) + a[12 + r] * b[4 * c + 3];
}

__global__ __launch_bounds__(64) void k_pose_skeletons(const r3n_pose_request16 *__restrict__ requests, uint32_t n_requests,
                                                       const r3n_anim_rig16 *__restrict__ rigs, const r3n_anim_joint80 *__restrict__ joints,
                                                       const r3n_anim_clip16 *__restrict__ clips, const r3n_anim_track80 *__restrict__ tracks,
                                                       const float *__restrict__ times, const float *__restrict__ values,
                                                       float *__restrict__ out) {
    extern __shared__ float s_dyn[];  // 64 B per joint of the largest rig (a fixed 512-joint array would cap a CU at 5 waves)
    float (*s_m)[16] = reinterpret_cast<float (*)[16]>(s_dyn);
    const uint32_t req = blockIdx.x;
    if (req >= n_requests) return;
    const r3n_pose_request16 rq = requests[req];
    const r3n_anim_clip16 clip = clips[rq.clip];
    const r3n_anim_rig16 rig = rigs[clip.rig];
    float t = rq.time;  // pose_animation_frame: time.clamp(0.0, duration)
    if (t < 0.0f) t = 0.0f;
    if (t > clip.duration) t = clip.duration;
    const float ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    // local matrices
    for (uint32_t j = threadIdx.x; j < rig.n_joints; j += 64u) {
        const r3n_anim_track80 &tr = tracks[clip.first_track + j];
        float m[16];
        if (tr.animated & 1u) {  // bits 1..3: sorted flags added by r3n_animation_write
            float sc[3], ro[4], tl[3];
            if (tr.key_count[0]) {
                uint32_t p, n; float x;
                sample_keys(times + tr.key_first[0], tr.key_count[0], ((tr.animated >> 1) & 1u) != 0u, t, p, n, x);
                const float *a = values + tr.value_first[0] + 3u * p, *b = values + tr.value_first[0] + 3u * n;
#pragma unroll
                for (int k = 0; k < 3; ++k) tl[k] = a[k] + ((b[k] - a[k]) * x);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) tl[k] = tr.bind_t[k];
            }
            if (tr.key_count[1]) {
                uint32_t p, n; float x;
                sample_keys(times + tr.key_first[1], tr.key_count[1], ((tr.animated >> 2) & 1u) != 0u, t, p, n, x);
                const float *a = values + tr.value_first[1] + 4u * p, *b = values + tr.value_first[1] + 4u * n;
                const float qa[4] = {a[0], a[1], a[2], a[3]}, qb[4] = {b[0], b[1], b[2], b[3]};
                const float bias = dot4(qa, qb) >= 0.0f ? 1.0f : -1.0f;
#pragma unroll
                for (int k = 0; k < 4; ++k) ro[k] = qa[k] + (((qb[k] * bias) - qa[k]) * x);
                normalize4(ro);
                normalize4(ro);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) ro[k] = tr.bind_r[k];
            }
            if (tr.key_count[2]) {
                uint32_t p, n; float x;
                sample_keys(times + tr.key_first[2], tr.key_count[2], ((tr.animated >> 3) & 1u) != 0u, t, p, n, x);
                const float *a = values + tr.value_first[2] + 3u * p, *b = values + tr.value_first[2] + 3u * n;
#pragma unroll
                for (int k = 0; k < 3; ++k) sc[k] = a[k] + ((b[k] - a[k]) * x);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) sc[k] = tr.bind_s[k];
            }
            mat4_from_srt(sc, ro, tl, m);
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) m[k] = ident[k];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s_m[j][k] = m[k];
    }
    __syncthreads();
    // global matrices, one hierarchy level at a time (a joint needs only its parent's global matrix)
    for (uint32_t d = 0; d <= rig.max_depth; ++d) {
        for (uint32_t j = threadIdx.x; j < rig.n_joints; j += 64u) {
            const r3n_anim_joint80 &jt = joints[rig.first_joint + j];
            if (jt.depth != d || jt.parent == -1) continue;  // -1: no parent node, global = local
            float g[16];
            mat4_mul(jt.parent >= 0 ? s_m[jt.parent] : ident, s_m[j], g);  // -2: the parent is not a joint of this skin
#pragma unroll
            for (int k = 0; k < 16; ++k) s_m[j][k] = g[k];
        }
        __syncthreads();
    }
    for (uint32_t j = threadIdx.x; j < rig.n_joints; j += 64u) {
        float o[16];
        mat4_mul(s_m[j], joints[rig.first_joint + j].inverse_bind, o);
        float *dst = out + ((size_t)rq.matrix_base + j) * 16u;
#pragma unroll
        for (int k = 0; k < 16; ++k) dst[k] = o[k];
    }
}

}  // namespace

extern "C" int r3n_internal_pose_skeletons(const void *requests, uint32_t n, const void *rigs, const void *joints, const void *clips,
                                           const void *tracks, const float *times, const float *values, float *out, uint32_t max_joints,
                                           hipStream_t stream) {
    hipLaunchKernelGGL(k_pose_skeletons, dim3(n), dim3(64), (size_t)max_joints * 64u, stream, static_cast<const r3n_pose_request16 *>(requests), n,
                       static_cast<const r3n_anim_rig16 *>(rigs), static_cast<const r3n_anim_joint80 *>(joints),
                       static_cast<const r3n_anim_clip16 *>(clips), static_cast<const r3n_anim_track80 *>(tracks), times, values, out);
    return (int)hipGetLastError();
}
