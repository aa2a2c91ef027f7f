// Layout conversion and the memory-bound head/tail kernels of the PartI / PartII networks.
//
// External layout (the reference's): group feature (K,32,60) f32, group axis innermost.
// Internal layout (see gconv.hip):   [tile = K/32][c8 = C/8][g = 60][h = 2][kp = 32][e = 4],
//                                    channel c = c8*8 + h*4 + e.
#include "common.h"

namespace yoho {

__device__ __forceinline__ size_t iidx(int tile, int C8, int c, int g, int kp) {
    return ((((size_t)tile * C8 + (c >> 3)) * G + g) * 2 + ((c >> 2) & 1)) * (TILE * 4) + kp * 4 + (c & 3);
}

// ---------------------------------------------------------------------------------------------
// PartI input: x (B,32,60) -> internal (C8 = 4).  One workgroup per (tile, c8): the 8 channels of a
// keypoint are 480 contiguous floats in x, so reads are coalesced; the 60 KiB chunk is assembled
// in LDS and written out linearly.  Rows >= B are zero-filled.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_partI_kernel(const float* __restrict__ x, int B, float* __restrict__ out) {
    __shared__ float lds[CHUNK_FLOATS];
    const int tile = blockIdx.x >> 2, c8 = blockIdx.x & 3;
    for (int i = threadIdx.x; i < TILE * 8 * G; i += 256) {
        const int kp = i / (8 * G);
        const int r = i - kp * (8 * G);
        const int cl = r / G, g = r - cl * G;
        const int b = tile * TILE + kp;
        const float v = b < B ? x[(size_t)b * (F * G) + (c8 * 8 + cl) * G + g] : 0.f;
        lds[(g * 2 + (cl >> 2)) * (TILE * 4) + kp * 4 + (cl & 3)] = v;
    }
    __syncthreads();
    float4* o = reinterpret_cast<float4*>(out + ((size_t)tile * 4 + c8) * CHUNK_FLOATS);
    const float4* l = reinterpret_cast<const float4*>(lds);
    for (int i = threadIdx.x; i < CHUNK_FLOATS / 4; i += 256) o[i] = l[i];
}

int launch_pack_partI(const float* x, int B, int nTiles, float* out, hipStream_t s) {
    hipLaunchKernelGGL(pack_partI_kernel, dim3(nTiles * 4), dim3(256), 0, s, x, B, out);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// PartI tail (utils/network.py:98-103): eqv = y + x; inv = mean_g(eqv); eqv /= max(|eqv|_c, 1e-4);
// inv /= max(|inv|, 1e-4).  Optionally inv_np = numpy-order fp32 mean of the normalised eqv over
// the group axis (what tests/matcher.py:35 recomputes from the saved file).
// One 64-lane workgroup per keypoint.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float np_pairwise_mean60(const float* v) {
    // numpy pairwise_sum for n = 60 (< 128): 8 running sums, fixed combine tree, 4 leftovers, / 60
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = v[j];
#pragma unroll
    for (int b = 1; b < 7; ++b)
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], v[8 * b + j]);
    float s = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                        __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
#pragma unroll
    for (int j = 56; j < 60; ++j) s = __fadd_rn(s, v[j]);
    return __fdiv_rn(s, 60.0f);
}

// fp32 raw layout of the bf16x3 variant: [tile16][c8][g][kp16][8 ch]
__device__ __forceinline__ size_t iidx16(int tile, int C8, int c, int g, int kp) {
    return (((size_t)tile * C8 + (c >> 3)) * G + g) * 128 + kp * 8 + (c & 7);
}

__global__ __launch_bounds__(64) void finalize_partI_kernel(const float* __restrict__ y, const float* __restrict__ x, int B,
                                                            float* __restrict__ eqv, float* __restrict__ inv,
                                                            float* __restrict__ inv_np, int layout16, const float* __restrict__ x1, int B0) {
    __shared__ float e[F * G];
    __shared__ float rn[G];
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const int tw = layout16 == 1 ? 16 : TILE;
    const int tile = b / tw, kp = b - tile * tw;
    const float* xb = (x1 && b >= B0) ? x1 + (size_t)(b - B0) * (F * G) : x + (size_t)b * (F * G);
    for (int i = lane; i < F * G; i += 64) {
        const int c = i / G, g = i - c * G;
        // layout16: 0 = fp32 32-keypoint tile layout, 1 = 16-keypoint tile layout, 2 = plain (B,32,60)
        e[i] = y[layout16 == 2 ? (size_t)b * (F * G) + i : (layout16 ? iidx16(tile, 4, c, g, kp) : iidx(tile, 4, c, g, kp))] + xb[i];
    }
    __syncthreads();
    if (lane < G) {
        float s = 0.f;
        for (int c = 0; c < F; ++c) { const float v = e[c * G + lane]; s += v * v; }
        rn[lane] = fmaxf(sqrtf(s), 1e-4f);
    }
    if (lane < F && inv != nullptr) {
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += e[lane * G + g];
        const float m = s / 60.0f;
        float n2 = m * m;                                     // |inv|^2 over the 32 channel lanes
        for (int o = 16; o >= 1; o >>= 1) n2 += __shfl_xor(n2, o, 32);
        inv[(size_t)b * F + lane] = m / fmaxf(sqrtf(n2), 1e-4f);
    }
    __syncthreads();
    float* eb = eqv + (size_t)b * (F * G);
    for (int i = lane; i < F * G; i += 64) {
        const int g = i % G;
        const float v = e[i] / rn[g];
        e[i] = v;
        eb[i] = v;
    }
    if (inv_np != nullptr) {
        __syncthreads();
        if (lane < F) inv_np[(size_t)b * F + lane] = np_pairwise_mean60(e + lane * G);
    }
}

int launch_finalize_partI(const float* y, const float* x, int B, float* eqv, float* inv, float* inv_np, int layout16, hipStream_t s,
                          const float* x1, int B0) {
    hipLaunchKernelGGL(finalize_partI_kernel, dim3(B), dim3(64), 0, s, y, x, B, eqv, inv, inv_np, layout16, x1, B0);
    HIPCHK(hipGetLastError());
    return 0;
}

// out (B,32) = np.mean(eqv (B,32,60), axis=-1), bit-exact (tests/matcher.py:35-36).
__global__ __launch_bounds__(256) void group_mean_np_kernel(const float* __restrict__ eqv, int rows, float* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;      // row = (b, c)
    if (r >= rows) return;
    float v[G];
    const float4* p = reinterpret_cast<const float4*>(eqv + (size_t)r * G);   // 240 B rows: 16-B aligned
#pragma unroll
    for (int i = 0; i < G / 4; ++i) {
        const float4 t = p[i];
        v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w;
    }
    out[r] = np_pairwise_mean60(v);
}

int launch_group_mean_np(const float* eqv, int B, float* out, hipStream_t s) {
    const int rows = B * F;
    hipLaunchKernelGGL(group_mean_np_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, eqv, rows, out);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// PartII head (utils/network.py:266-269 + Conv_init's BN/ReLU :16-17): per match, permute the group
// axis of before_eqv0 / after_eqv0 by P[pre_idx], concatenate 4 x 32 channels, apply BN(128)+ReLU,
// write the internal layout (C8 = 16).  One workgroup per (tile, c8); c8>>2 selects the source.
// The caller's tensors are not modified (the reference permutes them in place).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_partII_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                          const float* __restrict__ s2, const float* __restrict__ s3,
                                                          const int64_t* __restrict__ pre_idx, const int* __restrict__ P,
                                                          const float* __restrict__ bn_s, const float* __restrict__ bn_t,
                                                          int M, float* __restrict__ out) {
    __shared__ float lds[CHUNK_FLOATS];
    const int tile = blockIdx.x >> 4, c8 = blockIdx.x & 15;
    const int src = c8 >> 2;
    const float* sp = src == 0 ? s0 : (src == 1 ? s1 : (src == 2 ? s2 : s3));
    const bool permute = (src == 0) || (src == 2);
    const int cbase = (c8 & 3) * 8;
    for (int i = threadIdx.x; i < TILE * 8 * G; i += 256) {
        const int kp = i / (8 * G);
        const int r = i - kp * (8 * G);
        const int cl = r / G, g = r - cl * G;
        const int m = tile * TILE + kp;
        float v = 0.f;
        if (m < M) {
            int gs = g;
            if (permute) {
                long long pi = pre_idx[m];
                pi = pi < 0 ? 0 : (pi > 59 ? 59 : pi);
                gs = P[(int)pi * G + g];
            }
            const int cc = c8 * 8 + cl;
            v = sp[(size_t)m * (F * G) + (cbase + cl) * G + gs];
            v = fmaxf(v * bn_s[cc] + bn_t[cc], 0.f);
        }
        lds[(g * 2 + (cl >> 2)) * (TILE * 4) + kp * 4 + (cl & 3)] = v;
    }
    __syncthreads();
    float4* o = reinterpret_cast<float4*>(out + ((size_t)tile * 16 + c8) * CHUNK_FLOATS);
    const float4* l = reinterpret_cast<const float4*>(lds);
    for (int i = threadIdx.x; i < CHUNK_FLOATS / 4; i += 256) o[i] = l[i];
}

int launch_pack_partII(const float* s0, const float* s1, const float* s2, const float* s3, const int64_t* pre_idx,
                       const int* P, const float* bn_s, const float* bn_t, int M, int nTiles, float* out, hipStream_t s) {
    hipLaunchKernelGGL(pack_partII_kernel, dim3(nTiles * 16), dim3(256), 0, s, s0, s1, s2, s3, pre_idx, P, bn_s, bn_t, M, out);
    HIPCHK(hipGetLastError());
    return 0;
}

// PartII tail (utils/network.py:276-277): q = fc[:, :, g=0]; q /= |q| (no clamp).
__global__ __launch_bounds__(256) void quat_norm_kernel(const float* __restrict__ y, int M, float* __restrict__ quat) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int tile = m / TILE, kp = m - tile * TILE;
    const float4 q = *reinterpret_cast<const float4*>(y + iidx(tile, 4, 0, 0, kp));
    const float n = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    float4 o; o.x = q.x / n; o.y = q.y / n; o.z = q.z / n; o.w = q.w / n;
    *reinterpret_cast<float4*>(quat + (size_t)m * 4) = o;
}

int launch_quat_norm(const float* y, int M, float* quat, hipStream_t s) {
    hipLaunchKernelGGL(quat_norm_kernel, dim3((M + 255) / 256), dim3(256), 0, s, y, M, quat);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
