// Distance arithmetic shared by the brute-force and the grid nearest-neighbour kernels (match.hip, estim.hip,
// gridnn.hip): explicit round-to-nearest intrinsics in the summation order the reference's CPU reduction uses
// (pinned by tests/golden/pdist.npz); the translation units that include this are built with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace yoho {

template <int D>
__device__ __forceinline__ float dist2_f32(const float* a, const float* b) {
    if constexpr (D == 32) {
        float l[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = __fsub_rn(a[k], b[k]); l[k] = __fmul_rn(d, d); }
#pragma unroll
        for (int blk = 1; blk < 4; ++blk)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float d = __fsub_rn(a[8 * blk + k], b[8 * blk + k]);
                l[k] = __fadd_rn(l[k], __fmul_rn(d, d));
            }
        float s = l[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) s = __fadd_rn(s, l[k]);
        return s;
    } else {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float d = __fsub_rn(a[k], b[k]);
            const float q = __fmul_rn(d, d);
            s = k == 0 ? q : __fadd_rn(s, q);
        }
        return s;
    }
}

// pdist 'L2': sqrt(D2 + 1e-7) as torch evaluates it on fp32 (utils/knn_search.py:17-20)
__device__ __forceinline__ float dist_of_f32(float d2) { return (float)sqrt((double)__fadd_rn(d2, 1e-7f)); }

// one coordinate of keys @ Rg^T in f64 (YOHO_testset.py:157) and the f64 squared distance to a widened f32 point
struct GnMat3 { double m[9]; };
__device__ __forceinline__ double rotate_key_f64(const double* kk, const double* row) {
    return fma(kk[2], row[2], fma(kk[1], row[1], kk[0] * row[0]));
}
__device__ __forceinline__ double dist2_key_f64(const double* kr, double px, double py, double pz) {
    const double d0 = __dsub_rn(kr[0], px), d1 = __dsub_rn(kr[1], py), d2 = __dsub_rn(kr[2], pz);
    return __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
}

}  // namespace yoho
