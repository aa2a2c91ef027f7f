// Shared by the two irrep-GEMM kernels (gemmf.hip: 256 x 256 tiles, one workgroup per CU; gemmf2.hip: 256 x 128 tiles, two
// workgroups per CU): operand pack constants, launch arguments, small device helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "common.h"

namespace yoho {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int FG_STAGE = 32768;               // bytes of one operand tile per K32 stage
constexpr int FG_LDS = 4 * FG_STAGE;          // 2 buffers x (A + B)

// irreps in launch order (heaviest first)
static const int FG_ORD_D[NIR_ORD] = {5, 4, 3, 3, 1};
static const int FG_ORD_R[NIR_ORD] = {4, 3, 1, 2, 0};
static const int FG_IR_BASE[5] = {0, 1, 10, 19, 35};
static const int FG_IR_D[5] = {1, 3, 3, 4, 5};

struct FGemmArgs {
    const char* A;        // weight planes, all irreps
    const char* B;        // activation planes, all irreps
    const float* bias;
    const float* res;     // fp32 coefficient slabs [tile32][60 q][cout8][h][kp32][4] or null (q-major: a workgroup owns one q)
    float* out;           // same layout
    long long a_off[NIR_ORD], b_off[NIR_ORD];
    int NT[NIR_ORD], MT[NIR_ORD], rot[NIR_ORD];
    int dim[NIR_ORD], qbase[NIR_ORD];     // dimension and first coefficient index of the t-th irrep in launch order
    int cin, cout, kppad, nT32;
    float descale;
    int* rflag;           // fp16 range flag: raised when an output coefficient will not fit the consumer's fp16 planes (x HF_ASCALE)
    const unsigned* amax; // gconv_mode 7 (fgemm3c): largest |value| of the B planes as a float bit pattern, left by their producer; null = fgemm3
};

template <int I, int N, typename Fn>
__device__ __forceinline__ void sfor(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

__device__ __forceinline__ floatx16 mfma_h(uintx4 a, uintx4 b, floatx16 c) {
    union { uintx4 u; halfx8 h; } ca, cb;
    ca.u = a; cb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ca.h, cb.h, c, 0, 0, 0);
}


// uniform (SGPR) copy of a wave-uniform pointer
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

void fgemm_fill_args(FGemmArgs& a, const Layer& L, const char* Bplanes, int kppad, int nT32, const float* res, float* out, int* rflag);
int launch_fgemm2(const FGemmArgs& a, int flags, hipStream_t s);
int launch_fgemm3(const FGemmArgs& a, int flags, hipStream_t s);
int fgemm2_init();

}  // namespace yoho
