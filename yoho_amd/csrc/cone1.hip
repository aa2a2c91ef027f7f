// PartII, last 13-tap layer (utils/network.py:272-274 evaluated at the identity only): the quaternion head reads
// group element 0 of this layer's output, so it is one dense product over (tap, channel):
//     out[o, m] = sum_k sum_c W[o,c,k] * act[c, N[0][k], m]            K = 13 * 512
// on the fp16x2 split MFMA.  Input = the activated 13-cone planes the previous layer left behind
// ([tile16][c8][plane][60][16 kp][8 ch] fp16, only the slabs N[0][k] are valid).  A workgroup owns 32 matches (two
// 16-match tiles = the two column halves of the MFMA) and 128 output channels (one 32-row block per wave); per
// 8-channel chunk only the 13 needed slabs of both tiles and planes are gathered into LDS (14 KiB, double buffered).
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

constexpr int C1_CHUNK = 30720;                 // bytes of one (tile16, c8) chunk of fp16x2 planes
constexpr int C1_PLANE = 15360;
constexpr int C1_STAGE = 2 * 2 * 14 * 256;      // [plane][tile half][tap slot 14][256 B] = 14336
constexpr int C1_UNITS = C1_STAGE / 1024;       // 14 LDS-DMA wave instructions per stage

struct Cone1Args {
    const char* X;
    const char* Wp;        // [ob][cin8][tap pair 7][plane 2][lane 64][8] fp16 (Layer::wph)
    const float* bias;
    const float* res;      // fp32 [tile32][cout8][60][h][kp32][4], slab 0 used
    float* out;            // same layout, slab 0 written
    int nTiles32, nTiles16, cin8, cout8;
    float descale;
    int n0[14];            // N[0][k]; n0[13] = n0[0] (tap 13 has zero weights, its slab only has to be finite)
};

__device__ __forceinline__ floatx16 mfma_c1(uintx4 a, uintx4 b, floatx16 c) {
    union { uintx4 u; halfx8 h; } ca, cb;
    ca.u = a; cb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ca.h, cb.h, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void cone1_kernel(Cone1Args a) {
    __shared__ __attribute__((aligned(16))) char smem[2 * C1_STAGE];
    __shared__ int n0s[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nog = a.cout8 / 16;                         // groups of 128 output channels
    const int tile32 = blockIdx.x / nog, og = blockIdx.x - tile32 * nog;
    const int ob = og * 4 + w;
    if (tid < 14) n0s[tid] = a.n0[tid];
    __syncthreads();

    // LDS-DMA gather: unit u (1 KiB) = slots 4u .. 4u+3 of [plane][half][tap]; this wave owns units w, w+4, w+8, w+12
    long long soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = w + 4 * i;
        const int slot = (u < C1_UNITS ? u : 0) * 4 + (lane >> 4);
        const int pl = slot / 28, r = slot - pl * 28, ht = r / 14, k = r - ht * 14;
        int t16 = 2 * tile32 + ht;
        t16 = t16 < a.nTiles16 ? t16 : a.nTiles16 - 1;
        soff[i] = (long long)t16 * a.cin8 * C1_CHUNK + pl * C1_PLANE + n0s[k] * 256 + (lane & 15) * 16;
    }
    auto stage = [&](int c8, char* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int u = w + 4 * i;
            if (u < C1_UNITS)
                __builtin_amdgcn_global_load_lds((gptr_t)(a.X + soff[i] + (long long)c8 * C1_CHUNK), (lptr_t)(dst + u * 1024), 16, 0, 0);
        }
    };

    floatx16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // B fragment: column = lane & 31 (tile half = col >> 4, match = col & 15), K group = lane >> 5 = tap of the pair
    const int lane_b = ((lane >> 4) & 1) * (14 * 256) + (lane >> 5) * 256 + (lane & 15) * 16;
    const uintx4* Wl = reinterpret_cast<const uintx4*>(a.Wp) + (size_t)ob * a.cin8 * (7 * 2 * 64) + lane;

    uintx4 wcur[7][2], wnxt[7][2];
#pragma unroll
    for (int tp = 0; tp < 7; ++tp) { wcur[tp][0] = Wl[(tp * 2) * 64]; wcur[tp][1] = Wl[(tp * 2 + 1) * 64]; }
    stage(0, smem);
    for (int c8 = 0; c8 < a.cin8; ++c8) {
        char* cur = smem + (c8 & 1) * C1_STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int cn = c8 + 1 < a.cin8 ? c8 + 1 : c8;
        stage(cn, smem + ((c8 + 1) & 1) * C1_STAGE);
        const uintx4* Wn = Wl + (size_t)cn * (7 * 2 * 64);
#pragma unroll
        for (int tp = 0; tp < 7; ++tp) { wnxt[tp][0] = Wn[(tp * 2) * 64]; wnxt[tp][1] = Wn[(tp * 2 + 1) * 64]; }
#pragma unroll
        for (int tp = 0; tp < 7; ++tp) {
            const uintx4 bh = *reinterpret_cast<const uintx4*>(cur + lane_b + tp * 512);
            const uintx4 bl = *reinterpret_cast<const uintx4*>(cur + lane_b + tp * 512 + 2 * 14 * 256);
            acc[0] = mfma_c1(wcur[tp][1], bh, acc[0]);      // three independent accumulation chains
            acc[1] = mfma_c1(wcur[tp][0], bl, acc[1]);
            acc[2] = mfma_c1(wcur[tp][0], bh, acc[2]);
        }
#pragma unroll
        for (int tp = 0; tp < 7; ++tp) { wcur[tp][0] = wnxt[tp][0]; wcur[tp][1] = wnxt[tp][1]; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // epilogue: D[o][col]: lane (col = lane & 31 = match within the 32-tile, half = lane >> 5), reg r -> o = (r&3) + 8 (r>>2) + 4 half
    const int kp32 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ch = ob * 32 + q * 8 + half * 4;
        floatx4 val;
        val.x = (acc[0][4 * q + 0] + acc[1][4 * q + 0]) + acc[2][4 * q + 0];
        val.y = (acc[0][4 * q + 1] + acc[1][4 * q + 1]) + acc[2][4 * q + 1];
        val.z = (acc[0][4 * q + 2] + acc[1][4 * q + 2]) + acc[2][4 * q + 2];
        val.w = (acc[0][4 * q + 3] + acc[1][4 * q + 3]) + acc[2][4 * q + 3];
        val *= a.descale;
        val += *reinterpret_cast<const floatx4*>(a.bias + ch);
        const size_t off = (((((size_t)tile32 * a.cout8 + ob * 4 + q) * G + 0) * 2 + half) * TILE + kp32) * 4;
        if (a.res) val += *reinterpret_cast<const floatx4*>(a.res + off);
        *reinterpret_cast<floatx4*>(a.out + off) = val;
    }
}

int launch_cone1(const Layer& L, const char* X, int nTiles32, int nTiles16, const float* res, float* out, const int* n0, hipStream_t s) {
    if (L.cout_pad % 128 || !L.wph) { set_error("cone1: needs cout %% 128 == 0 and fp16x2 weights"); return YOHO_EINVAL; }
    Cone1Args a;
    a.X = X; a.Wp = reinterpret_cast<const char*>(L.wph); a.bias = L.bias; a.res = res; a.out = out;
    a.nTiles32 = nTiles32; a.nTiles16 = nTiles16; a.cin8 = L.cin / 8; a.cout8 = L.cout_pad / 8; a.descale = L.wph_descale;
    for (int k = 0; k < 13; ++k) a.n0[k] = n0[k];
    a.n0[13] = n0[0];
    if (nTiles32 == 0) return 0;
    hipLaunchKernelGGL(cone1_kernel, dim3(nTiles32 * (a.cout8 / 16)), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
