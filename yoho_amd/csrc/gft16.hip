// BN + ReLU between two Fourier-domain convs on the fp16x2 split MFMA:
//     out = F * relu(s * (F^T * in) + t)            F = 60 x 60 group-Fourier matrix (orthogonal)
// One (32-keypoint tile, 8-channel block) chunk = 60 coefficient slabs x 256 columns (col = h*128 + kp*4 + e) at a time;
// persistent workgroups walk the chunks with the next chunk's LDS DMA in flight.  Both 64x64x64 (zero padded) products
// per wave run on v_mfma_f32_32x32x16_f16 with each operand as two fp16 planes (hi + lo, three products per term).
//   product 1: A = F^T planes (registers, constant), B = coefficients read from the fp32 chunk in LDS; a lane reads 8
//              coefficients of its two columns (ds_read_b64 per coefficient) and splits them itself.
//   product 2: A = F planes with its columns permuted to the order in which product 1 leaves the group elements in
//              the accumulator registers, so the activated values go from registers straight into B fragments.
// Output: fp16x2 operand planes of the irrep GEMMs (gemmf.hip), gathered to 16-byte units through LDS, or the fp32
// coefficient chunk again (input of the fp32 Fourier kernel).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include <vector>
#include <cmath>
#include <cstring>

#include "common.h"

namespace yoho {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int GB = 65536;                       // one chunk buffer: 64 rows x 1 KiB (rows 60..63 stay zero)
constexpr int G16_LDS = 2 * GB + 1024;          // + per-coefficient output offsets
constexpr int G16X_MAXC = 512;                  // gft16x keeps the BN scale / shift of every channel in LDS
constexpr int G16X_LDS = G16_LDS + 2 * G16X_MAXC * 4;
constexpr float F_SCALE = 1024.f;               // |F| <= sqrt(5/60): planes of F * 2^10

struct Gft16Args {
    const float* in;
    float* out32;
    char* planes;
    const uintx4* Ffrag;      // [matrix 2][rb 2][kb 4][plane 2][lane 64] x 16 B
    const float* bn_s;
    const float* bn_t;
    int nChunks, C8, B;
    float* res0;              // G16_INVP: raw group-domain values at group element 0, fp32 [tile32][C8][60][h][kp32][4] (slab 0 only)
    int nTiles16;             // G16_INVP: activated values as direct-conv planes [tile16][C8][plane][60][16][8] into `planes`
    long long qbase[G];       // byte offset of coefficient q inside the operand planes (irrep pack + j and m terms)
    int qstride[G];           // bytes per 256-column tile of q's irrep (= K stages * 32 KiB)
    int* rflag;               // fp16 range flag of the context (note_range)
    int drain;                // experiment (YOHO_PARTI_DEBUG=drain): wait for every outstanding vector-memory operation instead of the counted wait
                              // bit 1 (YOHO_PARTI_DEBUG=ldnt): gft16x stages its coefficient chunks with non-temporal LDS-DMA loads
    int* ctr;                 // gft16x work stealing: [0] next chunk ticket, [1] finished workgroups (both 0 between launches), or null = static striding
    unsigned* amax;           // gft16x, gconv_mode 7: atomicMax of the largest |value| written to the planes (float bit pattern), or null
};

__device__ __forceinline__ floatx16 mfma_hh(uintx4 a, uintx4 b, floatx16 c) {
    union { uintx4 u; halfx8 h; } ca, cb;
    ca.u = a; cb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ca.h, cb.h, c, 0, 0, 0);
}

// (x0, x1) -> packed fp16 hi pair and lo pair (round to nearest even)
// `top` collects the largest magnitude a kernel WRITES as fp16 planes (bit pattern of |x|, compared as an unsigned integer: inf and
// NaN then order above every finite value, so a non-finite result is caught as well).  Above the fp16 range the hi plane would be
// +-inf; the kernel reports it through the context's range flag (common.h: note_range_bits) and the host repeats the pass in a
// wider format.  Conversions of a kernel's INPUTS (B operands built in registers) are not tracked: nothing between them and the
// kernel's outputs can swallow an inf (no ReLU on that path), so an overflow there surfaces as inf / NaN in the tracked outputs -
// and tracking inside the MFMA loops costs 40-70 registers (spills in head16 / head2).
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    floatx2 x;
    x.x = x0; x.y = x1;
    const halfx2 h = __builtin_convertvector(x, halfx2);
    const floatx2 r = x - __builtin_convertvector(h, floatx2);
    const halfx2 l = __builtin_convertvector(r, halfx2);
    __builtin_memcpy(&hi, &h, 4);
    __builtin_memcpy(&lo, &l, 4);
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo, unsigned& top) {
    top = max(max(top, __float_as_uint(x0) & 0x7FFFFFFFu), __float_as_uint(x1) & 0x7FFFFFFFu);
    split_pair(x0, x1, hi, lo);
}

// the 60 coefficient rows (1 KiB each) of chunk (tile32, c8) of an irrep-GEMM output, stored [tile32][q][C8][1 KiB] (the GEMM
// workgroup that owns one coefficient q of 32 channel groups then writes 32 KiB contiguously; the rows read here are C8 KiB apart)
__device__ __forceinline__ void stage_chunk(const float* in, int chunk, int C8, char* dst, int w, int lane) {
    const int c8 = chunk % C8, tile32 = chunk / C8;
    const char* s = reinterpret_cast<const char*>(in) + ((size_t)tile32 * G * C8 + c8) * 1024;
    const size_t rs = (size_t)C8 * 1024;
    for (int p = w; p < G; p += 4) __builtin_amdgcn_global_load_lds((gptr_t)(s + p * rs + lane * 16), (lptr_t)(dst + p * 1024), 16, 0, 0);
}

enum { G16_ACT32 = 0, G16_ACTP = 1, G16_INV = 2, G16_INVP = 3, G16_INVG = 4 };

template <int MODE>
__global__ __launch_bounds__(256, 1) void gft16_kernel(Gft16Args a) {
    constexpr bool PLANES = MODE == G16_ACTP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Lp = lane & 31, kg = lane >> 5;

    // zero rows 60..63 of both buffers (never overwritten), per-coefficient output offsets
    for (int i = tid; i < 2 * 256; i += 256) {
        const int bsel = i >> 8, j = i & 255;
        reinterpret_cast<uintx4*>(smem + bsel * GB + G * 1024)[j] = uintx4{0u, 0u, 0u, 0u};
    }
    long long* qb = reinterpret_cast<long long*>(smem + 2 * GB);
    int* qs = reinterpret_cast<int*>(smem + 2 * GB + 512);
    if (tid < G) { qb[tid] = a.qbase[tid]; qs[tid] = a.qstride[tid]; }

    uintx4 A1[2][4][2], A2[2][4][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                A1[rb][kb][pl] = a.Ffrag[(((0 * 2 + rb) * 4 + kb) * 2 + pl) * 64 + lane];
                A2[rb][kb][pl] = a.Ffrag[(((1 * 2 + rb) * 4 + kb) * 2 + pl) * 64 + lane];
            }

    unsigned top = 0u;
    int chunk = blockIdx.x;
    if (chunk < a.nChunks) stage_chunk(a.in, chunk, a.C8, smem, w, lane);
    for (int it = 0; chunk < a.nChunks; chunk += gridDim.x, ++it) {
        char* cur = smem + (it & 1) * GB;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int next = chunk + gridDim.x;
        if (next < a.nChunks) stage_chunk(a.in, next, a.C8, smem + ((it + 1) & 1) * GB, w, lane);
        const int c8 = chunk % a.C8, tile32 = chunk / a.C8;

        // ---- product 1: group domain = F^T * coefficients (this wave's 64 columns: lane -> columns 64w + 2Lp + {0,1})
        floatx16 acc[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][f][r] = 0.f;
        const char* colp = cur + (64 * w + 2 * Lp) * 4 + kg * 8 * 1024;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            floatx2 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const floatx2*>(colp + (kb * 16 + e) * 1024) * HF_ASCALE;
            uintx4 bh[2], bl[2];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned h0, l0, h1, l1;
                split_pair(v[2 * p].x, v[2 * p + 1].x, h0, l0);
                split_pair(v[2 * p].y, v[2 * p + 1].y, h1, l1);
                bh[0][p] = h0; bl[0][p] = l0; bh[1][p] = h1; bl[1][p] = l1;
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[rb][f] = mfma_hh(A1[rb][kb][1], bh[f], acc[rb][f]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[rb][f] = mfma_hh(A1[rb][kb][0], bl[f], acc[rb][f]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[rb][f] = mfma_hh(A1[rb][kb][0], bh[f], acc[rb][f]);
        }

        if constexpr (MODE == G16_INV) {
            // group-domain values straight out: y[kp][c][g] (B,32,60), 4 consecutive group elements per store
            const float osc = 1.f / (F_SCALE * HF_ASCALE);
            const int kpg = tile32 * TILE + (w & 1) * 16 + (Lp >> 1);
            if (kpg < a.B) {
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int c = c8 * 8 + (w >> 1) * 4 + 2 * (Lp & 1) + f;
                    float* dst = a.out32 + ((size_t)kpg * F + c) * G;
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                        for (int t4 = 0; t4 < 4; ++t4) {
                            const int g0 = 32 * rb + 8 * t4 + 4 * kg;
                            floatx4 o;
                            o.x = acc[rb][f][4 * t4] * osc; o.y = acc[rb][f][4 * t4 + 1] * osc;
                            o.z = acc[rb][f][4 * t4 + 2] * osc; o.w = acc[rb][f][4 * t4 + 3] * osc;
                            if (g0 < G) *reinterpret_cast<floatx4*>(dst + g0) = o;
                        }
                }
            }
            continue;
        }

        if constexpr (MODE == G16_INVP || MODE == G16_INVG) {
            // inverse transform + BN + ReLU, output in the group domain for the direct cone kernels (PartII)
            const int ch0 = c8 * 8 + (w >> 1) * 4 + 2 * (Lp & 1);
            const float d1 = 1.f / (F_SCALE * HF_ASCALE);
            const float s0 = a.bn_s[ch0] * d1 * H2_ASCALE, s1 = a.bn_s[ch0 + 1] * d1 * H2_ASCALE;
            const float t0 = a.bn_t[ch0] * H2_ASCALE, t1 = a.bn_t[ch0 + 1] * H2_ASCALE;
            const int kp = (w & 1) * 16 + (Lp >> 1);
            if (kg == 0) {                                     // group element 0 = accumulator row 0 of the lanes with kg = 0
                floatx2 o;
                o.x = acc[0][0][0] * d1; o.y = acc[0][1][0] * d1;
                *reinterpret_cast<floatx2*>(a.res0 + (((((size_t)tile32 * a.C8 + c8) * G + 0) * 2 + (w >> 1)) * TILE + kp) * 4 + 2 * (Lp & 1)) = o;
            }
            __syncthreads();                                   // every wave is done reading the coefficients of this buffer
            char* st = cur + (w >> 1) * 256 + kp * 8 + (Lp & 1) * 4;     // staging image [plane][g][h][kp 32][4 ch] fp16
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int g = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    unsigned hi, lo;
                    split_pair(fmaxf(acc[rb][0][r] * s0 + t0, 0.f), fmaxf(acc[rb][1][r] * s1 + t1, 0.f), hi, lo, top);
                    if (g < G) {
                        *reinterpret_cast<unsigned*>(st + g * 512) = hi;
                        *reinterpret_cast<unsigned*>(st + 30720 + g * 512) = lo;
                    }
                }
            __syncthreads();
            if constexpr (MODE == G16_INVG) {
                // B-operand stage blocks of the cone GEMM (gemmf2.hip: cgemm_kernel): [column tile of 256 matches][cone slot][32-channel block]
                // x 32 KiB = [plane][K16 sub-step][K8 group][col 256][8 ch]; slot = a.qstride[g] (< 0: the cone does not read this element)
                const int cb = c8 >> 2, sub = (c8 >> 1) & 1, kgr = c8 & 1, nslot = a.nTiles16, CB = a.C8 >> 2;
                char* dstg = a.planes + ((size_t)(tile32 >> 3) * nslot * CB + cb) * 32768 + ((sub * 2 + kgr) * 256 + (tile32 & 7) * 32) * 16;
#pragma unroll
                for (int i = 0; i < 15; ++i) {
                    const int idx = i * 256 + tid;             // (plane, g, kp)
                    const int pl = idx >= 1920 ? 1 : 0, rem = idx - pl * 1920;
                    const int g = rem >> 5, kpp = rem & 31;
                    const int sl = qs[g];
                    const char* sp = cur + pl * 30720 + g * 512 + kpp * 8;
                    const uint2 c03 = *reinterpret_cast<const uint2*>(sp);
                    const uint2 c47 = *reinterpret_cast<const uint2*>(sp + 256);
                    if (sl >= 0) *reinterpret_cast<uintx4*>(dstg + (size_t)sl * CB * 32768 + pl * 16384 + kpp * 16) = uintx4{c03.x, c03.y, c47.x, c47.y};
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                const int idx = i * 256 + tid;                 // (plane, g, kp)
                const int pl = idx >= 1920 ? 1 : 0, rem = idx - pl * 1920;
                const int g = rem >> 5, kpp = rem & 31;
                const int t16 = 2 * tile32 + (kpp >> 4);
                const char* sp = cur + pl * 30720 + g * 512 + kpp * 8;
                const uint2 c03 = *reinterpret_cast<const uint2*>(sp);
                const uint2 c47 = *reinterpret_cast<const uint2*>(sp + 256);
                if (t16 < a.nTiles16)
                    *reinterpret_cast<uintx4*>(a.planes + (((size_t)t16 * a.C8 + c8) * 2 + pl) * 15360 + g * 256 + (kpp & 15) * 16) =
                        uintx4{c03.x, c03.y, c47.x, c47.y};
            }
            continue;
        }

        // ---- BN + ReLU in registers, product 2: coefficients = F * activated
        // column -> channel: col = h*128 + kp*4 + e, channel = c8*8 + h*4 + e; this lane: e = 2*(Lp&1) + f
        const int ch0 = c8 * 8 + (w >> 1) * 4 + 2 * (Lp & 1);
        const float dscale1 = 1.f / (F_SCALE * HF_ASCALE);
        const float s0 = a.bn_s[ch0] * dscale1 * H2_ASCALE, s1 = a.bn_s[ch0 + 1] * dscale1 * H2_ASCALE;
        const float t0 = a.bn_t[ch0] * H2_ASCALE, t1 = a.bn_t[ch0 + 1] * H2_ASCALE;
        floatx16 acc2[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[rb][f][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            // K block kb of product 2 = registers 8*(kb&1) .. +7 of accumulator row block kb>>1 (the permuted order of A2)
            uintx4 bh[2], bl[2];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = 8 * (kb & 1) + 2 * p;
                const float y00 = fmaxf(acc[kb >> 1][0][r] * s0 + t0, 0.f), y01 = fmaxf(acc[kb >> 1][0][r + 1] * s0 + t0, 0.f);
                const float y10 = fmaxf(acc[kb >> 1][1][r] * s1 + t1, 0.f), y11 = fmaxf(acc[kb >> 1][1][r + 1] * s1 + t1, 0.f);
                unsigned h0, l0, h1, l1;
                split_pair(y00, y01, h0, l0);
                split_pair(y10, y11, h1, l1);
                bh[0][p] = h0; bl[0][p] = l0; bh[1][p] = h1; bl[1][p] = l1;
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc2[rb][f] = mfma_hh(A2[rb][kb][1], bh[f], acc2[rb][f]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc2[rb][f] = mfma_hh(A2[rb][kb][0], bl[f], acc2[rb][f]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc2[rb][f] = mfma_hh(A2[rb][kb][0], bh[f], acc2[rb][f]);
        }

        // ---- output.  acc2[rb][f][r]: coefficient q = 32 rb + (r&3) + 8 (r>>2) + 4 kg, column 64w + 2Lp + f
        if (PLANES) {
            const float osc = HF_ASCALE / (F_SCALE * H2_ASCALE);
            __syncthreads();                                   // every wave is done reading the coefficients of this buffer
            // staging image [plane][q][h][kp 32][4 ch] fp16: the 32 lanes of a half-wave write 128 contiguous bytes
            const int kp = (w & 1) * 16 + (Lp >> 1);
            char* st = cur + (w >> 1) * 256 + kp * 8 + (Lp & 1) * 4;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    unsigned hi, lo;
                    split_pair(acc2[rb][0][r] * osc, acc2[rb][1][r] * osc, hi, lo, top);
                    if (q < G) {
                        *reinterpret_cast<unsigned*>(st + q * 512) = hi;
                        *reinterpret_cast<unsigned*>(st + 30720 + q * 512) = lo;
                    }
                }
            __syncthreads();
            char* dst0 = a.planes + (size_t)(c8 >> 2) * 32768 + (c8 & 3) * 4096 + (tile32 & 7) * 512;
            const int nt = tile32 >> 3;
#pragma unroll
            for (int i = 0; i < 15; ++i) {
                const int idx = i * 256 + tid;                 // (plane, q, kp): 16 bytes = channels 0-3 (h = 0) | 4-7 (h = 1)
                const int pl = idx >= 1920 ? 1 : 0, rem = idx - pl * 1920;
                const int q = rem >> 5, kpp = rem & 31;
                const char* sp = cur + pl * 30720 + q * 512 + kpp * 8;
                const uint2 c03 = *reinterpret_cast<const uint2*>(sp);
                const uint2 c47 = *reinterpret_cast<const uint2*>(sp + 256);
                *reinterpret_cast<uintx4*>(dst0 + qb[q] + (long long)nt * qs[q] + kpp * 16 + pl * 16384) = uintx4{c03.x, c03.y, c47.x, c47.y};
            }
        } else {
            const float osc = 1.f / (F_SCALE * H2_ASCALE);
            float* dst = a.out32 + (size_t)chunk * CHUNK_FLOATS + 64 * w + 2 * Lp;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kg;
                    floatx2 o;
                    o.x = acc2[rb][0][r] * osc; o.y = acc2[rb][1][r] * osc;
                    if (q < G) *reinterpret_cast<floatx2*>(dst + q * 256) = o;
                }
        }
    }
    note_range_bits(a.rflag, top);
    if constexpr (MODE == G16_INVG) {
        // largest plane value written, for the fp8 scale of the cone GEMM's correction products (as gft16x leaves it for fgemm3c)
        if (a.amax) {
            unsigned m = top;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
            if (lane == 0 && m) atomicMax(a.amax, m);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// gft16x_kernel: the BN + ReLU transform between two irrep GEMMs (gft16_kernel<G16_ACTP>) with TWO waves per SIMD.
// gft16_kernel runs one wave per SIMD (388 registers: both transform matrices as A fragments + 2 x 64 accumulators), and a wave
// is in order: while it converts, stages or stores, its matrix pipe idles - the kernel's time follows the shader clock, not the
// HBM rate.  Here the 256 columns of a chunk are split over eight waves (32 columns each: half the accumulators, <= 256
// registers), so every SIMD has a second wave to fill those phases.  Same products in the same order per column: bit-identical
// planes.  Wave w8: columns 32 w8 + n (n = lane & 31), i.e. col = h*128 + kp*4 + e with h = w8 >> 2, kp = 8 (w8 & 3) + (n >> 2),
// e = n & 3.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned half_of(float x) {
    const _Float16 h = (_Float16)x;
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

__global__ __launch_bounds__(512, 2) void gft16x_kernel(Gft16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, kg = lane >> 5;

    for (int i = tid; i < 2 * 256; i += 512) {               // zero rows 60..63 of both buffers (never overwritten)
        const int bsel = i >> 8, j = i & 255;
        reinterpret_cast<uintx4*>(smem + bsel * GB + G * 1024)[j] = uintx4{0u, 0u, 0u, 0u};
    }
    long long* qb = reinterpret_cast<long long*>(smem + 2 * GB);
    int* qs = reinterpret_cast<int*>(smem + 2 * GB + 512);
    if (tid < G) { qb[tid] = a.qbase[tid]; qs[tid] = a.qstride[tid]; }
    // BN scale / shift of all channels in LDS: an ordinary global load inside the loop would make the compiler drain the
    // vector-memory counter (vmcnt(0)) at its use - with it the LDS DMA of the next chunk and the stores of the last one
    float* bnl = reinterpret_cast<float*>(smem + G16_LDS);
    for (int i = tid; i < a.C8 * 8; i += 512) { bnl[i] = a.bn_s[i]; bnl[G16X_MAXC + i] = a.bn_t[i]; }

    uintx4 A1[2][4][2], A2[2][4][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                A1[rb][kb][pl] = a.Ffrag[(((0 * 2 + rb) * 4 + kb) * 2 + pl) * 64 + lane];
                A2[rb][kb][pl] = a.Ffrag[(((1 * 2 + rb) * 4 + kb) * 2 + pl) * 64 + lane];
            }
    // workgroup barrier that does not touch the vector-memory counter (__syncthreads() would wait vmcnt(0) while an LDS DMA
    // is pending): LDS traffic of this wave drained, then the raw barrier
    auto lds_barrier = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stage = [&](int chunk, char* dst) {
        const int c8 = chunk % a.C8, tile32 = chunk / a.C8;
        const char* src = reinterpret_cast<const char*>(a.in) + ((size_t)tile32 * G * a.C8 + c8) * 1024;
        const size_t rs = (size_t)a.C8 * 1024;
#ifdef YOHO_EXPERIMENTS
        if (a.drain & 2) {
            for (int p = w8; p < G; p += 8) __builtin_amdgcn_global_load_lds((gptr_t)(src + p * rs + lane * 16), (lptr_t)(dst + p * 1024), 16, 0, 2);
            return;
        }
#endif
        for (int p = w8; p < G; p += 8) __builtin_amdgcn_global_load_lds((gptr_t)(src + p * rs + lane * 16), (lptr_t)(dst + p * 1024), 16, 0, 0);
    };
    const int hsel = w8 >> 2, kp = 8 * (w8 & 3) + (n >> 2), e = n & 3;
    unsigned top = 0u;
    // Chunks are TAKEN, not dealt (a.ctr): the grid is one persistent workgroup per CU with 136 KB of LDS, so a CU that holds any other
    // resident workgroup (the estimator's kernels of the pair pipeline, a clock probe) starts its workgroup late - with static striding
    // that workgroup's 1/256 of the chunks became a second round (+4 % streamed, +50 % beside a 1 ms probe, DESIGN 8 of round 3); with
    // tickets a late workgroup simply takes fewer.  Thread 0 draws tickets from a device counter: two in the prologue (the chunk to
    // compute and the chunk to stage at the top of iteration 0), then one per iteration - requested right behind the iteration's DMA,
    // read at the top of the next one where the counted wait for that DMA has covered it (vector-memory operations complete in issue
    // order), handed to the other waves through LDS at the barrier that is there anyway.  A chunk's arithmetic does not depend on who
    // takes it: same bits.  The last workgroup to finish zeroes the counters for the next launch.
    int* ids = reinterpret_cast<int*>(smem + 2 * GB + 768);      // [0], [1]: ticket slots of the loop, written by thread 0 in front of a barrier; [2]: the prologue's
    const bool steal = a.ctr != nullptr;
    int chunk = blockIdx.x, nxt = blockIdx.x + gridDim.x;
    int ticket = 0;
    if (steal) {
        // the prologue's pair of tickets travels through a slot of its own: iteration 0 writes slot 0 in front of its first barrier, and
        // nothing orders the other waves' read below before that write (only the latency of wave 0's DMA wait did)
        if (tid == 0) {
            const int t0 = __hip_atomic_fetch_add(a.ctr, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ids[2] = t0;
        }
        __syncthreads();
        chunk = ids[2];
        nxt = chunk + 1;
    }
    if (chunk < a.nChunks) stage(chunk, smem);
    // first chunk, tables, transform fragments: a wait the compiler's own scoreboard sees (vmcnt(0) expcnt(7) lgkmcnt(15)) - behind
    // an opaque asm wait it would put a vmcnt(0) in front of the first use of the fragments, inside the loop
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int it = 0; chunk < a.nChunks; ++it) {
        char* cur = smem + (it & 1) * GB;
        // vector-memory operations complete in issue order: behind this chunk's DMA (issued one iteration ago) are only the
        // plane stores of the previous chunk - 8 per thread in waves 0-3, 7 in waves 4-7 - which may stay in flight
        if (it > 0) {
            if (a.drain & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (w8 < 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        }
        if (steal) {
            // Thread 0 (EXEC = lane 0 of wave 0; empty in the other waves, where the two memory instructions are not issued at all):
            // publish the ticket requested one iteration ago - the counted wait above has covered its return, it was issued before
            // that iteration's plane stores - and request the next one INTO THE SAME REGISTER.  Hidden from the compiler on purpose:
            // a returning atomic it knows about makes it drain the vector-memory counter at the next control-flow join, i.e. the
            // DMA just issued (measured in the ISA: s_waitcnt vmcnt(0) right behind the atomic).  A hidden operation can only make
            // the compiler's own counted waits stricter.  `ticket` must stay in one physical register and untouched between two
            // executions of this block: tools/check_isa.py verifies that on the built object (no spill, no copy, no other reference).
            const unsigned emask = (unsigned)__builtin_amdgcn_readfirstlane(w8 == 0 ? 1 : 0);
            unsigned long long esave;
            asm volatile("s_mov_b64 %[sv], exec\n\t"
                         "s_mov_b32 exec_lo, %[m]\n\t"
                         "s_mov_b32 exec_hi, 0\n\t"
                         "ds_write_b32 %[slot], %[t]\n\t"
                         "global_atomic_add %[t], %[addr], %[one], off sc0\n\t"
                         "s_mov_b64 exec, %[sv]"
                         : [t] "+v"(ticket), [sv] "=&s"(esave)
                         : [m] "s"(emask), [slot] "v"((unsigned)(size_t)(ids + (it & 1))), [addr] "v"(a.ctr), [one] "v"(1)
                         : "memory");
        }
        lds_barrier();
        if (steal && it > 0) {
            // through asm: in front of a C++ LDS load the compiler waits for every pending vector-memory operation
            int t;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((unsigned)(size_t)(ids + (it & 1))) : "memory");
            nxt = __builtin_amdgcn_readfirstlane(t);
        }
        const int next = nxt;
        if (next < a.nChunks) stage(next, smem + ((it + 1) & 1) * GB);
        const int c8 = chunk % a.C8, tile32 = chunk / a.C8;

        // ---- product 1: group domain = F^T * coefficients, this lane's column
        floatx16 acc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
        const char* colp = cur + (32 * w8 + n) * 4 + kg * 8 * 1024;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const float*>(colp + (kb * 16 + i) * 1024) * HF_ASCALE;
            uintx4 bh, bl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned h, l;
                split_pair(v[2 * p], v[2 * p + 1], h, l);
                bh[p] = h; bl[p] = l;
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[rb] = mfma_hh(A1[rb][kb][1], bh, acc[rb]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[rb] = mfma_hh(A1[rb][kb][0], bl, acc[rb]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[rb] = mfma_hh(A1[rb][kb][0], bh, acc[rb]);
        }
        // ---- BN + ReLU in registers, product 2: coefficients = F * activated (K order of A2 = accumulator register order)
        const int ch = c8 * 8 + hsel * 4 + e;
        const float dscale1 = 1.f / (F_SCALE * HF_ASCALE);
        const float sc = bnl[ch] * dscale1 * H2_ASCALE, sh = bnl[G16X_MAXC + ch] * H2_ASCALE;
        floatx16 acc2[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[rb][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            uintx4 bh, bl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int r = 8 * (kb & 1) + 2 * p;
                unsigned h, l;
                split_pair(fmaxf(acc[kb >> 1][r] * sc + sh, 0.f), fmaxf(acc[kb >> 1][r + 1] * sc + sh, 0.f), h, l);
                bh[p] = h; bl[p] = l;
            }
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc2[rb] = mfma_hh(A2[rb][kb][1], bh, acc2[rb]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc2[rb] = mfma_hh(A2[rb][kb][0], bl, acc2[rb]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc2[rb] = mfma_hh(A2[rb][kb][0], bh, acc2[rb]);
        }
        // ---- output planes through the staging image [plane][q][h][kp 32][4 ch] fp16 (in the chunk buffer just consumed)
        const float osc = HF_ASCALE / (F_SCALE * H2_ASCALE);
        lds_barrier();                                     // every wave is done reading the coefficients of this buffer
        char* st = cur + hsel * 256 + kp * 8 + e * 2;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kg;
                const float x = acc2[rb][r] * osc;
                top = max(top, __float_as_uint(x) & 0x7FFFFFFFu);
                const _Float16 hh = (_Float16)x;
                if (q < G) {
                    *reinterpret_cast<unsigned short*>(st + q * 512) = (unsigned short)half_of(x);
                    *reinterpret_cast<unsigned short*>(st + 30720 + q * 512) = (unsigned short)half_of(x - (float)hh);
                }
            }
        lds_barrier();
        char* dst0 = a.planes + (size_t)(c8 >> 2) * 32768 + (c8 & 3) * 4096 + (tile32 & 7) * 512;
        const int nt = tile32 >> 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = i * 512 + tid;                 // (plane, q, kp): 16 bytes = channels 0-3 (h = 0) | 4-7 (h = 1)
            if (idx < 2 * G * 32) {
                const int pl = idx >= 1920 ? 1 : 0, rem = idx - pl * 1920;
                const int q = rem >> 5, kpp = rem & 31;
                const char* sp = cur + pl * 30720 + q * 512 + kpp * 8;
                // read through asm: in front of a C++ LDS load the compiler waits for every pending LDS DMA - a vmcnt(0) that
                // would also drain the next chunk's DMA issued at the top of this iteration
                uint2 c03, c47;
                asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:256\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(c03), "=&v"(c47) : "v"((unsigned)(size_t)sp) : "memory");
                *reinterpret_cast<uintx4*>(dst0 + qb[q] + (long long)nt * qs[q] + kpp * 16 + pl * 16384) = uintx4{c03.x, c03.y, c47.x, c47.y};
            }
        }
        chunk = next;
        if (!steal) nxt = next + gridDim.x;
    }
    note_range_bits(a.rflag, top);
    if (a.amax) {                                            // the consumer GEMM scales its fp8 correction operands by this (fgemm3c)
        unsigned t = top;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) t = max(t, (unsigned)__shfl_xor((int)t, o));
        if (lane == 0) atomicMax(a.amax, t);
    }
    if (steal && tid == 0) {
        // every ticket this workgroup requested has been answered before it reports (the adds complete in issue order, the last one
        // is waited for by using its value); the last workgroup to report leaves the counters at zero for the next launch
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the last ticket request has been answered
        const int done = __hip_atomic_fetch_add(a.ctr + 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (done == (int)gridDim.x - 1) {
            __hip_atomic_store(a.ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.ctr + 1, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static inline unsigned short hbits(float x) {
    const _Float16 h = (_Float16)x;
    unsigned short u;
    std::memcpy(&u, &h, 2);
    return u;
}

// MFMA A fragments of F^T (natural K order), of F (K order = accumulator register order of product 1) and of F (natural K order)
void build_gft16_frags(const FourierBasis& fb, std::vector<unsigned short>& out) {
    out.assign((size_t)3 * 2 * 4 * 2 * 64 * 8, 0);
    auto Fv = [&](int q, int g) -> float { return (q < G && g < G) ? (float)fb.F[q * G + g] : 0.f; };
    for (int mat = 0; mat < 3; ++mat)
        for (int rb = 0; rb < 2; ++rb)
            for (int kb = 0; kb < 4; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int i = lane & 31, kg = lane >> 5;
                        float v;
                        if (mat == 0) {
                            const int g = 32 * rb + i, q = 16 * kb + 8 * kg + e;
                            v = Fv(q, g);
                        } else if (mat == 1) {
                            const int q = 32 * rb + i;
                            const int g = 32 * (kb >> 1) + (e & 3) + 8 * (2 * (kb & 1) + (e >> 2)) + 4 * kg;
                            v = Fv(q, g);
                        } else {
                            const int q = 32 * rb + i, g = 16 * kb + 8 * kg + e;      // F, natural K order (head kernel)
                            v = Fv(q, g);
                        }
                        v *= F_SCALE;
                        const _Float16 hi = (_Float16)v;
                        const size_t base = ((((size_t)mat * 2 + rb) * 4 + kb) * 2) * 512 + lane * 8 + e;
                        out[base] = hbits(v);
                        out[base + 512] = hbits(v - (float)hi);
                    }
}

int gft16_init() {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gft16_kernel<G16_ACT32>), hipFuncAttributeMaxDynamicSharedMemorySize, G16_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gft16_kernel<G16_ACTP>), hipFuncAttributeMaxDynamicSharedMemorySize, G16_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gft16_kernel<G16_INV>), hipFuncAttributeMaxDynamicSharedMemorySize, G16_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gft16_kernel<G16_INVP>), hipFuncAttributeMaxDynamicSharedMemorySize, G16_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gft16_kernel<G16_INVG>), hipFuncAttributeMaxDynamicSharedMemorySize, G16_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gft16x_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G16X_LDS));
    return 0;
}

static void fill_qtables(int kppad, int cin, long long* qbase, int* qstride) {
    int qi[G * 4];
    long long off[NIR_ORD];
    fgemm_qinfo(qi);
    fgemm_plane_offsets(kppad, cin, off);
    for (int q = 0; q < G; ++q) {
        const int t = qi[q * 4], m = qi[q * 4 + 1], j = qi[q * 4 + 2], d = qi[q * 4 + 3];
        const long long KS = d * cin / 32;
        qstride[q] = (int)(KS * 32768);
        qbase[q] = off[t] + ((long long)j * (kppad / 256) * KS + (long long)m * cin / 32) * 32768;
    }
}

// planes != null: BN + ReLU, operand planes for the irrep GEMMs (kppad columns per j);
// else bn_s != null: BN + ReLU, fp32 chunks to out32 (may alias in);
// else: inverse transform only, out32 = group-domain values (B,32,60) (C8 must be 4)
int launch_gft16(const float* in, float* out32, char* planes, int kppad, const void* Ffrag, const float* bn_s, const float* bn_t, int nTiles,
                 int C8, int nCU, hipStream_t s, int B, int* rflag, int variant, int* ctr, unsigned* amax) {
    Gft16Args a;
    a.B = B; a.res0 = nullptr; a.nTiles16 = 0; a.rflag = rflag; a.amax = amax;
    a.ctr = ctr;                       // null: static striding (the caller's context was created with YOHO_XF_STEAL=0, or the pass is small)
    a.in = in; a.out32 = out32; a.planes = planes; a.Ffrag = reinterpret_cast<const uintx4*>(Ffrag); a.bn_s = bn_s; a.bn_t = bn_t;
    a.nChunks = nTiles * C8; a.C8 = C8;
    static const int dbg_drain = [] { const char* e = experiment_env("YOHO_PARTI_DEBUG"); return (e && std::strstr(e, "drain")) ? 1 : 0; }();
    static const int dbg_xf1 = [] { const char* e = experiment_env("YOHO_PARTI_DEBUG"); return (e && std::strstr(e, "xf1")) ? 1 : 0; }();
    static const int dbg_ldnt = [] { const char* e = experiment_env("YOHO_PARTI_DEBUG"); return (e && std::strstr(e, "ldnt")) ? 2 : 0; }();
    a.drain = dbg_drain | dbg_ldnt;
    if (dbg_xf1) variant = 1;
    if (planes) {
        fill_qtables(kppad, C8 * 8, a.qbase, a.qstride);
    } else {
        for (int q = 0; q < G; ++q) { a.qbase[q] = 0; a.qstride[q] = 0; }
    }
    // experiment (YOHO_PARTI_DEBUG=xfgridN): N workgroups per CU's worth of chunks instead of one persistent workgroup per CU
    static const int gmult = [] { const char* e = experiment_env("YOHO_PARTI_DEBUG"); const char* q = e ? std::strstr(e, "xfgrid") : nullptr; return q ? std::atoi(q + 6) : 1; }();
    const int want = nCU * (gmult > 0 ? gmult : 1);
    const int grid = a.nChunks < want ? a.nChunks : want;
    if (grid == 0) return 0;
    // every workgroup draws TWO tickets up front: with fewer than two chunks per workgroup half of them would get nothing and the rest
    // two each (a pass of <= 256 keypoints, the tail of a chunked pass) - static striding deals those one each
    if (a.nChunks <= 2 * grid) a.ctr = nullptr;
    if (planes && variant != 1 && a.C8 * 8 <= G16X_MAXC) hipLaunchKernelGGL(gft16x_kernel, dim3(grid), dim3(512), G16X_LDS, s, a);       // two waves per SIMD
    else if (planes) hipLaunchKernelGGL(gft16_kernel<G16_ACTP>, dim3(grid), dim3(256), G16_LDS, s, a);
    else if (bn_s) hipLaunchKernelGGL(gft16_kernel<G16_ACT32>, dim3(grid), dim3(256), G16_LDS, s, a);
    else hipLaunchKernelGGL(gft16_kernel<G16_INV>, dim3(grid), dim3(256), G16_LDS, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// head: x (B,32,60) fp32 -> Fourier coefficients as operand planes of the first irrep GEMM (cin = 32).
// One wave per (32-keypoint tile, 8-channel block): MFMA column = keypoint, eight column blocks = the eight channels,
// so a lane ends up with the 8 channels of one (coefficient, keypoint) = one 16-byte unit of each plane.  The group
// axis is contiguous in x, i.e. K-contiguous per column: B fragments are two 16-byte global loads.
// ---------------------------------------------------------------------------------------------------------------
struct Head16Args {
    const float* x;
    const float* x1;          // rows >= B0 come from x1 (second fragment of a pair), or null
    int B0;
    char* planes;
    const uintx4* Ffrag;
    int B, nTiles;
    long long qbase[G];
    int qstride[G];
    int* rflag;
};

__global__ __launch_bounds__(256, 1) void head16_kernel(Head16Args a) {
    __shared__ long long qb[G];
    __shared__ int qs[G];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);       // = 8-channel block
    const int Lp = lane & 31, kg = lane >> 5;
    if (tid < G) { qb[tid] = a.qbase[tid]; qs[tid] = a.qstride[tid]; }
    __syncthreads();
    const int tile32 = blockIdx.x;
    const int kp = tile32 * TILE + Lp;
    unsigned top = 0u;
    uintx4 A[2][4][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) A[rb][kb][pl] = a.Ffrag[(((2 * 2 + rb) * 4 + kb) * 2 + pl) * 64 + lane];

    floatx16 acc[8][2];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][rb][r] = 0.f;
        const float* xrow = (a.x1 && kp >= a.B0) ? a.x1 + (size_t)(kp - a.B0) * (F * G) : a.x + (size_t)kp * (F * G);
        const float* xp = xrow + (size_t)(8 * w + f) * G + 8 * kg;
        floatx4 v[4][2];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const bool ok = kp < a.B;
            v[kb][0] = ok ? *reinterpret_cast<const floatx4*>(xp + 16 * kb) : floatx4{0.f, 0.f, 0.f, 0.f};
            v[kb][1] = (ok && (16 * kb + 8 * kg + 4) < G) ? *reinterpret_cast<const floatx4*>(xp + 16 * kb + 4) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            uintx4 bh, bl;
            unsigned h, l;
            split_pair(v[kb][0].x * H2_ASCALE, v[kb][0].y * H2_ASCALE, h, l); bh.x = h; bl.x = l;
            split_pair(v[kb][0].z * H2_ASCALE, v[kb][0].w * H2_ASCALE, h, l); bh.y = h; bl.y = l;
            split_pair(v[kb][1].x * H2_ASCALE, v[kb][1].y * H2_ASCALE, h, l); bh.z = h; bl.z = l;
            split_pair(v[kb][1].z * H2_ASCALE, v[kb][1].w * H2_ASCALE, h, l); bh.w = h; bl.w = l;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[f][rb] = mfma_hh(A[rb][kb][1], bh, acc[f][rb]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[f][rb] = mfma_hh(A[rb][kb][0], bl, acc[f][rb]);
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) acc[f][rb] = mfma_hh(A[rb][kb][0], bh, acc[f][rb]);
        }
    }
    // acc[f][rb][r]: coefficient q = 32 rb + (r&3) + 8 (r>>2) + 4 kg of (keypoint Lp, channel 8w + f)
    const float osc = HF_ASCALE / (F_SCALE * H2_ASCALE);
    char* dst0 = a.planes + (w & 3) * 4096 + ((tile32 & 7) * 32 + Lp) * 16;         // cin = 32: K stage 0 of every m
    const int nt = tile32 >> 3;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kg;
            uintx4 ph, pl;
            unsigned h, l;
            split_pair(acc[0][rb][r] * osc, acc[1][rb][r] * osc, h, l, top); ph.x = h; pl.x = l;
            split_pair(acc[2][rb][r] * osc, acc[3][rb][r] * osc, h, l, top); ph.y = h; pl.y = l;
            split_pair(acc[4][rb][r] * osc, acc[5][rb][r] * osc, h, l, top); ph.z = h; pl.z = l;
            split_pair(acc[6][rb][r] * osc, acc[7][rb][r] * osc, h, l, top); ph.w = h; pl.w = l;
            if (q < G) {
                char* d = dst0 + qb[q] + (long long)nt * qs[q];
                *reinterpret_cast<uintx4*>(d) = ph;
                *reinterpret_cast<uintx4*>(d + 16384) = pl;
            }
        }
    note_range_bits(a.rflag, top);
}

int launch_head16(const float* x, int B, int nTiles, char* planes, int kppad, const void* Ffrag, hipStream_t s, const float* x1, int B0,
                  int* rflag) {
    Head16Args a;
    a.rflag = rflag;
    a.x = x; a.x1 = x1; a.B0 = B0; a.planes = planes; a.Ffrag = reinterpret_cast<const uintx4*>(Ffrag); a.B = B; a.nTiles = nTiles;
    fill_qtables(kppad, 32, a.qbase, a.qstride);
    if (nTiles == 0) return 0;
    hipLaunchKernelGGL(head16_kernel, dim3(nTiles), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// PartII: fp32 coefficients (C8 * 8 channels) -> inverse transform; raw values at group element 0 -> res0 (fp32 tile layout),
// relu(bn) of all 60 group elements -> fp16x2 planes of the direct cone kernels
int launch_gft16_invp(const float* in, float* res0, char* planes16, int nTiles16, const void* Ffrag, const float* bn_s, const float* bn_t,
                      int nTiles, int C8, int nCU, hipStream_t s, int* rflag) {
    Gft16Args a;
    a.rflag = rflag; a.drain = 0; a.ctr = nullptr; a.amax = nullptr;
    a.in = in; a.out32 = nullptr; a.planes = planes16; a.Ffrag = reinterpret_cast<const uintx4*>(Ffrag); a.bn_s = bn_s; a.bn_t = bn_t;
    a.nChunks = nTiles * C8; a.C8 = C8; a.B = 0; a.res0 = res0; a.nTiles16 = nTiles16;
    for (int q = 0; q < G; ++q) { a.qbase[q] = 0; a.qstride[q] = 0; }
    const int grid = a.nChunks < nCU ? a.nChunks : nCU;
    if (grid == 0) return 0;
    hipLaunchKernelGGL(gft16_kernel<G16_INVP>, dim3(grid), dim3(256), G16_LDS, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// The same with the activated values laid out as the B-operand stage blocks of the cone GEMM (cgemm_kernel, gemmf2.hip):
// planesG [ceil(nTiles / 8) column tiles][nslot][C8 / 4][32 KiB]; slot_of[g] = the cone slot of group element g or -1; amax (or null)
// receives the largest plane value written (atomicMax of float bit patterns; the caller zeroes it)
int launch_gft16_invg(const float* in, float* res0, char* planesG, const int* slot_of, int nslot, const void* Ffrag, const float* bn_s,
                      const float* bn_t, int nTiles, int C8, int nCU, hipStream_t s, int* rflag, unsigned* amax) {
    Gft16Args a;
    a.rflag = rflag; a.drain = 0; a.ctr = nullptr; a.amax = amax;
    a.in = in; a.out32 = nullptr; a.planes = planesG; a.Ffrag = reinterpret_cast<const uintx4*>(Ffrag); a.bn_s = bn_s; a.bn_t = bn_t;
    a.nChunks = nTiles * C8; a.C8 = C8; a.B = 0; a.res0 = res0; a.nTiles16 = nslot;
    for (int q = 0; q < G; ++q) { a.qbase[q] = 0; a.qstride[q] = slot_of[q]; }
    const int grid = a.nChunks < nCU ? a.nChunks : nCU;
    if (grid == 0) return 0;
    hipLaunchKernelGGL(gft16_kernel<G16_INVG>, dim3(grid), dim3(256), G16_LDS, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// PartII head (utils/network.py:266-269 + Conv_init's BN/ReLU :16-17): per match, the group axis of before_eqv0 /
// after_eqv0 is permuted by P[pre_idx], the four 32-channel sources are concatenated, BN(128) + ReLU, then the forward
// transform -> operand planes of the first irrep GEMM (cin = 128).  Same scheme as head16_kernel; a wave takes the
// 8-channel blocks w, w+4, w+8, w+12 = one block of each source.  Permuted sources are gathered element-wise.
// ---------------------------------------------------------------------------------------------------------------
struct Head2Args {
    const float* src[4];
    const int64_t* ridx[4];   // optional row indices into src[s] (null: row = match number), istride elements apart
    int istride;
    const int64_t* pre_idx;
    const int* P;
    const float* bn_s;
    const float* bn_t;
    char* planes;
    const uintx4* Ffrag;
    int M;
    long long qbase[G];
    int qstride[G];
    int* rflag;
};

// Permuted sources (before_eqv0 / after_eqv0): a lane needs x[m][c][P[idx_m][g']] for its own match m - 4-byte gathers at 64 different
// rows per load instruction when taken straight from global memory (1.2 TB/s, round 3).  Instead the 240-byte run (one channel, 60
// group elements) of each of the wave's 32 matches is copied into a wave-private LDS image by LDS DMA - 15 lanes x 16 bytes per
// match, four matches per instruction, whole lines used - and the permutation is applied by the LDS reads.  Double buffered per
// wave: the copy of channel f + 1 is issued right after channel f's values are in registers and lands under f's conversions and
// MFMAs.  Same values into the same arithmetic: bits unchanged.
constexpr int H2_STG = 8192;                  // one staging image: 8 groups of 4 matches, 1 KiB per group (4 x 240 B + 64 B unused)

__global__ __launch_bounds__(256, 1) void head2_kernel(Head2Args a) {
    __shared__ long long qb[G];
    __shared__ int qs[G];
    __shared__ __attribute__((aligned(16))) char stg[4][2][H2_STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Lp = lane & 31, kg = lane >> 5;
    if (tid < G) { qb[tid] = a.qbase[tid]; qs[tid] = a.qstride[tid]; }
    __syncthreads();
    const int tile32 = blockIdx.x;
    const int sidx = blockIdx.y;                           // one workgroup per (match tile, source): 4x the parallelism of a tile loop
    const int m = tile32 * TILE + Lp;
    const bool ok = m < a.M;
    unsigned top = 0u;
    uintx4 A[2][4][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) A[rb][kb][pl] = a.Ffrag[(((2 * 2 + rb) * 4 + kb) * 2 + pl) * 64 + lane];
    // source group element of this lane's K slots (g' = 16 kb + 8 kg + e) under the match's permutation
    int gsrc[4][8];
    {
        long long pi = ok ? a.pre_idx[m] : 0;
        pi = pi < 0 ? 0 : (pi > 59 ? 59 : pi);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int gp = 16 * kb + 8 * kg + e;
                gsrc[kb][e] = gp < G ? a.P[(int)pi * G + gp] : 0;
            }
    }
    const int nt = tile32 >> 3;
    const float osc = HF_ASCALE / (F_SCALE * H2_ASCALE);
    {
        const int cblk = sidx * 4 + w;                     // 8-channel block of the concatenated 128 channels
        const size_t srow = (ok && a.ridx[sidx]) ? (size_t)a.ridx[sidx][(size_t)m * a.istride] : (size_t)m;
        const float* sp = a.src[sidx] + srow * (F * G) + (size_t)(w * 8) * G;
        const bool permute = (sidx == 0) || (sidx == 2);
        // LDS-DMA staging of the permuted sources: instruction j copies matches 4 j .. 4 j + 3 of the tile (lane -> match 4 j + lane / 15,
        // 16-byte piece lane % 15; lanes 60-63 repeat pieces of the fourth match into the unused tail of the group)
        int rowj[8];
        const int dl = lane < 60 ? lane : lane - 15;
        const int dmi = dl / 15, dpc = dl - dmi * 15;
        if (permute) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int mm = tile32 * TILE + 4 * j + dmi;
                mm = mm < a.M ? mm : a.M - 1;                  // rows beyond M: any valid row (their values are masked below)
                rowj[j] = a.ridx[sidx] ? (int)a.ridx[sidx][(size_t)mm * a.istride] : mm;
            }
        }
        auto stage = [&](int f, int buf) {
            const char* base = reinterpret_cast<const char*>(a.src[sidx]) + (size_t)(w * 8 + f) * (G * 4) + dpc * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds((gptr_t)(base + (size_t)rowj[j] * (F * G * 4)), (lptr_t)(&stg[w][buf][j * 1024]), 16, 0, 0);
        };
        if (permute) stage(0, 0);
        const char* myrow = &stg[w][0][(Lp >> 2) * 1024 + (Lp & 3) * 240];
        floatx16 acc[8][2];
#pragma unroll
        for (int f = 0; f < 8; ++f) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][rb][r] = 0.f;
            const float bs = a.bn_s[cblk * 8 + f] * H2_ASCALE, bt = a.bn_t[cblk * 8 + f] * H2_ASCALE;
            const float* xp = sp + f * G;
            float v[4][8];
            if (permute) {
                const char* rowf = myrow + (f & 1) * H2_STG;   // (the compiler waits for the pending DMA in front of these reads)
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = *reinterpret_cast<const float*>(rowf + gsrc[kb][e] * 4);
                        v[kb][e] = ok ? t : 0.f;
                    }
                if (f + 1 < 8) {
                    // the values of channel f are in registers before the copy of f + 1 is issued (it targets the other buffer anyway)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    stage(f + 1, (f + 1) & 1);
                }
            } else {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const floatx4 lo4 = ok ? *reinterpret_cast<const floatx4*>(xp + 16 * kb + 8 * kg) : floatx4{0.f, 0.f, 0.f, 0.f};
                    const floatx4 hi4 = (ok && (16 * kb + 8 * kg + 4) < G) ? *reinterpret_cast<const floatx4*>(xp + 16 * kb + 8 * kg + 4)
                                                                           : floatx4{0.f, 0.f, 0.f, 0.f};
                    v[kb][0] = lo4.x; v[kb][1] = lo4.y; v[kb][2] = lo4.z; v[kb][3] = lo4.w;
                    v[kb][4] = hi4.x; v[kb][5] = hi4.y; v[kb][6] = hi4.z; v[kb][7] = hi4.w;
                }
            }
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                uintx4 bh, bl;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const bool v0 = ok && (16 * kb + 8 * kg + 2 * p) < G, v1 = ok && (16 * kb + 8 * kg + 2 * p + 1) < G;
                    const float y0 = v0 ? fmaxf(v[kb][2 * p] * bs + bt, 0.f) : 0.f;
                    const float y1 = v1 ? fmaxf(v[kb][2 * p + 1] * bs + bt, 0.f) : 0.f;
                    unsigned h, l;
                    split_pair(y0, y1, h, l);
                    bh[p] = h; bl[p] = l;
                }
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) acc[f][rb] = mfma_hh(A[rb][kb][1], bh, acc[f][rb]);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) acc[f][rb] = mfma_hh(A[rb][kb][0], bl, acc[f][rb]);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) acc[f][rb] = mfma_hh(A[rb][kb][0], bh, acc[f][rb]);
            }
        }
        // cin = 128: K index = m_coef * 128 + channel -> stage (k >> 5) = 4 m_coef + (cblk >> 2), slot (cblk & 3)
        char* dst0 = a.planes + (size_t)(cblk >> 2) * 32768 + (cblk & 3) * 4096 + ((tile32 & 7) * 32 + Lp) * 16;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * kg;
                uintx4 ph, pl;
                unsigned h, l;
                split_pair(acc[0][rb][r] * osc, acc[1][rb][r] * osc, h, l, top); ph.x = h; pl.x = l;
                split_pair(acc[2][rb][r] * osc, acc[3][rb][r] * osc, h, l, top); ph.y = h; pl.y = l;
                split_pair(acc[4][rb][r] * osc, acc[5][rb][r] * osc, h, l, top); ph.z = h; pl.z = l;
                split_pair(acc[6][rb][r] * osc, acc[7][rb][r] * osc, h, l, top); ph.w = h; pl.w = l;
                if (q < G) {
                    char* d = dst0 + qb[q] + (long long)nt * qs[q];
                    *reinterpret_cast<uintx4*>(d) = ph;
                    *reinterpret_cast<uintx4*>(d + 16384) = pl;
                }
            }
    }
    note_range_bits(a.rflag, top);
}

int launch_head2(const float* s0, const float* s1, const float* s2, const float* s3, const int64_t* pre_idx, const int* P, const float* bn_s,
                 const float* bn_t, int M, int nTiles, char* planes, int kppad, const void* Ffrag, hipStream_t s, const int64_t* const* ridx,
                 int istride, int* rflag) {
    Head2Args a;
    a.rflag = rflag;
    a.src[0] = s0; a.src[1] = s1; a.src[2] = s2; a.src[3] = s3;
    for (int i = 0; i < 4; ++i) a.ridx[i] = ridx ? ridx[i] : nullptr;
    a.istride = istride;
    a.pre_idx = pre_idx; a.P = P; a.bn_s = bn_s; a.bn_t = bn_t; a.planes = planes; a.Ffrag = reinterpret_cast<const uintx4*>(Ffrag); a.M = M;
    fill_qtables(kppad, 128, a.qbase, a.qstride);
    if (nTiles == 0) return 0;
    hipLaunchKernelGGL(head2_kernel, dim3(nTiles, 4), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
