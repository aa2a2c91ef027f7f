// Group-Fourier conv as five dense GEMMs on the fp16x2 split MFMA.
//
// On Fourier coefficients the 13-tap icosahedral group conv is block diagonal over the irreps r (d = 1,3,3,4,5):
//     Yhat(r,i,j)[o, kp] = sum_m sum_c What(r,i,m)[o,c] * Xhat(r,m,j)[c, kp]
// which for a fixed irrep is ONE plain matrix product
//     Y[(i,o), (j,kp)] = sum_(m,c) A[(i,o), (m,c)] * B[(m,c), (j,kp)],      M = d*Cout, K = d*Cin, N = d*KPpad
// (sum_r d^3 = 244 slab products instead of the 780 of the direct 13-tap form).  Both operands are kept as two fp16
// planes (x * 2^s = hi + lo) and every term is evaluated as lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16 with
// fp32 accumulation (error <= 3 * 2^-22 per product).
//
// Blocking: 256 x 256 output tile per workgroup, 128 x 128 per wave (4 x 4 MFMA tiles, 256 accumulator registers),
// K in stages of 32 = two MFMA sub-steps of 16.  Operands are stored in HBM already in the order the LDS wants them:
//     pack[tile 256][stage K32][plane 2][sub-step 2][k-group 2][row/col 256][8 x fp16]        (32 KiB per tile and stage)
// so a stage is two straight 32 KiB LDS-DMA copies.  LDS holds two stages (128 KiB).  A fragment of sub-step
// s+1 is read into registers while the MFMAs of sub-step s issue; the barrier of a stage sits between its two
// sub-steps, after which the stage's buffer is refilled with stage + 2.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <type_traits>
#include <vector>
#include <cmath>
#include <cstring>
#include <cstdlib>

#include "common.h"
#include "gemmf.h"

namespace yoho {

struct Frags {
    uintx4 ah[4], al[4], bh[4], bl[4];
};

// one fragment read of sub-step SUB from the stage at LDS address pa / pb (lane offsets already applied).
// Read order = order of use: A lo, B hi (first 16 products), then A hi, B lo.
template <int SUB, int R>
__device__ __forceinline__ void read_frag(const char* pa, const char* pb, Frags& f) {
    constexpr int t = R & 3;
    if constexpr (R < 4) f.al[t] = *reinterpret_cast<const uintx4*>(pa + (2 + SUB) * 8192 + t * 512);
    else if constexpr (R < 8) f.bh[t] = *reinterpret_cast<const uintx4*>(pb + (0 + SUB) * 8192 + t * 512);
    else if constexpr (R < 12) f.ah[t] = *reinterpret_cast<const uintx4*>(pa + (0 + SUB) * 8192 + t * 512);
    else f.bl[t] = *reinterpret_cast<const uintx4*>(pb + (2 + SUB) * 8192 + t * 512);
}

template <int SUB>
__device__ __forceinline__ void read_frags(const char* pa, const char* pb, Frags& f) {
    sfor<0, 16>([&](auto rc) { read_frag<SUB, decltype(rc)::value>(pa, pb, f); });
}

// 1 KiB of the LDS DMA of a stage: unit U of 16 per wave (8 of the A tile, 8 of the B tile).  srcA / srcB are wave
// uniform (scalar base + per-lane 32-bit offset addressing, no 64-bit vector adds between the MFMAs)
template <int U>
__device__ __forceinline__ void stage_unit(const char* srcA, const char* srcB, char* dst, int w, int lane16) {
    constexpr int p = U & 7;
    const int blk = p * 4 + w;
    const char* src = (U < 8 ? srcA : srcB) + blk * 1024;
    char* d = dst + (U < 8 ? 0 : FG_STAGE) + blk * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)(src + lane16), (lptr_t)d, 16, 0, 0);
}

__device__ __forceinline__ void stage_tile(const char* src, char* dst, int w, int lane) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int blk = p * 4 + w;                     // 32 x 1 KiB
        __builtin_amdgcn_global_load_lds((gptr_t)(src + blk * 1024 + lane * 16), (lptr_t)(dst + blk * 1024), 16, 0, 0);
    }
}

// One K16 sub-step: 48 MFMAs on fragment set `f` (every accumulator gets lo*hi, hi*lo, hi*hi, 16 MFMAs apart), with
// the 16 fragment reads of the next sub-step (and, if DMA, the 16 LDS-DMA units of a later stage) issued one per MFMA
// behind the first 16.  sched_barrier pins the order: left alone the scheduler puts dependent MFMAs back to back.
template <int RSUB, bool DMA>
__device__ __forceinline__ void substep(const Frags& f, floatx16 (&acc)[4][4], const char* ra, const char* rb, Frags& nf,
                                        const char* srcA, const char* srcB, char* dmadst, int w, int lane16) {
    sfor<0, 16>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        acc[g >> 2][g & 3] = mfma_h(f.al[g >> 2], f.bh[g & 3], acc[g >> 2][g & 3]);
        read_frag<RSUB, g>(ra, rb, nf);
        __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, 16>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        acc[g >> 2][g & 3] = mfma_h(f.ah[g >> 2], f.bl[g & 3], acc[g >> 2][g & 3]);
        if constexpr (DMA) {
            stage_unit<g>(srcA, srcB, dmadst, w, lane16);
            __builtin_amdgcn_sched_barrier(0);
        }
    });
    __builtin_amdgcn_sched_barrier(0);
    sfor<0, 16>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        acc[g >> 2][g & 3] = mfma_h(f.ah[g >> 2], f.bh[g & 3], acc[g >> 2][g & 3]);
    });
    __builtin_amdgcn_sched_barrier(0);
}

__global__ __launch_bounds__(256, 1) void fgemm_kernel(FGemmArgs a, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Work map.  Workgroup b runs on XCD b & 7.  A column tile (irrep t, ntile) and its MT[t] row tiles stay on one XCD
    // (the B panel is then read from HBM once and served from that XCD's L2 to the other row tiles); the column tiles
    // of every irrep are dealt round-robin over the XCDs so that all eight get the same mix of long and short K loops.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int t = -1, local = 0, r = 0;
    {
        int start = 0;
#pragma unroll
        for (int u = 0; u < NIR_ORD; ++u) {
            const int ru = (xcd + a.rot[u]) & 7;
            const int cnt = a.NT[u] > ru ? ((a.NT[u] - 1 - ru) / 8 + 1) * a.MT[u] : 0;
            if (t < 0 && slot < start + cnt) { t = u; local = slot - start; r = ru; }
            start += cnt;
        }
    }
    if (t < 0) return;
    const int d = a.dim[t], qbase = a.qbase[t];
    const int MT = a.MT[t], KS = d * a.cin / 32;
    const int cg = local / MT, mtile = local - cg * MT;
    const int ntile = r + 8 * cg;
    const char* Ag = a.A + a.a_off[t] + (size_t)mtile * KS * FG_STAGE;
    const char* Bg = a.B + a.b_off[t] + (size_t)ntile * KS * FG_STAGE;
    const int wm = w >> 1, wn = w & 1;
    const int lane16 = lane * 16;

    floatx16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS: buffer u at u * 64 KiB: A tile, then B tile
    const int lane_a = (lane >> 5) * 4096 + (wm * 128 + (lane & 31)) * 16;
    const int lane_b = FG_STAGE + (lane >> 5) * 4096 + (wn * 128 + (lane & 31)) * 16;

    stage_tile(Ag, smem, w, lane);
    stage_tile(Bg, smem + FG_STAGE, w, lane);
    if (KS > 1) {
        stage_tile(Ag + FG_STAGE, smem + 2 * FG_STAGE, w, lane);
        stage_tile(Bg + FG_STAGE, smem + 3 * FG_STAGE, w, lane);
    }
    if ((flags & EPI_RES) && (mtile * 256 + wm * 128) < d * a.cout) {
        // The accumulators start from the residual (scaled by 1 / descale, a power of two): its 64 loads per wave are in
        // flight together with the first two DMA stages.  Added in the epilogue instead they are serialised behind the
        // accumulators' registers (no room to prefetch) and cost a third of a millisecond per pass.
        const float inv = 1.f / a.descale;
        const int half = lane >> 5, kp32 = lane & 31, cout8 = a.cout >> 3;
        const int col0 = ntile * 256 + wn * 128;
        const int jidx = col0 / a.kppad, kp0 = col0 - jidx * a.kppad;
#pragma unroll
        for (int bi = 0; bi < 4; ++bi) {
            const int tile32 = (kp0 >> 5) + bi;
#pragma unroll
            for (int ai = 0; ai < 4; ++ai) {
                const int rowb = mtile * 256 + wm * 128 + ai * 32;
                const int iidx = rowb / a.cout, o0 = rowb - iidx * a.cout;
                const bool ok = tile32 < a.nT32 && iidx < d;
                const int q = qbase + iidx * d + jidx;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int o = o0 + q4 * 8 + half * 4;
                    const size_t off = (((((size_t)tile32 * G + q) * cout8 + (o >> 3)) * 2 + half) * TILE + kp32) * 4;
                    // branch-free, so that all loads go out back to back
                    const floatx4 v = *reinterpret_cast<const floatx4*>(a.res + (ok ? off : 0)) * (ok ? inv : 0.f);
                    acc[ai][bi][4 * q4 + 0] = v.x; acc[ai][bi][4 * q4 + 1] = v.y;
                    acc[ai][bi][4 * q4 + 2] = v.z; acc[ai][bi][4 * q4 + 3] = v.w;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if ((mtile * 256 + wm * 128) >= d * a.cout) {
        // all 128 rows of this wave are padding (small cout): only keep the LDS DMA and the barriers going
        for (int s = 0; s < KS; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int s2 = s + 2 < KS ? s + 2 : KS - 1;
            char* cur = smem + (s & 1) * (2 * FG_STAGE);
            stage_tile(Ag + (size_t)s2 * FG_STAGE, cur, w, lane);
            stage_tile(Bg + (size_t)s2 * FG_STAGE, cur + FG_STAGE, w, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    Frags P, Q;
    read_frags<0>(smem + lane_a, smem + lane_b, P);
    for (int s = 0; s < KS; ++s) {
        char* cur = smem + (s & 1) * (2 * FG_STAGE);
        char* nxt = smem + ((s + 1) & 1) * (2 * FG_STAGE);
        // sub-step 0: MFMAs on P, fragments of sub-step 1 into Q
        substep<1, false>(P, acc, cur + lane_a, cur + lane_b, Q, nullptr, nullptr, nullptr, w, lane16);
        // everybody has its sub-step-1 fragments in registers: the buffer is free, and stage s+1 has landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        // sub-step 1: MFMAs on Q; refill the freed buffer with stage s+2 and read the first fragments of stage s+1 into P.
        // Past the end both are repeated on the last stage (branch-free, the data is not used).
        const int s2 = s + 2 < KS ? s + 2 : KS - 1;
        substep<0, true>(Q, acc, nxt + lane_a, nxt + lane_b, P, uniform_ptr(Ag + (size_t)s2 * FG_STAGE), uniform_ptr(Bg + (size_t)s2 * FG_STAGE),
                              cur, w, lane16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue: D[row][col]: lane (col = lane & 31, half = lane >> 5), reg r -> row = (r & 3) + 8 * (r >> 2) + 4 * half
    const int half = lane >> 5, kp32 = lane & 31;
    const int cout8 = a.cout >> 3;
    const int col0 = ntile * 256 + wn * 128;
    const int jidx = col0 / a.kppad, kp0 = col0 - jidx * a.kppad;
    const bool addb = (d == 1);                              // trivial irrep: coefficient 0 carries sqrt(60) * bias
    unsigned top = 0u;                                       // largest |coefficient| written (bit pattern; inf / NaN order above)
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
        const int tile32 = (kp0 >> 5) + bi;
        if (tile32 >= a.nT32) continue;
#pragma unroll
        for (int ai = 0; ai < 4; ++ai) {
            // the 32 rows of an MFMA tile share i (cout is a multiple of 32)
            const int rowb = mtile * 256 + wm * 128 + ai * 32;
            const int iidx = rowb / a.cout, o0 = rowb - iidx * a.cout;
            if (iidx >= d) continue;
            const int q = qbase + iidx * d + jidx;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int o = o0 + q4 * 8 + half * 4;
                floatx4 val;
                val.x = acc[ai][bi][4 * q4 + 0]; val.y = acc[ai][bi][4 * q4 + 1];
                val.z = acc[ai][bi][4 * q4 + 2]; val.w = acc[ai][bi][4 * q4 + 3];
                val *= a.descale;
                if (addb) val += *reinterpret_cast<const floatx4*>(a.bias + o) * 7.745966692414834f;
                const size_t off = (((((size_t)tile32 * G + q) * cout8 + (o >> 3)) * 2 + half) * TILE + kp32) * 4;
                *reinterpret_cast<floatx4*>(a.out + off) = val;
                top = max(max(top, __float_as_uint(val.x) & 0x7FFFFFFFu), __float_as_uint(val.y) & 0x7FFFFFFFu);
                top = max(max(top, __float_as_uint(val.z) & 0x7FFFFFFFu), __float_as_uint(val.w) & 0x7FFFFFFFu);
            }
        }
    }
    // the transform kernel that reads these coefficients multiplies them by HF_ASCALE and converts to fp16
    note_range_bits(a.rflag, top, FP16_MAX / HF_ASCALE);
}

int fgemm_init() {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, FG_LDS));
    if (int rc = cgemm_init()) return rc;
    return fgemm2_init();
}

// byte offset of the t-th (launch order) irrep inside an operand pack with `rows` rows/cols per irrep-dimension unit
// and kdim = channels of the contracted side
static long long pack_off(int t, int rows, int kdim) {
    long long off = 0;
    for (int u = 0; u < t; ++u) off += (long long)((FG_ORD_D[u] * rows + 255) / 256) * (FG_ORD_D[u] * kdim / 32) * FG_STAGE;
    return off;
}

size_t fgemm_planes_bytes(int kppad, int cin) { return (size_t)pack_off(NIR_ORD, kppad, cin); }

// B-operand addressing for the transform kernel: coefficient q = (r, m, j) -> irrep slot
void fgemm_qinfo(int* qi /* [60][4]: t, m, j, d */) {
    for (int t = 0; t < NIR_ORD; ++t) {
        const int r = FG_ORD_R[t], d = FG_IR_D[r];
        for (int m = 0; m < d; ++m)
            for (int j = 0; j < d; ++j) {
                int* p = qi + (FG_IR_BASE[r] + m * d + j) * 4;
                p[0] = t; p[1] = m; p[2] = j; p[3] = d;
            }
    }
}
void fgemm_plane_offsets(int kppad, int cin, long long* off /* [NIR_ORD] */) {
    for (int t = 0; t < NIR_ORD; ++t) off[t] = pack_off(t, kppad, cin);
}

static inline unsigned short half_bits_h(float x) {
    const _Float16 h = (_Float16)x;
    unsigned short u;
    std::memcpy(&u, &h, 2);
    return u;
}

// OCP fp8 E4M3 (bias 7, largest finite 448, no infinities) of x: round to nearest even, saturating
static unsigned char e4m3_bits(float x) {
    const unsigned sgn = std::signbit(x) ? 0x80u : 0u;
    const float a = std::fabs(x);
    if (!(a == a)) return 0x7F;
    if (a >= 464.f) return (unsigned char)(sgn | 0x7E);                       // 448 (464 is half way to the next step)
    if (a < std::ldexp(1.f, -6)) {                                            // subnormal: quantum 2^-9
        const int r = (int)std::nearbyint(std::ldexp(a, 9));                  // 0 .. 8 (8 = the smallest normal)
        return (unsigned char)(sgn | (r == 8 ? 0x08 : r));
    }
    int e;
    (void)std::frexp(a, &e);                                                  // a = m 2^e, 0.5 <= m < 1
    int r = (int)std::nearbyint(std::ldexp(a, 3 - (e - 1)));                  // a / 2^(e-1) in [1, 2) -> 8 .. 16
    if (r == 16) { r = 8; ++e; }
    const int E = e - 1 + 7;
    if (E > 15 || (E == 15 && r - 8 > 6)) return (unsigned char)(sgn | 0x7E);
    return (unsigned char)(sgn | (E << 3) | (r - 8));
}
static inline float half_value_h(unsigned short u) {
    _Float16 h;
    std::memcpy(&h, &u, 2);
    return (float)h;
}

// What(r,i,m)[o][c] = sum_k W[o][c][k] rho_r(n_k)[m][i]  ->  A pack (fp16x2 planes of What * 2^s)
// out8 (optional): the pack fgemm3c reads - the hi plane as in `out`; every 16-byte unit of the lo plane (the 8 K-values of one row and
// k-group) replaced by the unit's fp8 operand: bytes 0-7 = e4m3(hi / 4), bytes 8-15 = e4m3(lo * 512)   (|hi| < 1024, |lo| <= 0.5)
int pack_fgemm_weights(const FourierBasis& fb, const float* W, int cin, int cout, std::vector<unsigned short>& out, float* descale,
                       std::vector<unsigned short>* out8) {
    if ((cout % 256 && cout != 32) || cin % 32) return -1;     // row tiles must not straddle i unless the whole irrep fits one tile
    std::vector<float> what((size_t)60 * cout * cin);
    std::vector<double> coef(60 * NTAP);
    for (int r = 0; r < 5; ++r) {
        const int d = FG_IR_D[r];
        for (int i = 0; i < d; ++i)
            for (int m = 0; m < d; ++m)
                for (int k = 0; k < NTAP; ++k) coef[(FG_IR_BASE[r] + i * d + m) * NTAP + k] = fb.rho[r][fb.n0[k]][m * d + i];
    }
    float wmax = 0.f;
    for (int f = 0; f < 60; ++f)
        for (int o = 0; o < cout; ++o)
            for (int c = 0; c < cin; ++c) {
                const float* wk = W + ((size_t)o * cin + c) * NTAP;
                double acc = 0.0;
                for (int k = 0; k < NTAP; ++k) acc += (double)wk[k] * coef[f * NTAP + k];
                const float v = (float)acc;
                what[((size_t)f * cout + o) * cin + c] = v;
                wmax = std::fmax(wmax, std::fabs(v));
            }
    int ex = 0;
    if (wmax > 0.f && std::isfinite(wmax)) (void)std::frexp(wmax, &ex);
    const float wscale = std::ldexp(1.f, 10 - ex);                 // max |What| * wscale in [2^9, 2^10)
    *descale = 1.f / (wscale * HF_ASCALE);
    out.assign((size_t)pack_off(NIR_ORD, cout, cin) / 2, 0);
    for (int t = 0; t < NIR_ORD; ++t) {
        const int r = FG_ORD_R[t], d = FG_IR_D[r];
        const int KS = d * cin / 32;
        unsigned short* base = out.data() + pack_off(t, cout, cin) / 2;
        for (int i = 0; i < d; ++i)
            for (int o = 0; o < cout; ++o) {
                const int row = i * cout + o, mtile = row >> 8, rr = row & 255;
                for (int m = 0; m < d; ++m) {
                    const float* src = &what[((size_t)(FG_IR_BASE[r] + i * d + m) * cout + o) * cin];
                    for (int c = 0; c < cin; ++c) {
                        const int k = m * cin + c, ks = k >> 5, sub = (k >> 4) & 1, kg = (k >> 3) & 1, e = k & 7;
                        const float x = src[c] * wscale;
                        const _Float16 hi = (_Float16)x;
                        unsigned short* blk = base + ((size_t)mtile * KS + ks) * (FG_STAGE / 2);
                        const size_t idx = ((size_t)(sub * 2 + kg) * 256 + rr) * 8 + e;      // within a plane (8192 halfs per plane-substep pair)
                        blk[0 * 8192 + idx] = half_bits_h(x);
                        blk[1 * 8192 + idx] = half_bits_h(x - (float)hi);
                    }
                }
            }
    }
    if (out8) {
        *out8 = out;
        const size_t nblk = out.size() / (FG_STAGE / 2);
        for (size_t b = 0; b < nblk; ++b) {
            const unsigned short* src = out.data() + b * (FG_STAGE / 2);
            unsigned char* dst = reinterpret_cast<unsigned char*>(out8->data() + b * (FG_STAGE / 2) + 8192);
            for (int u = 0; u < 1024; ++u)                                     // 16-byte units of the plane
                for (int e = 0; e < 8; ++e) {
                    dst[u * 16 + e] = e4m3_bits(half_value_h(src[u * 8 + e]) * 0.25f);
                    dst[u * 16 + 8 + e] = e4m3_bits(half_value_h(src[8192 + u * 8 + e]) * 512.f);
                }
        }
    }
    return 0;
}

// the 13-tap weights W[o][c][k] of a direct layer as the plain matrix A[o][k * cin + c] in the same A pack (row tiles of 256 output
// channels x K32 stages; planes of W * 2^s with max |W| 2^s in [2^9, 2^10)) - the A operand of the cone GEMM (gemmf2.hip); out8 as above
int pack_cgemm_weights(const float* W, int cin, int cout, int ntaps, std::vector<unsigned short>& out, float* descale, std::vector<unsigned short>& out8) {
    if (cout % 256 || cin % 32) return -1;
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)cout * cin * ntaps; ++i) wmax = std::fmax(wmax, std::fabs(W[i]));
    int ex = 0;
    if (wmax > 0.f && std::isfinite(wmax)) (void)std::frexp(wmax, &ex);
    const float wscale = std::ldexp(1.f, 10 - ex);
    *descale = 1.f / (wscale * H2_ASCALE);
    const int KS = ntaps * cin / 32;
    out.assign((size_t)(cout / 256) * KS * (FG_STAGE / 2), 0);
    for (int o = 0; o < cout; ++o) {
        const int mtile = o >> 8, rr = o & 255;
        for (int tap = 0; tap < ntaps; ++tap)
            for (int c = 0; c < cin; ++c) {
                const int k = tap * cin + c, ks = k >> 5, sub = (k >> 4) & 1, kg = (k >> 3) & 1, e = k & 7;
                const float x = W[((size_t)o * cin + c) * ntaps + tap] * wscale;
                const _Float16 hi = (_Float16)x;
                unsigned short* blk = out.data() + ((size_t)mtile * KS + ks) * (FG_STAGE / 2);
                const size_t idx = ((size_t)(sub * 2 + kg) * 256 + rr) * 8 + e;
                blk[0 * 8192 + idx] = half_bits_h(x);
                blk[1 * 8192 + idx] = half_bits_h(x - (float)hi);
            }
    }
    out8 = out;
    const size_t nblk = out.size() / (FG_STAGE / 2);
    for (size_t b = 0; b < nblk; ++b) {
        const unsigned short* src = out.data() + b * (FG_STAGE / 2);
        unsigned char* dst = reinterpret_cast<unsigned char*>(out8.data() + b * (FG_STAGE / 2) + 8192);
        for (int u = 0; u < 1024; ++u)
            for (int e = 0; e < 8; ++e) {
                dst[u * 16 + e] = e4m3_bits(half_value_h(src[u * 8 + e]) * 0.25f);
                dst[u * 16 + 8 + e] = e4m3_bits(half_value_h(src[8192 + u * 8 + e]) * 512.f);
            }
    }
    return 0;
}

void fgemm_fill_args(FGemmArgs& a, const Layer& L, const char* Bplanes, int kppad, int nT32, const float* res, float* out, int* rflag) {
    a.rflag = rflag; a.amax = nullptr;
    a.A = reinterpret_cast<const char*>(L.wpg); a.B = Bplanes; a.bias = L.bias; a.res = res; a.out = out;
    a.cin = L.cin; a.cout = L.cout; a.kppad = kppad; a.nT32 = nT32; a.descale = L.wpg_descale;
    static const int ROT[NIR_ORD] = {0, 0, 4, 4, 2};
    for (int t = 0; t < NIR_ORD; ++t) {
        a.a_off[t] = pack_off(t, L.cout, L.cin);
        a.b_off[t] = pack_off(t, kppad, L.cin);
        a.MT[t] = (FG_ORD_D[t] * L.cout + 255) / 256;
        a.NT[t] = FG_ORD_D[t] * kppad / 256;
        a.rot[t] = ROT[t];
        a.dim[t] = FG_ORD_D[t];
        a.qbase[t] = FG_IR_BASE[FG_ORD_R[t]];
    }
}

// variant 2 (default): 256 x 128 tiles, two workgroups per CU (gemmf2.hip); variant 1: 256 x 256 tiles, one workgroup per CU
int launch_fgemm(const Layer& L, const char* Bplanes, int kppad, int nT32, const float* res, float* out, int flags, hipStream_t s, int* rflag,
                 int variant, const unsigned* amax) {
    FGemmArgs a;
    fgemm_fill_args(a, L, Bplanes, kppad, nT32, res, out, rflag);
    static const int dbg_gemm1 = [] { const char* e = experiment_env("YOHO_PARTI_DEBUG"); return (e && std::strstr(e, "gemm1")) ? 1 : 0; }();
    if (dbg_gemm1) variant = 1;
    // fgemm3c and its fp8 weight pack - chosen AFTER every override of the variant: the other kernels read A as fp16 planes
    if (variant == 3 && amax && L.wpg8) { a.amax = amax; a.A = reinterpret_cast<const char*>(L.wpg8); }
    if (variant == 3) return launch_fgemm3(a, flags, s);
    if (variant != 1) return launch_fgemm2(a, flags, s);
    int tot = 0;
    for (int x = 0; x < 8; ++x) {
        int n = 0;
        for (int t = 0; t < NIR_ORD; ++t) {
            const int r = (x + a.rot[t]) & 7;
            if (a.NT[t] > r) n += ((a.NT[t] - 1 - r) / 8 + 1) * a.MT[t];
        }
        tot = n > tot ? n : tot;
    }
    tot *= 8;
    hipLaunchKernelGGL(fgemm_kernel, dim3(tot), dim3(256), FG_LDS, s, a, flags);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
