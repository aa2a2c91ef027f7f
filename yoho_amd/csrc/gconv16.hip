// Icosahedral group convolution on bf16 MFMA with an fp32-accurate 3-way operand split, gfx950.
//
// Same contraction as gconv.hip (utils/network.py:46-52,12-21,23-65):
//     out[b, o, g] = bias[o] + sum_k sum_c W[o, c, 0, k] * act[b, c, N[g, k]]
// but every fp32 operand x is carried as three bf16 planes x = x_h + x_m + x_l (24 mantissa bits,
// exact to 2^-25 |x|) and a product is evaluated as the six largest cross terms
//     a_h b_h + a_h b_m + a_m b_h + a_h b_l + a_l b_h + a_m b_m        (dropped terms <= 2^-24 |ab|)
// on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  gfx950 has no TF32; fp32-input MFMA runs at 1/16 of
// the bf16 rate, so six bf16 MFMAs are 2.7x cheaper than the fp32 MFMAs they replace, at fp32-level
// accuracy (measured against the fp32 kernel and the oracle in tests/test_gpu_kernels.py).
//
// Shape of one MFMA:  D(32 o x 32 cols) += Wfrag(32 o x 16 k) * X(16 k x 32 cols)
//   cols = 2 output group elements (a "unit" = g-pair) x 16 keypoints (a tile is 16 keypoints),
//   k    = 2 taps (a "tap pair") x 8 channels (one c8 chunk): lanes 0-31 carry tap t0, lanes 32-63 tap t1.
// so the B operand of lane (col j, half h) is 16 contiguous bytes of slab N[g(j>>4)][tap(h)] at keypoint
// j&15 - four wave-uniform slab offsets per (unit, tap pair), selected per lane; no gather through memory.
// The 13 taps make 7 tap pairs, the last one half empty (zero weights): 14/13 issue overhead.
//
// Workgroup = 4 waves, 1 per SIMD, 1 workgroup per CU.  LDS: two 45 KiB buffers, each one c8 chunk =
// 3 planes x 60 slabs x (16 kp x 8 ch) bf16, filled by global_load_lds DMA one chunk ahead.
//   NOB = 2: waves {0,1} / {2,3} own two 32-channel output blocks, each wave 15 of the 30 units;
//   NOB = 1: the four waves own 8/8/7/7 units of one output block (Cout = 32 layer).
#include "common.h"
#include <type_traits>

namespace yoho {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));

constexpr int T16 = 16;                               // keypoints per tile
constexpr int SLAB16_BYTES = T16 * 8 * 2;             // 256 B: one group element, 8 channels, one plane
constexpr int PLANE16_BYTES = G * SLAB16_BYTES;       // 15360
// NPL = number of 16-bit planes per fp32 operand:
//   3: bf16 x = h + m + l, six cross products  (any fp32 range)
//   2: fp16 x = h + l,     three cross products (|x| within fp16 range; error <= 3 * 2^-22 per product)
constexpr int chunk16_bytes(int npl) { return npl * PLANE16_BYTES; }       // 46080 / 30720
constexpr int lds16_total(int npl) { return 2 * chunk16_bytes(npl) + 4 * 8 * 15 * 4 * 4; }   // chunks + per-wave slab-offset tables
constexpr int NTP = 7;                                // tap pairs
constexpr int NUNIT = 32;                             // unit slots per configuration (30 used for 60 g)

// c_slab4[cfg][tp][unit] = the four slab indices N[ga][t0], N[ga][t1], N[gb][t0], N[gb][t1] packed one per byte
//                           (a lane picks byte 2*gsel + h with one v_bfe_u32; no divergent control flow)
// c_unitg[cfg][unit][2]  = output group elements (ga, gb) of the unit, -1 = unused
// cfg: 0 = all 60 group elements (30 units), 1 = the 45-element 2-hop cone of element 0 (23 units),
//      2 = its 13-element 1-hop cone (7 units)   [PartII only needs group element 0 of its last feature map]
constexpr int NCFG16 = 3;
int upload_slot_tables16(const int* slab4_h, const int* unitg_h, SlotTables& t) {
    HIPCHK(hipMalloc((void**)&t.slab4, sizeof(int) * NCFG16 * NTP * NUNIT));
    HIPCHK(hipMalloc((void**)&t.unitg, sizeof(int) * NCFG16 * NUNIT * 2));
    HIPCHK(hipMemcpy(t.slab4, slab4_h, sizeof(int) * NCFG16 * NTP * NUNIT, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(t.unitg, unitg_h, sizeof(int) * NCFG16 * NUNIT * 2, hipMemcpyHostToDevice));
    return 0;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NPL>
__device__ __forceinline__ void stage_chunk16(const char* src, char* dst, int w, int lane) {
    for (int p = w; p < chunk16_bytes(NPL) / 1024; p += 4) {
        const char* s = src + p * 1024 + lane * 16;
        char* d = dst + p * 1024;                      // wave-uniform; hardware adds lane*16
        __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)d, 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 as_bf16x8(uintx4 v) {
    union { uintx4 u; bf16x8 b; } c;
    c.u = v;
    return c.b;
}

// fp32 -> (hi, mid, lo) bf16 bit patterns, round-to-nearest-even at every step
__device__ __forceinline__ unsigned bf16_rne_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_rne_bits(x);
    const float r1 = x - __uint_as_float(h << 16);
    m = bf16_rne_bits(r1);
    const float r2 = r1 - __uint_as_float(m << 16);
    l = bf16_rne_bits(r2);
}
// fp32 -> (hi, lo) fp16 bit patterns (round-to-nearest-even conversions)
__device__ __forceinline__ unsigned half_bits(float x) {
    const _Float16 h = (_Float16)x;
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
__device__ __forceinline__ void split2h(float x, unsigned& h, unsigned& l) {
    const _Float16 hh = (_Float16)x;
    const float r = x - (float)hh;
    unsigned short u;
    __builtin_memcpy(&u, &hh, 2);
    h = u;
    l = half_bits(r);
}
// planes of one fp32 value, plane order = LDS plane order (0 = hi)
// amax: largest fp16-bound magnitude converted (fp16 range guard, common.h; bf16 planes have the fp32 exponent range)
template <int NPL>
__device__ __forceinline__ void split_planes(float x, unsigned (&p)[3], float& amax) {
    if constexpr (NPL == 3) split3(x, p[0], p[1], p[2]);
    else {
        const float xs = x * H2_ASCALE;                          // exact power-of-two scaling keeps the lo plane out of the fp16 subnormals
        amax = fmaxf(amax, fabsf(xs));
        split2h(xs, p[0], p[1]); p[2] = 0;
    }
}
// D += A(plane pa) * B(plane pb) on the MFMA of the plane type
template <int NPL>
__device__ __forceinline__ floatx16 mfma16(uintx4 a, uintx4 b, floatx16 c) {
    if constexpr (NPL == 3) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
    } else {
        union { uintx4 u; halfx8 h; } ca, cb;
        ca.u = a; cb.u = b;
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(ca.h, cb.h, c, 0, 0, 0);
    }
}

template <int I, int N, typename Fn>
__device__ __forceinline__ void static_for(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int NM, int NR>
__device__ __forceinline__ void pin_interleave() {
    // scheduling directive only: NM MFMAs of this step with NR LDS reads of the next step spread between them
    if constexpr (NR == 0) {
        __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    } else {
        // reads go out behind the first NR MFMAs, so the last one has (NM - NR) MFMAs (>= 190 cycles) to land
        // before the step boundary, where hipcc waits with lgkmcnt(0)
        static_for<0, (NR < NM ? NR : NM)>([](auto) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        if constexpr (NR > NM) __builtin_amdgcn_sched_group_barrier(0x100, NR - NM, 0);
        if constexpr (NM > NR) __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
    }
}

struct Conv16Args {
    const char* X;         // activated input planes: [tile][cin8][3 planes][60][16 kp][8 ch] bf16
    const char* Wp;        // [ob][cin8][tp 7][3 planes][64 lanes][8] bf16
    const float* bias;
    const float* bn_s;
    const float* bn_t;
    const float* res;      // raw fp32 residual [tile][cout8][60][16 kp][8 ch]
    float* out_raw;        // same layout (EPI_RAW)
    char* out_act;         // plane layout of the next layer (EPI_ACT)
    float* out_raw32;      // fp32, 32-keypoint tile layout of gconv.hip (EPI_RAW32): hand-over to the fp32 kernels
    float* out_act32;      // same layout, relu(v*s + t) (EPI_ACT32)
    int nTiles, cin8, cout8, nOBgrid, cfg;
    float descale;         // fp16x2: 1 / (weight scale * H2_ASCALE), a power of two; bf16x3: unused
    int* rflag;            // fp16 range flag (note_range)
    const int* slab4;      // SlotTables::slab4 / unitg of the context
    const int* unitg;
};

template <int UPW, int NOB, int NPL>
__global__ __launch_bounds__(256, 1) void gconv16_kernel(Conv16Args a, int flags) {
    constexpr int CHB = chunk16_bytes(NPL);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7, slot8 = b >> 3;
    const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot8;
    const int tile = v / a.nOBgrid;
    const int obk = v - tile * a.nOBgrid;
    // NOB = 2: waves {0,1}/{2,3} own two o-blocks and split the unit slots in halves; NOB = 1: one o-block, the four
    // waves split the unit slots; NOB = 4: every wave owns an o-block of its own and all unit slots
    const int ob = NOB == 2 ? obk * 2 + (w >> 1) : (NOB == 4 ? obk * 4 + w : obk);
    const int ubase = NOB == 2 ? (w & 1) * UPW : (NOB == 4 ? 0 : w * UPW);         // first unit slot of this wave
    const int cfg = a.cfg;

    floatx16 acc[UPW];
#pragma unroll
    for (int j = 0; j < UPW; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    const char* Xt = a.X + (size_t)tile * a.cin8 * CHB;
    const uintx4* Wb = reinterpret_cast<const uintx4*>(a.Wp) + (size_t)ob * a.cin8 * NTP * (64 * NPL) + lane;   // NPL planes x 64 lanes per step

    const int laneoff = (lane & 15) * 16;

    // Per-lane LDS byte offsets of the B slabs (slab index * 256 B) for every (tap pair, unit) of this wave live
    // in a small LDS table [tp 0..7][unit][lane class 4] (tp 7 repeats tp 0 so that "next" never wraps).  They
    // are read one tap pair ahead with ds_read_b32 inside the MFMA pipeline: constant-memory (SMEM) loads
    // share the LGKM counter with LDS reads but return out of order, which would force full lgkmcnt(0) drains.
    int* tab = reinterpret_cast<int*>(smem + 2 * CHB) + w * (8 * UPW * 4);
    for (int i = lane; i < 8 * UPW * 4; i += 64) {
        const int cls = i & 3, j = (i >> 2) % UPW, tp = (i >> 2) / UPW;
        const unsigned packed = (unsigned)a.slab4[cfg * (NTP * NUNIT) + (tp == NTP ? 0 : tp) * NUNIT + ubase + j];
        tab[i] = (int)(((packed >> (8 * cls)) & 0xFFu) << 8);
    }
    // lane class: 2 * (column belongs to the unit's second group element) + (second tap of the pair)
    const int cls = 2 * ((lane >> 4) & 1) + (lane >> 5);
    const int* tabl = tab + cls;

    stage_chunk16<NPL>(Xt, smem, w, lane);
    // A planes (weights) run a whole channel chunk ahead: ring slot tp holds tap pair tp of the current chunk and is
    // refilled with the next chunk's right after use, 7 steps (~2 us of MFMAs) before it is needed - an L2 / MALL round trip
    uintx4 wr[NTP][NPL];
#pragma unroll
    for (int tp = 0; tp < NTP; ++tp)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) wr[tp][pl] = Wb[(size_t)tp * (64 * NPL) + 64 * pl];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int soff[UPW];
#pragma unroll
    for (int j = 0; j < UPW; ++j) soff[j] = tabl[j * 4];

    for (int c8 = 0; c8 < a.cin8; ++c8) {
        const bool more = c8 + 1 < a.cin8;
        if (more) stage_chunk16<NPL>(Xt + (size_t)(c8 + 1) * CHB, smem + ((c8 + 1) & 1) * CHB, w, lane);
        const char* xb = smem + (c8 & 1) * CHB + laneoff;
        static_for<0, NTP>([&](auto tpc) {
            constexpr int tp = decltype(tpc)::value;
            uintx4 wa[3];                                            // A planes of this tap pair: [0] = hi .. [NPL-1] = lo
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) wa[pl] = wr[tp][pl];
            if (more) {
                const uintx4* wp = Wb + ((size_t)(c8 + 1) * NTP + tp) * (64 * NPL);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) wr[tp][pl] = wp[64 * pl];
            }
            const int* tabn = tabl + (tp + 1) * (UPW * 4);          // offsets of the next tap pair (tp 7 == tp 0)
            int soffn[UPW];
            uintx4 bp[2][2][3];                                      // [buffer][unit of the pair][plane]
            // LDS reads are issued in the order the MFMAs consume them: hi planes of both units first, then the lower ones
            auto fetch = [&](int j, int buf, bool two) {
                const char* p0 = xb + soff[j];
                const char* p1 = two ? xb + soff[j + 1] : p0;
                bp[buf][0][0] = *reinterpret_cast<const uintx4*>(p0);
                if (two) bp[buf][1][0] = *reinterpret_cast<const uintx4*>(p1);
                bp[buf][0][NPL - 1] = *reinterpret_cast<const uintx4*>(p0 + (NPL - 1) * PLANE16_BYTES);
                if (two) bp[buf][1][NPL - 1] = *reinterpret_cast<const uintx4*>(p1 + (NPL - 1) * PLANE16_BYTES);
                if constexpr (NPL == 3) {
                    bp[buf][0][1] = *reinterpret_cast<const uintx4*>(p0 + PLANE16_BYTES);
                    if (two) bp[buf][1][1] = *reinterpret_cast<const uintx4*>(p1 + PLANE16_BYTES);
                }
            };
            fetch(0, 0, UPW > 1);
            __builtin_amdgcn_sched_group_barrier(0x100, (UPW > 1 ? 2 : 1) * NPL, 0);      // first pair's planes are read before the pipeline starts
            // Units are processed two at a time with their MFMAs interleaved (each unit's products are a dependent
            // chain on one accumulator).  The planes of the next pair and the slab offsets of the next tap pair are
            // read from LDS behind the first MFMAs of the current pair.  Smallest terms are accumulated first.
            static_for<0, (UPW + 1) / 2>([&](auto pc) {
                constexpr int j = decltype(pc)::value * 2;
                constexpr int cur = (j >> 1) & 1, nxt = cur ^ 1;
                constexpr bool two = j + 1 < UPW;
                if constexpr (j + 2 < UPW) fetch(j + 2, nxt, j + 3 < UPW);
                soffn[j] = tabn[j * 4];
                if constexpr (two) soffn[j + 1] = tabn[(j + 1) * 4];
                auto prod = [&](int pa, int pb) {                    // acc += A[pa] * B[pb] for both units of the pair
                    acc[j] = mfma16<NPL>(wa[pa], bp[cur][0][pb], acc[j]);
                    if constexpr (two) acc[j + 1] = mfma16<NPL>(wa[pa], bp[cur][1][pb], acc[j + 1]);
                };
                if constexpr (NPL == 3) {
                    prod(2, 0); prod(0, 2); prod(1, 1); prod(1, 0); prod(0, 1); prod(0, 0);      // l.h  h.l  m.m  m.h  h.m  h.h
                } else {
                    prod(1, 0); prod(0, 1); prod(0, 0);                                          // l.h  h.l  h.h
                }
                constexpr int NPROD = NPL == 3 ? 6 : 3;
                pin_interleave<(two ? 2 : 1) * NPROD, (j + 3 < UPW ? 2 * NPL : (j + 2 < UPW ? NPL : 0)) + (two ? 2 : 1)>();
            });
#pragma unroll
            for (int j = 0; j < UPW; ++j) soff[j] = soffn[j];
        });
        // the next chunk's DMA (issued first) must have landed; the 7 x NPL weight loads issued after it may stay in flight
        if (more) {
            if constexpr (NPL == 3) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }

    // ---- epilogue.  D[i = o][j = col]: lane (col = lane&31, half = lane>>5), reg r -> o = (r&3) + 8*(r>>2) + 4*half
    const int kp = lane & 15, half = lane >> 5, gsel = (lane >> 4) & 1;
    const int* ug = a.unitg + cfg * (NUNIT * 2) + ubase * 2;
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < UPW; ++j) {
        const int g = gsel ? ug[2 * j + 1] : ug[2 * j];
        if (g < 0) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = ob * 32 + q * 8 + half * 4;
            floatx4 val;
            val.x = acc[j][4 * q + 0]; val.y = acc[j][4 * q + 1];
            val.z = acc[j][4 * q + 2]; val.w = acc[j][4 * q + 3];
            if constexpr (NPL == 2) val *= a.descale;
            val += *reinterpret_cast<const floatx4*>(a.bias + ch);
            const size_t slab = ((size_t)tile * a.cout8 + ob * 4 + q) * G + g;           // (tile, c8, g)
            const size_t roff = slab * (T16 * 8) + kp * 8 + half * 4;                    // fp32 raw layout
            if (flags & EPI_RES) val += *reinterpret_cast<const floatx4*>(a.res + roff);
            if (flags & EPI_RAW) *reinterpret_cast<floatx4*>(a.out_raw + roff) = val;
            // 32-keypoint tile layout [tile32][c8][g][h][kp32][4]: tile32 = tile/2, kp32 = 16*(tile&1) + kp
            const size_t off32 = (((((size_t)(tile >> 1) * a.cout8 + ob * 4 + q) * G + g) * 2 + half) * TILE + (tile & 1) * 16 + kp) * 4;
            if (flags & EPI_RAW32) *reinterpret_cast<floatx4*>(a.out_raw32 + off32) = val;
            if (flags & (EPI_ACT | EPI_ACT32)) {
                const floatx4 s = *reinterpret_cast<const floatx4*>(a.bn_s + ch);
                const floatx4 t = *reinterpret_cast<const floatx4*>(a.bn_t + ch);
                floatx4 y = val * s + t;
                if (flags & EPI_ACT32) {
                    floatx4 r;
                    r.x = fmaxf(y.x, 0.f); r.y = fmaxf(y.y, 0.f); r.z = fmaxf(y.z, 0.f); r.w = fmaxf(y.w, 0.f);
                    *reinterpret_cast<floatx4*>(a.out_act32 + off32) = r;
                }
                if (!(flags & EPI_ACT)) continue;
                unsigned p0[3], p1[3], p2[3], p3[3];
                split_planes<NPL>(fmaxf(y.x, 0.f), p0, amax);
                split_planes<NPL>(fmaxf(y.y, 0.f), p1, amax);
                split_planes<NPL>(fmaxf(y.z, 0.f), p2, amax);
                split_planes<NPL>(fmaxf(y.w, 0.f), p3, amax);
                // plane layout: [tile][c8][plane][g][kp][8 ch], 16-bit elements
                char* base = a.out_act + ((size_t)tile * a.cout8 + ob * 4 + q) * CHB + (size_t)g * SLAB16_BYTES + kp * 16 + half * 8;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    uintx2 pk;
                    pk.x = p0[pl] | (p1[pl] << 16); pk.y = p2[pl] | (p3[pl] << 16);
                    *reinterpret_cast<uintx2*>(base + pl * PLANE16_BYTES) = pk;
                }
            }
        }
    }
    if constexpr (NPL == 2) note_range(a.rflag, amax);
}

template <int UPW, int NOB, int NPL>
static int launch16_t(const Conv16Args& a, int flags, hipStream_t s) {
    const int grid = a.nTiles * a.nOBgrid;
    hipLaunchKernelGGL((gconv16_kernel<UPW, NOB, NPL>), dim3(grid), dim3(256), lds16_total(NPL), s, a, flags);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int UPW, int NOB, int NPL>
static int init16_t() {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv16_kernel<UPW, NOB, NPL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds16_total(NPL)));
    return 0;
}

int gconv16_init() {
    int rc;
    if ((rc = init16_t<15, 2, 3>()) || (rc = init16_t<8, 1, 3>()) || (rc = init16_t<12, 2, 3>()) || (rc = init16_t<7, 4, 3>())) return rc;
    if ((rc = init16_t<15, 2, 2>()) || (rc = init16_t<8, 1, 2>()) || (rc = init16_t<12, 2, 2>()) || (rc = init16_t<7, 4, 2>())) return rc;
    return 0;
}

template <int NPL>
static int launch16_cfg(Conv16Args& a, const Layer& L, int cfg, int flags, hipStream_t s) {
    if (cfg == 1) {
        if (L.cout_pad % 64) { set_error("gconv16: cone-45 variant needs cout %% 64 == 0"); return YOHO_EINVAL; }
        a.nOBgrid = L.cout_pad / 64;
        return launch16_t<12, 2, NPL>(a, flags, s);
    }
    if (cfg == 2) {
        if (L.cout_pad % 128) { set_error("gconv16: cone-13 variant needs cout %% 128 == 0"); return YOHO_EINVAL; }
        a.nOBgrid = L.cout_pad / 128;
        return launch16_t<7, 4, NPL>(a, flags, s);
    }
    if (L.cout_pad % 64 == 0) { a.nOBgrid = L.cout_pad / 64; return launch16_t<15, 2, NPL>(a, flags, s); }
    a.nOBgrid = L.cout_pad / 32;
    return launch16_t<8, 1, NPL>(a, flags, s);
}

// layer launch.  npl = 3: bf16x3 planes (L.wp16), npl = 2: fp16x2 planes (L.wph).  cfg 0 (all 60 outputs): cout_pad
// multiple of 64 -> NOB = 2, else the single-block variant; cfg 1 (45 outputs, 23 units): <12,2>; cfg 2 (13 outputs): <7,4>.
int launch_gconv16(const Layer& L, const char* X, int nTiles, const float* res, float* out_raw, char* out_act, int flags, hipStream_t s,
                   int cfg, float* out_raw32, float* out_act32, int npl, int* rflag) {
    Conv16Args a;
    a.rflag = rflag;
    if (!L.tabs || !L.tabs->slab4) { set_error("launch_gconv16: the layer carries no slot tables"); return YOHO_EINVAL; }
    a.slab4 = L.tabs->slab4; a.unitg = L.tabs->unitg;
    a.X = X; a.Wp = reinterpret_cast<const char*>(npl == 2 ? L.wph : L.wp16); a.bias = L.bias; a.bn_s = L.bn_s; a.bn_t = L.bn_t;
    a.res = res; a.out_raw = out_raw; a.out_act = out_act; a.out_raw32 = out_raw32; a.out_act32 = out_act32;
    a.nTiles = nTiles; a.cin8 = L.cin / 8; a.cout8 = L.cout_pad / 8; a.cfg = cfg; a.nOBgrid = 1;
    a.descale = L.wph_descale;
    return npl == 2 ? launch16_cfg<2>(a, L, cfg, flags, s) : launch16_cfg<3>(a, L, cfg, flags, s);
}

// ---------------------------------------------------------------------------------------------
// head / tail for the 16-keypoint tile layouts
// ---------------------------------------------------------------------------------------------
// x (B,32,60) f32 -> planes [tile][c8 = 4][3][60][16][8] bf16.  One workgroup per (tile, c8).
template <int NPL>
__global__ __launch_bounds__(256) void pack16_partI_kernel(const float* __restrict__ x, int B, char* __restrict__ out, int* rflag) {
    constexpr int CHB = chunk16_bytes(NPL);
    __shared__ __attribute__((aligned(16))) unsigned short lds[CHB / 2];
    const int tile = blockIdx.x >> 2, c8 = blockIdx.x & 3;
    float amax = 0.f;
    for (int i = threadIdx.x; i < T16 * 8 * G; i += 256) {
        const int kp = i / (8 * G);
        const int r = i - kp * (8 * G);
        const int cl = r / G, g = r - cl * G;
        const int bb = tile * T16 + kp;
        const float v = bb < B ? x[(size_t)bb * (F * G) + (c8 * 8 + cl) * G + g] : 0.f;
        unsigned pp[3];
        split_planes<NPL>(v, pp, amax);
        const int o = (g * T16 + kp) * 8 + cl;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) lds[pl * (PLANE16_BYTES / 2) + o] = (unsigned short)pp[pl];
    }
    __syncthreads();
    uintx4* o = reinterpret_cast<uintx4*>(out + ((size_t)tile * 4 + c8) * CHB);
    const uintx4* l = reinterpret_cast<const uintx4*>(lds);
    for (int i = threadIdx.x; i < CHB / 16; i += 256) o[i] = l[i];
    if constexpr (NPL == 2) note_range(rflag, amax);
}

int launch_pack16_partI(const float* x, int B, int nTiles, char* out, hipStream_t s, int npl, int* rflag) {
    if (npl == 2) hipLaunchKernelGGL(pack16_partI_kernel<2>, dim3(nTiles * 4), dim3(256), 0, s, x, B, out, rflag);
    else hipLaunchKernelGGL(pack16_partI_kernel<3>, dim3(nTiles * 4), dim3(256), 0, s, x, B, out, rflag);
    HIPCHK(hipGetLastError());
    return 0;
}

// PartII head for the bf16x3 path (utils/network.py:266-269 + Conv_init's BN/ReLU): permute the group axis of
// before_eqv0 / after_eqv0 by P[pre_idx], concatenate 4 x 32 channels, BN(128) + ReLU, split into bf16 planes.
// One workgroup per (tile16, c8); c8 >> 2 selects the source tensor.
template <int NPL>
__global__ __launch_bounds__(256) void pack16_partII_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                            const float* __restrict__ s2, const float* __restrict__ s3,
                                                            const int64_t* __restrict__ pre_idx, const int* __restrict__ P,
                                                            const float* __restrict__ bn_s, const float* __restrict__ bn_t,
                                                            int M, char* __restrict__ out, int* rflag) {
    constexpr int CHB = chunk16_bytes(NPL);
    __shared__ __attribute__((aligned(16))) unsigned short lds[CHB / 2];
    const int tile = blockIdx.x >> 4, c8 = blockIdx.x & 15;
    float amax = 0.f;
    const int src = c8 >> 2;
    const float* sp = src == 0 ? s0 : (src == 1 ? s1 : (src == 2 ? s2 : s3));
    const bool permute = (src == 0) || (src == 2);
    const int cbase = (c8 & 3) * 8;
    for (int i = threadIdx.x; i < T16 * 8 * G; i += 256) {
        const int kp = i / (8 * G);
        const int r = i - kp * (8 * G);
        const int cl = r / G, g = r - cl * G;
        const int m = tile * T16 + kp;
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
        unsigned pp[3];
        split_planes<NPL>(v, pp, amax);
        const int o = (g * T16 + kp) * 8 + cl;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) lds[pl * (PLANE16_BYTES / 2) + o] = (unsigned short)pp[pl];
    }
    __syncthreads();
    uintx4* o = reinterpret_cast<uintx4*>(out + ((size_t)tile * 16 + c8) * CHB);
    const uintx4* l = reinterpret_cast<const uintx4*>(lds);
    for (int i = threadIdx.x; i < CHB / 16; i += 256) o[i] = l[i];
    if constexpr (NPL == 2) note_range(rflag, amax);
}

int launch_pack16_partII(const float* s0, const float* s1, const float* s2, const float* s3, const int64_t* pre_idx, const int* P,
                         const float* bn_s, const float* bn_t, int M, int nTiles16, char* out, hipStream_t s, int npl, int* rflag) {
    if (npl == 2) hipLaunchKernelGGL(pack16_partII_kernel<2>, dim3(nTiles16 * 16), dim3(256), 0, s, s0, s1, s2, s3, pre_idx, P, bn_s, bn_t, M, out, rflag);
    else hipLaunchKernelGGL(pack16_partII_kernel<3>, dim3(nTiles16 * 16), dim3(256), 0, s, s0, s1, s2, s3, pre_idx, P, bn_s, bn_t, M, out, rflag);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
