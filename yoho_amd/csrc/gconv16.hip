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
typedef unsigned int uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned int uintx2 __attribute__((ext_vector_type(2)));

constexpr int T16 = 16;                               // keypoints per tile
constexpr int SLAB16_BYTES = T16 * 8 * 2;             // 256 B: one group element, 8 channels, one plane
constexpr int PLANE16_BYTES = G * SLAB16_BYTES;       // 15360
constexpr int CHUNK16_BYTES = 3 * PLANE16_BYTES;      // 46080
constexpr int LDS16_BYTES = 2 * CHUNK16_BYTES;        // 92160 (+ the per-wave slab-offset tables, LDS16_TOTAL)
constexpr int LDS16_TOTAL = LDS16_BYTES + 4 * 8 * 15 * 4 * 4;   // 99840
constexpr int NTP = 7;                                // tap pairs
constexpr int NUNIT = 32;                             // unit slots per configuration (30 used for 60 g)

// c_slab4[cfg][tp][unit] = the four slab indices N[ga][t0], N[ga][t1], N[gb][t0], N[gb][t1] packed one per byte
//                           (a lane picks byte 2*gsel + h with one v_bfe_u32; no divergent control flow)
// c_unitg[cfg][unit][2]  = output group elements (ga, gb) of the unit, -1 = unused
// cfg: 0 = all 60 group elements (30 units), 1 = the 45-element 2-hop cone of element 0 (23 units),
//      2 = its 13-element 1-hop cone (7 units)   [PartII only needs group element 0 of its last feature map]
constexpr int NCFG16 = 3;
__constant__ int c_slab4[NCFG16][NTP * NUNIT];
__constant__ int c_unitg[NCFG16][NUNIT * 2];

int upload_slot_tables16(const int* slab4_h, const int* unitg_h) {
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_slab4), slab4_h, sizeof(int) * NCFG16 * NTP * NUNIT));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_unitg), unitg_h, sizeof(int) * NCFG16 * NUNIT * 2));
    return 0;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void stage_chunk16(const char* src, char* dst, int w, int lane) {
    for (int p = w; p < CHUNK16_BYTES / 1024; p += 4) {
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
        static_for<0, NR>([](auto) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, NM - NR, 0);
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
};

template <int UPW, int NOB>
__global__ __launch_bounds__(256, 1) void gconv16_kernel(Conv16Args a, int flags) {
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

    const char* Xt = a.X + (size_t)tile * a.cin8 * CHUNK16_BYTES;
    const int total = a.cin8 * NTP;
    const uintx4* Wb = reinterpret_cast<const uintx4*>(a.Wp) + (size_t)ob * total * 192 + lane;   // 3 planes x 64 lanes per step

    const int laneoff = (lane & 15) * 16;

    // Per-lane LDS byte offsets of the B slabs (slab index * 256 B) for every (tap pair, unit) of this wave live
    // in a small LDS table [tp 0..7][unit][lane class 4] (tp 7 repeats tp 0 so that "next" never wraps).  They
    // are read one tap pair ahead with ds_read_b32 inside the MFMA pipeline: constant-memory (SMEM) loads
    // share the LGKM counter with LDS reads but return out of order, which would force full lgkmcnt(0) drains.
    int* tab = reinterpret_cast<int*>(smem + LDS16_BYTES) + w * (8 * UPW * 4);
    for (int i = lane; i < 8 * UPW * 4; i += 64) {
        const int cls = i & 3, j = (i >> 2) % UPW, tp = (i >> 2) / UPW;
        const unsigned packed = (unsigned)c_slab4[cfg][(tp == NTP ? 0 : tp) * NUNIT + ubase + j];
        tab[i] = (int)(((packed >> (8 * cls)) & 0xFFu) << 8);
    }
    // lane class: 2 * (column belongs to the unit's second group element) + (second tap of the pair)
    const int cls = 2 * ((lane >> 4) & 1) + (lane >> 5);
    const int* tabl = tab + cls;

    stage_chunk16(Xt, smem, w, lane);
    uintx4 wn_h = Wb[0], wn_m = Wb[64], wn_l = Wb[128];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int soff[UPW];
#pragma unroll
    for (int j = 0; j < UPW; ++j) soff[j] = tabl[j * 4];

    int it = 0;
    for (int c8 = 0; c8 < a.cin8; ++c8) {
        if (c8 + 1 < a.cin8) stage_chunk16(Xt + (size_t)(c8 + 1) * CHUNK16_BYTES, smem + ((c8 + 1) & 1) * CHUNK16_BYTES, w, lane);
        const char* xb = smem + (c8 & 1) * CHUNK16_BYTES + laneoff;
        for (int tp = 0; tp < NTP; ++tp) {
            const bf16x8 a_h = as_bf16x8(wn_h), a_m = as_bf16x8(wn_m), a_l = as_bf16x8(wn_l);
            ++it;
            if (it < total) {
                const uintx4* wp = Wb + (size_t)it * 192;
                wn_h = wp[0]; wn_m = wp[64]; wn_l = wp[128];
            }
            const int* tabn = tabl + (tp + 1) * (UPW * 4);          // offsets of the next tap pair (tp 7 == tp 0)
            int soffn[UPW];
            uintx4 bh[2][2], bm[2][2], bl[2][2];
            // LDS reads are issued in the order the MFMAs consume them: h0 h1 l0 l1 m0 m1
            auto fetch = [&](int j, int buf, bool two) {
                const char* p0 = xb + soff[j];
                const char* p1 = two ? xb + soff[j + 1] : p0;
                bh[buf][0] = *reinterpret_cast<const uintx4*>(p0);
                if (two) bh[buf][1] = *reinterpret_cast<const uintx4*>(p1);
                bl[buf][0] = *reinterpret_cast<const uintx4*>(p0 + 2 * PLANE16_BYTES);
                if (two) bl[buf][1] = *reinterpret_cast<const uintx4*>(p1 + 2 * PLANE16_BYTES);
                bm[buf][0] = *reinterpret_cast<const uintx4*>(p0 + PLANE16_BYTES);
                if (two) bm[buf][1] = *reinterpret_cast<const uintx4*>(p1 + PLANE16_BYTES);
            };
            fetch(0, 0, UPW > 1);
            __builtin_amdgcn_sched_group_barrier(0x100, UPW > 1 ? 6 : 3, 0);      // first pair's planes are read before the pipeline starts
            // Units are processed two at a time with their MFMAs interleaved (each unit's six products are a
            // dependent chain on one accumulator).  The planes of the next pair and the slab offsets of the
            // next tap pair are read from LDS behind the first MFMAs of the current pair.
            static_for<0, (UPW + 1) / 2>([&](auto pc) {
                constexpr int j = decltype(pc)::value * 2;
                constexpr int cur = (j >> 1) & 1, nxt = cur ^ 1;
                constexpr bool two = j + 1 < UPW;
                if constexpr (j + 2 < UPW) fetch(j + 2, nxt, j + 3 < UPW);
                soffn[j] = tabn[j * 4];
                if constexpr (two) soffn[j + 1] = tabn[(j + 1) * 4];
                const bf16x8 h0 = as_bf16x8(bh[cur][0]), m0 = as_bf16x8(bm[cur][0]), l0 = as_bf16x8(bl[cur][0]);
                const bf16x8 h1 = as_bf16x8(bh[cur][1]), m1 = as_bf16x8(bm[cur][1]), l1 = as_bf16x8(bl[cur][1]);
                // smallest terms first
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, h0, acc[j], 0, 0, 0);
                if constexpr (two) acc[j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l, h1, acc[j + 1], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, l0, acc[j], 0, 0, 0);
                if constexpr (two) acc[j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, l1, acc[j + 1], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, m0, acc[j], 0, 0, 0);
                if constexpr (two) acc[j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, m1, acc[j + 1], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, h0, acc[j], 0, 0, 0);
                if constexpr (two) acc[j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_m, h1, acc[j + 1], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, m0, acc[j], 0, 0, 0);
                if constexpr (two) acc[j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, m1, acc[j + 1], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, h0, acc[j], 0, 0, 0);
                if constexpr (two) acc[j + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h, h1, acc[j + 1], 0, 0, 0);
                pin_interleave<(two ? 12 : 6), (j + 3 < UPW ? 6 : (j + 2 < UPW ? 3 : 0)) + (two ? 2 : 1)>();
            });
#pragma unroll
            for (int j = 0; j < UPW; ++j) soff[j] = soffn[j];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue.  D[i = o][j = col]: lane (col = lane&31, half = lane>>5), reg r -> o = (r&3) + 8*(r>>2) + 4*half
    const int kp = lane & 15, half = lane >> 5, gsel = (lane >> 4) & 1;
    const int* ug = &c_unitg[cfg][ubase * 2];
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
                unsigned h[4], m[4], l[4];
                split3(fmaxf(y.x, 0.f), h[0], m[0], l[0]);
                split3(fmaxf(y.y, 0.f), h[1], m[1], l[1]);
                split3(fmaxf(y.z, 0.f), h[2], m[2], l[2]);
                split3(fmaxf(y.w, 0.f), h[3], m[3], l[3]);
                // plane layout: [tile][c8][plane][g][kp][8 ch] bf16
                char* base = a.out_act + ((size_t)tile * a.cout8 + ob * 4 + q) * CHUNK16_BYTES + (size_t)g * SLAB16_BYTES + kp * 16 + half * 8;
                uintx2 ph, pm, pl;
                ph.x = h[0] | (h[1] << 16); ph.y = h[2] | (h[3] << 16);
                pm.x = m[0] | (m[1] << 16); pm.y = m[2] | (m[3] << 16);
                pl.x = l[0] | (l[1] << 16); pl.y = l[2] | (l[3] << 16);
                *reinterpret_cast<uintx2*>(base) = ph;
                *reinterpret_cast<uintx2*>(base + PLANE16_BYTES) = pm;
                *reinterpret_cast<uintx2*>(base + 2 * PLANE16_BYTES) = pl;
            }
        }
    }
}

template <int UPW, int NOB>
static int launch16_t(const Conv16Args& a, int flags, hipStream_t s) {
    const int grid = a.nTiles * a.nOBgrid;
    hipLaunchKernelGGL((gconv16_kernel<UPW, NOB>), dim3(grid), dim3(256), LDS16_TOTAL, s, a, flags);
    HIPCHK(hipGetLastError());
    return 0;
}

int gconv16_init() {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv16_kernel<15, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS16_TOTAL));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv16_kernel<8, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS16_TOTAL));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv16_kernel<12, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS16_TOTAL));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv16_kernel<7, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS16_TOTAL));
    return 0;
}

// layer launch: L.wp16 packed weights.  cfg 0 (all 60 outputs): cout_pad multiple of 64 -> NOB = 2, else the
// single-block variant; cfg 1 (45 outputs, 23 units): <12,2>; cfg 2 (13 outputs, 7 units): <7,4>.
int launch_gconv16(const Layer& L, const char* X, int nTiles, const float* res, float* out_raw, char* out_act, int flags, hipStream_t s,
                   int cfg, float* out_raw32, float* out_act32) {
    Conv16Args a;
    a.X = X; a.Wp = reinterpret_cast<const char*>(L.wp16); a.bias = L.bias; a.bn_s = L.bn_s; a.bn_t = L.bn_t;
    a.res = res; a.out_raw = out_raw; a.out_act = out_act; a.out_raw32 = out_raw32; a.out_act32 = out_act32;
    a.nTiles = nTiles; a.cin8 = L.cin / 8; a.cout8 = L.cout_pad / 8; a.cfg = cfg;
    if (cfg == 1) {
        if (L.cout_pad % 64) { set_error("gconv16: cone-45 variant needs cout %% 64 == 0"); return YOHO_EINVAL; }
        a.nOBgrid = L.cout_pad / 64;
        return launch16_t<12, 2>(a, flags, s);
    }
    if (cfg == 2) {
        if (L.cout_pad % 128) { set_error("gconv16: cone-13 variant needs cout %% 128 == 0"); return YOHO_EINVAL; }
        a.nOBgrid = L.cout_pad / 128;
        return launch16_t<7, 4>(a, flags, s);
    }
    if (L.cout_pad % 64 == 0) { a.nOBgrid = L.cout_pad / 64; return launch16_t<15, 2>(a, flags, s); }
    a.nOBgrid = L.cout_pad / 32;
    return launch16_t<8, 1>(a, flags, s);
}

// ---------------------------------------------------------------------------------------------
// head / tail for the 16-keypoint tile layouts
// ---------------------------------------------------------------------------------------------
// x (B,32,60) f32 -> planes [tile][c8 = 4][3][60][16][8] bf16.  One workgroup per (tile, c8).
__global__ __launch_bounds__(256) void pack16_partI_kernel(const float* __restrict__ x, int B, char* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[CHUNK16_BYTES / 2];
    const int tile = blockIdx.x >> 2, c8 = blockIdx.x & 3;
    for (int i = threadIdx.x; i < T16 * 8 * G; i += 256) {
        const int kp = i / (8 * G);
        const int r = i - kp * (8 * G);
        const int cl = r / G, g = r - cl * G;
        const int bb = tile * T16 + kp;
        const float v = bb < B ? x[(size_t)bb * (F * G) + (c8 * 8 + cl) * G + g] : 0.f;
        unsigned h, m, l;
        split3(v, h, m, l);
        const int o = (g * T16 + kp) * 8 + cl;
        lds[o] = (unsigned short)h;
        lds[PLANE16_BYTES / 2 + o] = (unsigned short)m;
        lds[PLANE16_BYTES + o] = (unsigned short)l;
    }
    __syncthreads();
    uintx4* o = reinterpret_cast<uintx4*>(out + ((size_t)tile * 4 + c8) * CHUNK16_BYTES);
    const uintx4* l = reinterpret_cast<const uintx4*>(lds);
    for (int i = threadIdx.x; i < CHUNK16_BYTES / 16; i += 256) o[i] = l[i];
}

int launch_pack16_partI(const float* x, int B, int nTiles, char* out, hipStream_t s) {
    hipLaunchKernelGGL(pack16_partI_kernel, dim3(nTiles * 4), dim3(256), 0, s, x, B, out);
    HIPCHK(hipGetLastError());
    return 0;
}

// PartII head for the bf16x3 path (utils/network.py:266-269 + Conv_init's BN/ReLU): permute the group axis of
// before_eqv0 / after_eqv0 by P[pre_idx], concatenate 4 x 32 channels, BN(128) + ReLU, split into bf16 planes.
// One workgroup per (tile16, c8); c8 >> 2 selects the source tensor.
__global__ __launch_bounds__(256) void pack16_partII_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                            const float* __restrict__ s2, const float* __restrict__ s3,
                                                            const int64_t* __restrict__ pre_idx, const int* __restrict__ P,
                                                            const float* __restrict__ bn_s, const float* __restrict__ bn_t,
                                                            int M, char* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[CHUNK16_BYTES / 2];
    const int tile = blockIdx.x >> 4, c8 = blockIdx.x & 15;
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
        unsigned h, mm, l;
        split3(v, h, mm, l);
        const int o = (g * T16 + kp) * 8 + cl;
        lds[o] = (unsigned short)h;
        lds[PLANE16_BYTES / 2 + o] = (unsigned short)mm;
        lds[PLANE16_BYTES + o] = (unsigned short)l;
    }
    __syncthreads();
    uintx4* o = reinterpret_cast<uintx4*>(out + ((size_t)tile * 16 + c8) * CHUNK16_BYTES);
    const uintx4* l = reinterpret_cast<const uintx4*>(lds);
    for (int i = threadIdx.x; i < CHUNK16_BYTES / 16; i += 256) o[i] = l[i];
}

int launch_pack16_partII(const float* s0, const float* s1, const float* s2, const float* s3, const int64_t* pre_idx, const int* P,
                         const float* bn_s, const float* bn_t, int M, int nTiles16, char* out, hipStream_t s) {
    hipLaunchKernelGGL(pack16_partII_kernel, dim3(nTiles16 * 16), dim3(256), 0, s, s0, s1, s2, s3, pre_idx, P, bn_s, bn_t, M, out);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
