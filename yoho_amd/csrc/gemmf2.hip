// Irrep GEMMs of the group-Fourier conv (see gemmf.hip for the algebra and the operand packs), second blocking:
// 256 x 128 output tile per workgroup, TWO workgroups per CU.
//
// Why.  With one 256 x 256 workgroup per CU (gemmf.hip) nothing overlaps a workgroup's epilogue: its four waves hold 256
// accumulator registers each, write 256 KiB of fp32 coefficients and only then can the CU start the next tile - a quarter
// of the kernel's time is spent draining stores with the matrix pipes idle.  Here a wave owns 128 x 64 outputs (128
// accumulator registers, at most 256 registers in all), a workgroup needs 72 KiB of LDS, and two workgroups share a CU:
// while one of them runs its prologue (first DMA stages, residual loads) or its epilogue (stores), the other one's K loop
// has the matrix pipes to itself.
//
// K loop.  Steps of K = 16 (one MFMA sub-step): a step is 16 KiB of the A tile (256 rows x 16 k x {hi, lo}) + 8 KiB of
// one 128-column half of the B tile, copied by LDS DMA straight from the packs into a ring of three 24 KiB buffers
// (the packs are unchanged: a half column tile is four 2 KiB runs of the 256-column pack).  In iteration s a wave issues
// its 24 MFMAs on the fragments of step s (registers), reads the 12 fragments of step s+1 from the ring and issues its
// six 1 KiB pieces of the DMA of step s+3; at the end `s_waitcnt vmcnt(6)` (the six pieces just issued stay in flight) and
// one barrier make step s+2 readable: the DMA runs two steps ahead of its consumers.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "gemmf.h"

namespace yoho {

constexpr int F2_BUF = 24576;                 // one ring buffer: A 16 KiB [plane 2][k-group 2][row 256][16 B] | B 8 KiB [plane 2][k-group 2][col 128][16 B]
constexpr int F2_LDS = 3 * F2_BUF;

struct Frags2 {
    uintx4 ah[4], al[4], bh[2], bl[2];
};

// fragment read R (0..11) of a step, in the order of use: A lo x4, B hi x2 (products lo.hi), A hi x4, B lo x2
template <int R>
__device__ __forceinline__ void read_frag2(const char* pa, const char* pb, Frags2& f) {
    if constexpr (R < 4) f.al[R] = *reinterpret_cast<const uintx4*>(pa + 8192 + R * 512);
    else if constexpr (R < 6) f.bh[R - 4] = *reinterpret_cast<const uintx4*>(pb + (R - 4) * 512);
    else if constexpr (R < 10) f.ah[R - 6] = *reinterpret_cast<const uintx4*>(pa + (R - 6) * 512);
    else f.bl[R - 10] = *reinterpret_cast<const uintx4*>(pb + 4096 + (R - 10) * 512);
}

// piece U (0..5) of this wave's share of the DMA of one K16 step: 1 KiB each.  srcA / srcB point at the step inside the packs
// (stage * 32 KiB + sub-step * 8 KiB, wave uniform), la / lb are the per-lane byte offsets inside the A and B parts.
template <int U>
__device__ __forceinline__ void dma_piece(const char* srcA, const char* srcB, char* buf, int la, int lb, int w, int wkg, int wcb) {
    if constexpr (U < 4) {
        constexpr int plane = U >> 1, kg = U & 1;                       // A: piece (plane, kg, 64-row block w)
        __builtin_amdgcn_global_load_lds((gptr_t)(srcA + plane * 16384 + kg * 4096 + la), (lptr_t)(buf + plane * 8192 + kg * 4096 + w * 1024), 16, 0, 0);
    } else {
        constexpr int plane = U - 4;                                    // B: piece (plane, kg = w >> 1, 64-column block w & 1)
        __builtin_amdgcn_global_load_lds((gptr_t)(srcB + plane * 16384 + lb), (lptr_t)(buf + 16384 + plane * 4096 + wkg * 2048 + wcb * 1024), 16, 0, 0);
    }
}

__device__ __forceinline__ void dma_step(const char* srcA, const char* srcB, char* buf, int la, int lb, int w, int wkg, int wcb) {
    sfor<0, 6>([&](auto uc) { dma_piece<decltype(uc)::value>(srcA, srcB, buf, la, lb, w, wkg, wcb); });
}

// One K16 step: 24 MFMAs on `f` (every accumulator gets lo.hi, hi.lo, hi.hi, 8 MFMAs apart); behind them, one per MFMA, the
// 12 fragment reads of the next step and (DMA) the six DMA pieces of the step three ahead.
template <bool DMA>
__device__ __forceinline__ void step2(const Frags2& f, floatx16 (&acc)[4][2], const char* ra, const char* rb, Frags2& nf,
                                      const char* srcA, const char* srcB, char* dmabuf, int la, int lb, int w, int wkg, int wcb) {
    sfor<0, 8>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        acc[g >> 1][g & 1] = mfma_h(f.al[g >> 1], f.bh[g & 1], acc[g >> 1][g & 1]);
        read_frag2<g>(ra, rb, nf);
        __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, 8>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        acc[g >> 1][g & 1] = mfma_h(f.ah[g >> 1], f.bl[g & 1], acc[g >> 1][g & 1]);
        if constexpr (g < 4) read_frag2<8 + g>(ra, rb, nf);
        else if constexpr (DMA && g >= 4 && g < 8) dma_piece<g - 4>(srcA, srcB, dmabuf, la, lb, w, wkg, wcb);
        __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, 8>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        acc[g >> 1][g & 1] = mfma_h(f.ah[g >> 1], f.bh[g & 1], acc[g >> 1][g & 1]);
        if constexpr (DMA && g < 2) dma_piece<4 + g>(srcA, srcB, dmabuf, la, lb, w, wkg, wcb);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// Work map.  Workgroup b runs on XCD b & 7; a column tile (irrep t, ntile) and all its row tiles stay on one XCD (the B panel is
// read from HBM once and served from that XCD's L2), column tiles of an irrep are dealt round-robin over the XCDs (gemmf.hip).
// The two workgroups of a CU must NOT run in lockstep - if both reach their epilogue together nothing is gained over one large
// tile - so the slots of an XCD alternate between the d = 5 irrep (long K loops: 40000 of an XCD's 78000 steps for 256 -> 512)
// and the other four (shorter K loops, in launch order): co-resident workgroups then have different lengths and drift apart
// after the first round.  Returns false for a slot beyond the XCD's work.
enum { F2_NOSTORE = 0x100, F2_MIX = 0x200, F2_ST_SC1 = 0x400, F2_ST_NT = 0x800, F2_SPARSE4 = 0x1000, F2_SPARSE16 = 0x2000 };   // YOHO_FGEMM_DEBUG experiments, see launch_fgemm2
__host__ __device__ inline bool fg2_map(const FGemmArgs& a, int xcd, int slot, bool contiguous, int& t, int& local, int& r) {
    int cnt[NIR_ORD], ru[NIR_ORD];
    for (int u = 0; u < NIR_ORD; ++u) {
        ru[u] = (xcd + a.rot[u]) & 7;
        cnt[u] = a.NT[u] > ru[u] ? ((a.NT[u] - 1 - ru[u]) / 8 + 1) * a.MT[u] * 2 : 0;
    }
    const int L = cnt[0], S = cnt[1] + cnt[2] + cnt[3] + cnt[4];
    int list, idx;                                   // list 0: the first irrep, list 1: the others, idx: position in that list
    if (contiguous) { list = slot < L ? 0 : 1; idx = slot < L ? slot : slot - L; }
    else {
        const int P = L < S ? L : S;
        if (slot < 2 * P) { list = slot & 1; idx = slot >> 1; }
        else { list = L > S ? 0 : 1; idx = P + (slot - 2 * P); }
    }
    if (list == 0) {
        if (idx >= L) return false;
        t = 0; local = idx; r = ru[0];
        return true;
    }
    for (int u = 1; u < NIR_ORD; ++u) {
        if (idx < cnt[u]) { t = u; local = idx; r = ru[u]; return true; }
        idx -= cnt[u];
    }
    return false;
}

// byte offset of K16 step ks inside a tile's pack: stage (ks >> 1) * 32 KiB + sub-step (ks & 1) * 8 KiB
__device__ __forceinline__ size_t step_off(int ks) { return (size_t)(ks >> 1) * FG_STAGE + (size_t)(ks & 1) * 8192; }

__global__ __launch_bounds__(256, 2) void fgemm2_kernel(FGemmArgs a, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // two workgroups - the two 128-column halves - per (column tile, row tile), neighbours in their irrep's list: they stream the
    // same A stages at about the same time, so one of the two reads hits the XCD's L2
    int t = 0, local = 0, r = 0;
    if (!fg2_map(a, blockIdx.x & 7, blockIdx.x >> 3, (flags & F2_MIX) == 0, t, local, r)) return;
    const int d = a.dim[t], qbase = a.qbase[t];
    const int MT = a.MT[t], KS = d * a.cin / 32, KT = 2 * KS;
    const int nh = local & 1, pair = local >> 1;
    const int cg = pair / MT, mtile = pair - cg * MT;
    const int ntile = r + 8 * cg;
    const char* Ag = a.A + a.a_off[t] + (size_t)mtile * KS * FG_STAGE;
    const char* Bg = a.B + a.b_off[t] + (size_t)ntile * KS * FG_STAGE + nh * 2048;
    const int wm = w >> 1, wn = w & 1;
    const int wkg = w >> 1, wcb = w & 1;                                 // this wave's B piece: k-group, 64-column block
    const int la = w * 1024 + lane * 16;                                 // DMA source offsets inside a step
    const int lb = wkg * 4096 + wcb * 1024 + lane * 16;

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int lane_a = (lane >> 5) * 4096 + (wm * 128 + (lane & 31)) * 16;
    const int lane_b = 16384 + (lane >> 5) * 2048 + (wn * 64 + (lane & 31)) * 16;

    // prologue: steps 0, 1, 2 into the three buffers
    dma_step(Ag, Bg, smem, la, lb, w, wkg, wcb);
    dma_step(Ag + step_off(1), Bg + step_off(1), smem + F2_BUF, la, lb, w, wkg, wcb);
    if (KT > 2) dma_step(Ag + step_off(2), Bg + step_off(2), smem + 2 * F2_BUF, la, lb, w, wkg, wcb);
    const bool rows_live = (mtile * 256 + wm * 128) < d * a.cout;       // else: all 128 rows of this wave are padding (cout = 32)
    const int colbase = ntile * 256 + nh * 128 + wn * 64;
    const int jidx = colbase / a.kppad, kp0 = colbase - jidx * a.kppad;
    if ((flags & EPI_RES) && rows_live) {
        // the accumulators start from the residual (x 1 / descale, a power of two): its loads fly with the first DMA stages
        const float inv = 1.f / a.descale;
        const int half = lane >> 5, kp32 = lane & 31, cout8 = a.cout >> 3;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
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
                    const floatx4 v = *reinterpret_cast<const floatx4*>(a.res + (ok ? off : 0)) * (ok ? inv : 0.f);     // branch-free
                    acc[ai][bi][4 * q4 + 0] = v.x; acc[ai][bi][4 * q4 + 1] = v.y;
                    acc[ai][bi][4 * q4 + 2] = v.z; acc[ai][bi][4 * q4 + 3] = v.w;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (!rows_live) {
        // keep this wave's share of the DMA and the barriers going (the live waves meet at one barrier per step but the last)
        __builtin_amdgcn_s_barrier();                          // the live waves' barrier behind their reads of step 0 (below)
        for (int s = 0; s + 1 < KT; ++s) {
            if (s + 3 < KT) dma_step(Ag + step_off(s + 3), Bg + step_off(s + 3), smem + (s % 3) * F2_BUF, la, lb, w, wkg, wcb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    Frags2 P, Q;
    sfor<0, 12>([&](auto rc) { read_frag2<decltype(rc)::value>(smem + lane_a, smem + lane_b, P); });
    // every wave has step 0 in registers before any wave's first DMA (step 3) lands in buffer 0 (see fgemm3s_kloop)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ring offsets: step s lives in buffer s % 3
    int cur = 0, nxt = F2_BUF;                   // buffer of step s (already in registers: the target of the DMA of step s+3), of step s+1
    int s = 0;
    auto advance = [&]() { cur = nxt; nxt = nxt == 2 * F2_BUF ? 0 : nxt + F2_BUF; ++s; };
    // main loop: both steps of a pair still have a DMA to issue (s + 4 < KT)
    for (; s + 4 < KT;) {
        step2<true>(P, acc, smem + nxt + lane_a, smem + nxt + lane_b, Q, uniform_ptr(Ag + step_off(s + 3)), uniform_ptr(Bg + step_off(s + 3)),
                    smem + cur, la, lb, w, wkg, wcb);
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        advance();
        step2<true>(Q, acc, smem + nxt + lane_a, smem + nxt + lane_b, P, uniform_ptr(Ag + step_off(s + 3)), uniform_ptr(Bg + step_off(s + 3)),
                    smem + cur, la, lb, w, wkg, wcb);
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        advance();
    }
    // tail: the last four steps (two if KT == 2); only the first of them still has a DMA (step KT - 1) to issue
    if (KT >= 4) {
        step2<true>(P, acc, smem + nxt + lane_a, smem + nxt + lane_b, Q, uniform_ptr(Ag + step_off(s + 3)), uniform_ptr(Bg + step_off(s + 3)),
                    smem + cur, la, lb, w, wkg, wcb);
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        advance();
        step2<false>(Q, acc, smem + nxt + lane_a, smem + nxt + lane_b, P, nullptr, nullptr, nullptr, la, lb, w, wkg, wcb);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        advance();
    }
    step2<false>(P, acc, smem + nxt + lane_a, smem + nxt + lane_b, Q, nullptr, nullptr, nullptr, la, lb, w, wkg, wcb);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    advance();
    // last step: its "next" fragments are read from a buffer that holds valid (unused) data
    step2<false>(Q, acc, smem + cur + lane_a, smem + cur + lane_b, P, nullptr, nullptr, nullptr, la, lb, w, wkg, wcb);

    // ---- epilogue: D[row][col]: lane (col = lane & 31, half = lane >> 5), reg e -> row = (e & 3) + 8 * (e >> 2) + 4 * half
    const int half = lane >> 5, kp32 = lane & 31;
    const int cout8 = a.cout >> 3;
    const bool addb = (d == 1);                              // trivial irrep: coefficient 0 carries sqrt(60) * bias
    if ((flags & F2_SPARSE4) && (blockIdx.x >> 3) % 4 != 0) flags |= F2_NOSTORE;          // experiment: only every 4th / 16th workgroup stores
    if ((flags & F2_SPARSE16) && (blockIdx.x >> 3) % 16 != 0) flags |= F2_NOSTORE;
    unsigned top = 0u;                                       // largest |coefficient| written (bit pattern; inf / NaN order above)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
        const int tile32 = (kp0 >> 5) + bi;
        if (tile32 >= a.nT32) continue;
#pragma unroll
        for (int ai = 0; ai < 4; ++ai) {
            const int rowb = mtile * 256 + wm * 128 + ai * 32;     // the 32 rows of an MFMA tile share i (cout is a multiple of 32)
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
                if (flags & F2_ST_SC1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, val), orsrc, (int)(off * 4), 0, 16);
                else if (flags & F2_ST_NT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, val), orsrc, (int)(off * 4), 0, 2);
                else if (!(flags & F2_NOSTORE)) *reinterpret_cast<floatx4*>(a.out + off) = val;
                top = max(max(top, __float_as_uint(val.x) & 0x7FFFFFFFu), __float_as_uint(val.y) & 0x7FFFFFFFu);
                top = max(max(top, __float_as_uint(val.z) & 0x7FFFFFFFu), __float_as_uint(val.w) & 0x7FFFFFFFu);
            }
        }
    }
    note_range_bits(a.rflag, top, FP16_MAX / HF_ASCALE);     // the consumer multiplies by HF_ASCALE and converts to fp16
}

// ---------------------------------------------------------------------------------------------------------------
// fgemm3: the same K loop with the two 128-column halves of a 256 x 256 tile in ONE workgroup of eight waves (two per SIMD).
// The halves share the A stage in LDS, so the L2 -> LDS traffic and the fabric traffic are those of the 256 x 256 blocking
// (a step is 16 KiB of A + 16 KiB of B, ring of three 32 KiB buffers = 96 KiB), while every SIMD still has two waves.
// ---------------------------------------------------------------------------------------------------------------
constexpr int F3_BUF = 32768;                 // A 16 KiB [plane 2][k-group 2][row 256][16 B] | B 16 KiB [plane 2][k-group 2][col 256][16 B]
constexpr int F3_LDS = 3 * F3_BUF;

template <int R>
__device__ __forceinline__ void read_frag3(const char* pa, const char* pb, Frags2& f) {
    if constexpr (R < 4) f.al[R] = *reinterpret_cast<const uintx4*>(pa + 8192 + R * 512);
    else if constexpr (R < 6) f.bh[R - 4] = *reinterpret_cast<const uintx4*>(pb + (R - 4) * 512);
    else if constexpr (R < 10) f.ah[R - 6] = *reinterpret_cast<const uintx4*>(pa + (R - 6) * 512);
    else f.bl[R - 10] = *reinterpret_cast<const uintx4*>(pb + 8192 + (R - 10) * 512);
}

// piece U (0..3) of a wave's share of a step's DMA (32 pieces of 1 KiB over 8 waves): U = 0, 1: A plane U; U = 2, 3: B plane U - 2;
// the wave's (k-group, 64-row/column block) is folded into `la` (source) and `ldst` (destination)
template <int U>
__device__ __forceinline__ void dma_piece3(const char* srcA, const char* srcB, char* buf, int la, int ldst) {
    constexpr int plane = U & 1;
    if constexpr (U < 2) __builtin_amdgcn_global_load_lds((gptr_t)(srcA + plane * 16384 + la), (lptr_t)(buf + plane * 8192 + ldst), 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((gptr_t)(srcB + plane * 16384 + la), (lptr_t)(buf + 16384 + plane * 8192 + ldst), 16, 0, 0);
}

__device__ __forceinline__ void dma_step3(const char* srcA, const char* srcB, char* buf, int la, int lb, int w, int wkg, int wcb) {
    const int ldst = wkg * 4096 + wcb * 1024;
    sfor<0, 4>([&](auto uc) { dma_piece3<decltype(uc)::value>(srcA, srcB, buf, la, ldst); });
}

template <bool DMA>
__device__ __forceinline__ void step3(const Frags2& f, floatx16 (&acc)[4][2], const char* ra, const char* rb, Frags2& nf,
                                      const char* srcA, const char* srcB, char* dmabuf, int la, int lb, int w, int wkg, int wcb) {
    const int ldst = wkg * 4096 + wcb * 1024;
    sfor<0, 8>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        acc[g >> 1][g & 1] = mfma_h(f.al[g >> 1], f.bh[g & 1], acc[g >> 1][g & 1]);
        read_frag3<g>(ra, rb, nf);
        __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, 8>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        acc[g >> 1][g & 1] = mfma_h(f.ah[g >> 1], f.bl[g & 1], acc[g >> 1][g & 1]);
        if constexpr (g < 4) read_frag3<8 + g>(ra, rb, nf);
        else if constexpr (DMA) dma_piece3<g - 4>(srcA, srcB, dmabuf, la, ldst);
        __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, 8>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        acc[g >> 1][g & 1] = mfma_h(f.ah[g >> 1], f.bh[g & 1], acc[g >> 1][g & 1]);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// work map of gemmf.hip (one workgroup per 256 x 256 tile)
__host__ __device__ inline bool fg3_map(const FGemmArgs& a, int xcd, int slot, int& t, int& local, int& r) {
    int start = 0;
    for (int u = 0; u < NIR_ORD; ++u) {
        const int ru = (xcd + a.rot[u]) & 7;
        const int cnt = a.NT[u] > ru ? ((a.NT[u] - 1 - ru) / 8 + 1) * a.MT[u] : 0;
        if (slot < start + cnt) { t = u; local = slot - start; r = ru; return true; }
        start += cnt;
    }
    return false;
}

__global__ __launch_bounds__(512, 2) void fgemm3_kernel(FGemmArgs a, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);             // 8 waves: waves 0-3 own the left 128 columns, 4-7 the right
    const int w = w8 & 3;
    int t = 0, local = 0, r = 0;
    if (!fg3_map(a, blockIdx.x & 7, blockIdx.x >> 3, t, local, r)) return;
    const int d = a.dim[t], qbase = a.qbase[t];
    const int MT = a.MT[t], KS = d * a.cin / 32, KT = 2 * KS;
    const int nh = w8 >> 2, pair = local;
    const int cg = pair / MT, mtile = pair - cg * MT;
    const int ntile = r + 8 * cg;
    const char* Ag = a.A + a.a_off[t] + (size_t)mtile * KS * FG_STAGE;
    const char* Bg = a.B + a.b_off[t] + (size_t)ntile * KS * FG_STAGE;
    const int wm = w >> 1, wn = w & 1;
    const int wkg = w8 >> 2, wcb = w8 & 3;                               // this wave's DMA pieces: k-group, 64-row / 64-column block
    const int la = wkg * 4096 + wcb * 1024 + lane * 16;                  // DMA source offset inside a step (same for the A and the B piece)
    const int lb = la;

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int lane_a = (lane >> 5) * 4096 + (wm * 128 + (lane & 31)) * 16;
    const int lane_b = 16384 + (lane >> 5) * 4096 + (nh * 128 + wn * 64 + (lane & 31)) * 16;

    // prologue: steps 0, 1, 2 into the three buffers
    dma_step3(Ag, Bg, smem, la, lb, w, wkg, wcb);
    dma_step3(Ag + step_off(1), Bg + step_off(1), smem + F3_BUF, la, lb, w, wkg, wcb);
    if (KT > 2) dma_step3(Ag + step_off(2), Bg + step_off(2), smem + 2 * F3_BUF, la, lb, w, wkg, wcb);
    const bool rows_live = (mtile * 256 + wm * 128) < d * a.cout;       // else: all 128 rows of this wave are padding (cout = 32)
    const int colbase = ntile * 256 + nh * 128 + wn * 64;
    const int jidx = colbase / a.kppad, kp0 = colbase - jidx * a.kppad;
    if ((flags & EPI_RES) && rows_live) {
        // the accumulators start from the residual (x 1 / descale, a power of two): its loads fly with the first DMA stages
        const float inv = 1.f / a.descale;
        const int half = lane >> 5, kp32 = lane & 31, cout8 = a.cout >> 3;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
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
                    const floatx4 v = *reinterpret_cast<const floatx4*>(a.res + (ok ? off : 0)) * (ok ? inv : 0.f);     // branch-free
                    acc[ai][bi][4 * q4 + 0] = v.x; acc[ai][bi][4 * q4 + 1] = v.y;
                    acc[ai][bi][4 * q4 + 2] = v.z; acc[ai][bi][4 * q4 + 3] = v.w;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (!rows_live) {
        // keep this wave's share of the DMA and the barriers going (the live waves meet at one barrier per step but the last)
        __builtin_amdgcn_s_barrier();                          // the live waves' barrier behind their reads of step 0 (below)
        for (int s = 0; s + 1 < KT; ++s) {
            if (s + 3 < KT) dma_step3(Ag + step_off(s + 3), Bg + step_off(s + 3), smem + (s % 3) * F3_BUF, la, lb, w, wkg, wcb);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    Frags2 P, Q;
    sfor<0, 12>([&](auto rc) { read_frag3<decltype(rc)::value>(smem + lane_a, smem + lane_b, P); });
    // every wave has step 0 in registers before any wave's first DMA (step 3) lands in buffer 0 (see fgemm3s_kloop)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ring offsets: step s lives in buffer s % 3
    int cur = 0, nxt = F3_BUF;                   // buffer of step s (already in registers: the target of the DMA of step s+3), of step s+1
    int s = 0;
    auto advance = [&]() { cur = nxt; nxt = nxt == 2 * F3_BUF ? 0 : nxt + F3_BUF; ++s; };
    // main loop: both steps of a pair still have a DMA to issue (s + 4 < KT)
    for (; s + 4 < KT;) {
        step3<true>(P, acc, smem + nxt + lane_a, smem + nxt + lane_b, Q, uniform_ptr(Ag + step_off(s + 3)), uniform_ptr(Bg + step_off(s + 3)),
                    smem + cur, la, lb, w, wkg, wcb);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        advance();
        step3<true>(Q, acc, smem + nxt + lane_a, smem + nxt + lane_b, P, uniform_ptr(Ag + step_off(s + 3)), uniform_ptr(Bg + step_off(s + 3)),
                    smem + cur, la, lb, w, wkg, wcb);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        advance();
    }
    // tail: the last four steps (two if KT == 2); only the first of them still has a DMA (step KT - 1) to issue
    if (KT >= 4) {
        step3<true>(P, acc, smem + nxt + lane_a, smem + nxt + lane_b, Q, uniform_ptr(Ag + step_off(s + 3)), uniform_ptr(Bg + step_off(s + 3)),
                    smem + cur, la, lb, w, wkg, wcb);
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        advance();
        step3<false>(Q, acc, smem + nxt + lane_a, smem + nxt + lane_b, P, nullptr, nullptr, nullptr, la, lb, w, wkg, wcb);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        advance();
    }
    step3<false>(P, acc, smem + nxt + lane_a, smem + nxt + lane_b, Q, nullptr, nullptr, nullptr, la, lb, w, wkg, wcb);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    advance();
    // last step: its "next" fragments are read from a buffer that holds valid (unused) data
    step3<false>(Q, acc, smem + cur + lane_a, smem + cur + lane_b, P, nullptr, nullptr, nullptr, la, lb, w, wkg, wcb);

    // ---- epilogue: D[row][col]: lane (col = lane & 31, half = lane >> 5), reg e -> row = (e & 3) + 8 * (e >> 2) + 4 * half
    const int half = lane >> 5, kp32 = lane & 31;
    const int cout8 = a.cout >> 3;
    const bool addb = (d == 1);                              // trivial irrep: coefficient 0 carries sqrt(60) * bias
        unsigned top = 0u;                                       // largest |coefficient| written (bit pattern; inf / NaN order above)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
        const int tile32 = (kp0 >> 5) + bi;
        if (tile32 >= a.nT32) continue;
#pragma unroll
        for (int ai = 0; ai < 4; ++ai) {
            const int rowb = mtile * 256 + wm * 128 + ai * 32;     // the 32 rows of an MFMA tile share i (cout is a multiple of 32)
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
                if (flags & F2_ST_SC1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, val), orsrc, (int)(off * 4), 0, 16);
                else if (flags & F2_ST_NT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, val), orsrc, (int)(off * 4), 0, 2);
                else if (!(flags & F2_NOSTORE)) *reinterpret_cast<floatx4*>(a.out + off) = val;
                top = max(max(top, __float_as_uint(val.x) & 0x7FFFFFFFu), __float_as_uint(val.y) & 0x7FFFFFFFu);
                top = max(max(top, __float_as_uint(val.z) & 0x7FFFFFFFu), __float_as_uint(val.w) & 0x7FFFFFFFu);
            }
        }
    }
    note_range_bits(a.rflag, top, FP16_MAX / HF_ASCALE);     // the consumer multiplies by HF_ASCALE and converts to fp16
}


// ---------------------------------------------------------------------------------------------------------------
// fgemm3c (gconv_mode 7, "fgemm8"; round 5, opt-in): fgemm3 with the two CORRECTION products of the fp16 split on the fp8 matrix pipe.
// A product a * w is a_h w_h + a_h w_l + a_l w_h; the corrections are 2^-11 of the main term and survive e4m3 (3 mantissa bits: 2^-15
// of the sum; tools/fp8_correction_study.py: 1.5e-5 worst relative error of the descriptor against 1.4e-6 and a tolerance of 1e-4).
// v_mfma_scale_f32_32x32x64_f8f6f4 does K = 64 at twice the fp16 rate, and ONE such MFMA takes both corrections of TWO K16 steps by
// K-concatenation:  A' = [a_h(s0) | a_l(s0) | a_h(s1) | a_l(s1)],  B' = [w_l(s0) | w_h(s0) | w_l(s1) | w_h(s1)]  (32 fp8 per lane; the
// lane's eight values of a fragment keep their place, so A' and B' pair element by element exactly like the fp16 fragments do).
// Per K16 step: 8 fp16 MFMAs + 4 fp8 MFMAs of twice the length = 2 / 3 of fgemm3's matrix time, plus 16 v_cvt_scalef32_pk_fp8_f16
// (two values each) on the activation fragments; the weight operand's fp8 fragments are packed on the host (Layer::wpg8).  The K loop is the one tools/fp8_corr_probe.hip measured (1.23-1.32 x over
// the same loop with three fp16 products): fragments of the CURRENT step read at its head, DMA two steps ahead, no hand interleaving.
// Tile shape, operand packs, LDS image, work map, residual start and epilogue are fgemm3's; the sums differ from fgemm3's in the last
// bits by construction.
// ---------------------------------------------------------------------------------------------------------------
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef short shortx2 __attribute__((ext_vector_type(2)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
struct Frags8 {
    intx8 a[4], b[2];          // dwords [4 p .. 4 p + 1] = hi plane (A) / lo plane (B) of the pair's step p, [4 p + 2 .. 4 p + 3] = lo (A) / hi (B)
};
struct C8Scales {
    float ah, al, bh, bl;
    int mfma_b;
};
struct Int2 { int x, y; };
// eight fp16 (one fragment) / scale -> eight fp8 e4m3 (two dwords)
__device__ __forceinline__ Int2 frag_to_fp8(uintx4 f, float scale) {
    union { unsigned u; halfx2 h; } p0, p1, p2, p3;
    p0.u = f[0]; p1.u = f[1]; p2.u = f[2]; p3.u = f[3];
    shortx2 r = {0, 0}, q = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, p0.h, scale, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, p1.h, scale, true);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, p2.h, scale, false);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, p3.h, scale, true);
    Int2 o;
    __builtin_memcpy(&o.x, &r, 4);
    __builtin_memcpy(&o.y, &q, 4);
    return o;
}

// the products of one K16 step of a pair (PAR = 0, 1) on its fragments: main product on the fp16 pipe, the fragments into half PAR of the
// fp8 operands, and behind the pair's second step the eight fp8 MFMAs that take both corrections of both steps
template <int PAR>
__device__ __forceinline__ void step3c(const Frags2& f, floatx16 (&acc)[4][2], Frags8& c8, const C8Scales& sc) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_h(f.ah[i], f.bh[j], acc[i][j]);
    // the weight operand's fp8 fragments come ready-made: fgemm3c's weight pack (Layer::wpg8) carries, in the place of the lo plane, the
    // 16 bytes [e4m3(hi / 4) x 8 | e4m3(lo * 512) x 8] of every (row, k-group) unit - exactly this step's four dwords of A'
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) c8.a[i][4 * PAR + e] = (int)f.al[i][e];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const Int2 l = frag_to_fp8(f.bl[j], sc.bl), h = frag_to_fp8(f.bh[j], sc.bh);
        c8.b[j][4 * PAR] = l.x; c8.b[j][4 * PAR + 1] = l.y; c8.b[j][4 * PAR + 2] = h.x; c8.b[j][4 * PAR + 3] = h.y;
    }
    if constexpr (PAR == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(c8.a[i], c8.b[j], acc[i][j], 0, 0, 0, 127 + 2 - 11, 0, sc.mfma_b);
    }
}

__global__ __launch_bounds__(512, 2) void fgemm3c_kernel(FGemmArgs a, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);             // 8 waves: waves 0-3 own the left 128 columns, 4-7 the right
    const int w = w8 & 3;
    int t = 0, local = 0, r = 0;
    if (!fg3_map(a, blockIdx.x & 7, blockIdx.x >> 3, t, local, r)) return;
    const int d = a.dim[t], qbase = a.qbase[t];
    const int MT = a.MT[t], KS = d * a.cin / 32, KT = 2 * KS;
    const int nh = w8 >> 2, pair = local;
    const int cg = pair / MT, mtile = pair - cg * MT;
    const int ntile = r + 8 * cg;
    const char* Ag = a.A + a.a_off[t] + (size_t)mtile * KS * FG_STAGE;
    const char* Bg = a.B + a.b_off[t] + (size_t)ntile * KS * FG_STAGE;
    const int wm = w >> 1, wn = w & 1;
    const int wkg = w8 >> 2, wcb = w8 & 3;                               // this wave's DMA pieces: k-group, 64-row / 64-column block
    const int la = wkg * 4096 + wcb * 1024 + lane * 16;                  // DMA source offset inside a step (same for the A and the B piece)
    const int lb = la;

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fp8 conversion scales (the conversion DIVIDES by its scale operand; powers of two throughout).  Weights: max |What| * wscale is in
    // [2^9, 2^10) (pack_fgemm_weights) -> / 4 puts the hi plane below 256 (e4m3 tops out at 448); their lo plane is 2^-11 of that.
    // Activations: the producer of the planes left the largest magnitude it wrote in *a.amax (bit pattern of a non-negative float):
    // 2^k with k = floor(log2(amax)) - 7 puts it into [128, 256).  Both correction products then carry 1 / (4 * 2^k * 2^-11), undone by
    // the block scales of the MFMA (E8M0, 127 = 2^0).
    const unsigned amax_bits = __builtin_amdgcn_readfirstlane(*a.amax);
    int kexp = (int)(amax_bits >> 23) - 127 - 7;
    kexp = kexp < -60 ? -60 : (kexp > 9 ? 9 : kexp);
    C8Scales sc;
    sc.ah = 4.f; sc.al = 4.f / 2048.f;
    sc.bh = __uint_as_float((unsigned)(kexp + 127) << 23); sc.bl = __uint_as_float((unsigned)(kexp - 11 + 127) << 23);
    sc.mfma_b = 127 + kexp;
    const int lane_a = (lane >> 5) * 4096 + (wm * 128 + (lane & 31)) * 16;
    const int lane_b = 16384 + (lane >> 5) * 4096 + (nh * 128 + wn * 64 + (lane & 31)) * 16;

    // prologue: steps 0 and 1 (the loop stages two steps ahead of the one it multiplies)
    dma_step3(Ag, Bg, smem, la, lb, w, wkg, wcb);
    dma_step3(Ag + step_off(1), Bg + step_off(1), smem + F3_BUF, la, lb, w, wkg, wcb);
    const bool rows_live = (mtile * 256 + wm * 128) < d * a.cout;       // else: all 128 rows of this wave are padding (cout = 32)
    const int colbase = ntile * 256 + nh * 128 + wn * 64;
    const int jidx = colbase / a.kppad, kp0 = colbase - jidx * a.kppad;
    if ((flags & EPI_RES) && rows_live) {
        // the accumulators start from the residual (x 1 / descale, a power of two): its loads fly with the first DMA stages
        const float inv = 1.f / a.descale;
        const int half = lane >> 5, kp32 = lane & 31, cout8 = a.cout >> 3;
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
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
                    const floatx4 v = *reinterpret_cast<const floatx4*>(a.res + (ok ? off : 0)) * (ok ? inv : 0.f);     // branch-free
                    acc[ai][bi][4 * q4 + 0] = v.x; acc[ai][bi][4 * q4 + 1] = v.y;
                    acc[ai][bi][4 * q4 + 2] = v.z; acc[ai][bi][4 * q4 + 3] = v.w;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    Frags8 c8;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) c8.a[i][e] = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) c8.b[j][e] = 0;
    // Step s lives in ring buffer s % 3.  At the head of step s: its DMA (issued two steps ago) has landed - the four pieces issued one
    // step ago may stay in flight -, one barrier, then the DMA of step s + 2 goes into the buffer every wave finished reading BEFORE that
    // barrier (step s - 1's), the twelve fragments of step s are read, and the products are issued: the two waves of a SIMD cover each
    // other's fragment latency (tools/fp8_corr_probe.hip: this plain loop runs the three-product arithmetic as fast as fgemm3's
    // hand-interleaved one).  Past the end the last step is staged again (into a buffer nobody reads any more): uniform counted waits.
    int cur = 0, stg = 2 * F3_BUF;                // buffer of step s, buffer the DMA of step s + 2 goes to
    int s = 0;
    auto one_step = [&](auto parity) {
        constexpr int PAR = decltype(parity)::value;
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {
            const int s2 = s + 2 < KT ? s + 2 : KT - 1;
            dma_step3(uniform_ptr(Ag + step_off(s2)), uniform_ptr(Bg + step_off(s2)), smem + stg, la, lb, w, wkg, wcb);
        }
        Frags2 f;
        sfor<0, 12>([&](auto rc) { read_frag3<decltype(rc)::value>(smem + cur + lane_a, smem + cur + lane_b, f); });
        step3c<PAR>(f, acc, c8, sc);
        stg = cur;                               // (s + 3) % 3 == s % 3
        cur = cur == 2 * F3_BUF ? 0 : cur + F3_BUF;
        ++s;
    };
    if (rows_live) {
        for (int it = 0; it < KT; it += 2) {     // KT = 2 KS is even
            one_step(std::integral_constant<int, 0>{});
            one_step(std::integral_constant<int, 1>{});
        }
    } else {
        // all 128 rows of this wave are padding: keep its share of the DMA and the barriers going
        for (int it = 0; it < KT; ++it) {
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int s2 = s + 2 < KT ? s + 2 : KT - 1;
            dma_step3(uniform_ptr(Ag + step_off(s2)), uniform_ptr(Bg + step_off(s2)), smem + stg, la, lb, w, wkg, wcb);
            stg = cur;
            cur = cur == 2 * F3_BUF ? 0 : cur + F3_BUF;
            ++s;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the re-staged tail pieces must not land in the LDS of the next workgroup
    if (!rows_live) return;

    // ---- epilogue: D[row][col]: lane (col = lane & 31, half = lane >> 5), reg e -> row = (e & 3) + 8 * (e >> 2) + 4 * half
    const int half = lane >> 5, kp32 = lane & 31;
    const int cout8 = a.cout >> 3;
    const bool addb = (d == 1);                              // trivial irrep: coefficient 0 carries sqrt(60) * bias
        unsigned top = 0u;                                       // largest |coefficient| written (bit pattern; inf / NaN order above)
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
        const int tile32 = (kp0 >> 5) + bi;
        if (tile32 >= a.nT32) continue;
#pragma unroll
        for (int ai = 0; ai < 4; ++ai) {
            const int rowb = mtile * 256 + wm * 128 + ai * 32;     // the 32 rows of an MFMA tile share i (cout is a multiple of 32)
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
                if (flags & F2_ST_SC1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, val), orsrc, (int)(off * 4), 0, 16);
                else if (flags & F2_ST_NT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4, val), orsrc, (int)(off * 4), 0, 2);
                else if (!(flags & F2_NOSTORE)) *reinterpret_cast<floatx4*>(a.out + off) = val;
                top = max(max(top, __float_as_uint(val.x) & 0x7FFFFFFFu), __float_as_uint(val.y) & 0x7FFFFFFFu);
                top = max(max(top, __float_as_uint(val.z) & 0x7FFFFFFFu), __float_as_uint(val.w) & 0x7FFFFFFFu);
            }
        }
    }
    note_range_bits(a.rflag, top, FP16_MAX / HF_ASCALE);     // the consumer multiplies by HF_ASCALE and converts to fp16
}



// ---------------------------------------------------------------------------------------------------------------
// cgemm (round 6): PartII's 13-element cone layer (utils/network.py:243-249,23-65 evaluated only where the g = 0 output needs it) as ONE
// implicit GEMM on fgemm3c's K loop:
//     y[o, (j, m)] = sum_k sum_c W[o, c, k] * act[c, N[n_j][k], m]        M = cout = 512, N = 13 output elements x matches, K = 13 taps x cin
// The "gather" of the 13-tap group convolution is a choice of B-operand STAGE BLOCK per (output element j, tap k): the producer
// (gft16_kernel<G16_INVG>) leaves the activated first-layer output as [column tile of 256 matches][cone slot][32-channel block] x 32 KiB in
// exactly the image an LDS stage wants, so a K step's B half is still one straight LDS-DMA copy - only its base moves with the tap
// (a.slot: slot of N[n_j][k], 169 bytes).  A = the layer's weights as a plain [cout][tap * cin + c] matrix in fgemm's A pack (hi plane +
// fp8 correction operand in place of the lo plane for FP8, hi + lo planes else).  Tile shape, LDS image, DMA, fragment reads and the
// products are fgemm3c's / fgemm3's (three fp16 products, or one fp16 product + the two corrections on the fp8 pipe); the epilogue
// applies bias, the next layer's BN + ReLU and writes fp16x2 planes in the layout cone1_kernel stages from
// ([tile16][c8][plane][60 slabs][16 matches][8 ch], slab n_j).  Replaces gconv16_kernel<7,4,2> in PartII modes 3 / 4: no tap-pair or
// unit padding (14 / 13 twice), both operands blocked through LDS.
// ---------------------------------------------------------------------------------------------------------------
struct CGemmArgs {
    const char* A;            // [mtile][tap * KSt + cb][32 KiB]
    const char* B;            // [column tile][slot][cb][32 KiB]
    const float* bias;
    const float* bn_s;
    const float* bn_t;
    char* out;                // cone1 planes
    int NTm, MT, KSt, ntap, nj, nslot, c8out, nT32, nTiles16;
    float descale;
    int* rflag;
    const unsigned* amax;     // FP8: largest |B plane value| (float bit pattern) left by the producer
    unsigned char slot[13 * 13];
    unsigned char outg[16];
};

__device__ __forceinline__ void cg_split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x;
    x.x = x0; x.y = x1;
    const halfx2 h = __builtin_convertvector(x, halfx2);
    const f2 r = x - __builtin_convertvector(h, f2);
    const halfx2 l = __builtin_convertvector(r, halfx2);
    __builtin_memcpy(&hi, &h, 4);
    __builtin_memcpy(&lo, &l, 4);
}

template <bool FP8>
__global__ __launch_bounds__(512, 2) void cgemm_kernel(CGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int tapoff[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);             // 8 waves: waves 0-3 own the left 128 columns, 4-7 the right
    const int w = w8 & 3;
    // work map: both row tiles of a column tile are neighbours on one XCD (the B panel is served from that XCD's L2 to the second one)
    const int xcd = blockIdx.x & 7, slot_ = blockIdx.x >> 3;
    const int ct = xcd + 8 * (slot_ / a.MT), mtile = slot_ % a.MT;
    if (ct >= a.nj * a.NTm) return;
    const int j = ct / a.NTm, mt = ct - j * a.NTm;
    const int SPT = 2 * a.KSt;                                           // K16 steps per tap
    const int KT = a.ntap * SPT;
    if (tid < 16) tapoff[tid] = tid < a.ntap ? (int)a.slot[j * 13 + tid] : 0;
    const char* Ag = a.A + (size_t)mtile * (KT / 2) * FG_STAGE;
    const char* Bt = a.B + (size_t)mt * a.nslot * a.KSt * FG_STAGE;      // this column tile's slots
    const int nh = w8 >> 2;
    const int wm = w >> 1, wn = w & 1;
    const int wkg = w8 >> 2, wcb = w8 & 3;
    const int la = wkg * 4096 + wcb * 1024 + lane * 16;
    const int lb = la;
    __syncthreads();
    auto bsrc = [&](int s) -> const char* {
        const int tap = s / SPT, within = s - tap * SPT;
        const int sl = __builtin_amdgcn_readfirstlane(tapoff[tap]);
        return Bt + (size_t)sl * a.KSt * FG_STAGE + step_off(within);
    };

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;

    C8Scales sc;
    sc.ah = 4.f; sc.al = 4.f / 2048.f; sc.bh = 1.f; sc.bl = 1.f; sc.mfma_b = 127;
    if constexpr (FP8) {
        const unsigned amax_bits = __builtin_amdgcn_readfirstlane(*a.amax);
        int kexp = (int)(amax_bits >> 23) - 127 - 7;
        kexp = kexp < -60 ? -60 : (kexp > 9 ? 9 : kexp);
        sc.bh = __uint_as_float((unsigned)(kexp + 127) << 23); sc.bl = __uint_as_float((unsigned)(kexp - 11 + 127) << 23);
        sc.mfma_b = 127 + kexp;
    }
    const int lane_a = (lane >> 5) * 4096 + (wm * 128 + (lane & 31)) * 16;
    const int lane_b = 16384 + (lane >> 5) * 4096 + (nh * 128 + wn * 64 + (lane & 31)) * 16;

    dma_step3(Ag, uniform_ptr(bsrc(0)), smem, la, lb, w, wkg, wcb);
    dma_step3(Ag + step_off(1), uniform_ptr(bsrc(1)), smem + F3_BUF, la, lb, w, wkg, wcb);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    Frags8 c8;
    if constexpr (FP8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) c8.a[i][e] = 0;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 8; ++e) c8.b[jj][e] = 0;
    }
    // fgemm3c's loop: step s lives in ring buffer s % 3; at its head the DMA of step s + 2 is issued into the buffer every wave finished
    // reading before the barrier, the fragments of step s are read and the products issued; the two waves of a SIMD cover each other
    int cur = 0, stg = 2 * F3_BUF;
    int s = 0;
    auto one_step = [&](auto parity) {
        constexpr int PAR = decltype(parity)::value;
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {
            const int s2 = s + 2 < KT ? s + 2 : KT - 1;
            dma_step3(uniform_ptr(Ag + step_off(s2)), uniform_ptr(bsrc(s2)), smem + stg, la, lb, w, wkg, wcb);
        }
        Frags2 f;
        sfor<0, 12>([&](auto rc) { read_frag3<decltype(rc)::value>(smem + cur + lane_a, smem + cur + lane_b, f); });
        if constexpr (FP8) step3c<PAR>(f, acc, c8, sc);
        else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) acc[i][jj] = mfma_h(f.al[i], f.bh[jj], acc[i][jj]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) acc[i][jj] = mfma_h(f.ah[i], f.bl[jj], acc[i][jj]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) acc[i][jj] = mfma_h(f.ah[i], f.bh[jj], acc[i][jj]);
        }
        stg = cur;
        cur = cur == 2 * F3_BUF ? 0 : cur + F3_BUF;
        ++s;
    };
    for (int it = 0; it < KT; it += 2) {         // KT is even
        one_step(std::integral_constant<int, 0>{});
        one_step(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the re-staged tail pieces must not land in the LDS of the next workgroup

    // ---- epilogue: D[row][col]: lane (col = lane & 31, half = lane >> 5), reg e -> row = (e & 3) + 8 * (e >> 2) + 4 * half.
    // v = acc * descale + bias; the next layer's BN + ReLU; x H2_ASCALE; hi / lo planes; 8 bytes (4 channels) per plane and lane
    const int half = lane >> 5, c32 = lane & 31;
    const int g = a.outg[j];
    unsigned top = 0u;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
        const int m = mt * 256 + nh * 128 + wn * 64 + bi * 32 + c32;     // match
        const bool ok = (m >> 5) < a.nT32 && (m >> 4) < a.nTiles16;
#pragma unroll
        for (int ai = 0; ai < 4; ++ai) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int o = mtile * 256 + wm * 128 + ai * 32 + q4 * 8 + half * 4;
                const floatx4 b4 = *reinterpret_cast<const floatx4*>(a.bias + o);
                const floatx4 s4 = *reinterpret_cast<const floatx4*>(a.bn_s + o);
                const floatx4 t4 = *reinterpret_cast<const floatx4*>(a.bn_t + o);
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[ai][bi][4 * q4 + e] * a.descale + b4[e];
                    y[e] = fmaxf(v * s4[e] + t4[e], 0.f) * H2_ASCALE;
                    top = max(top, __float_as_uint(y[e]) & 0x7FFFFFFFu);
                }
                unsigned h0, l0, h1, l1;
                cg_split_pair(y[0], y[1], h0, l0);
                cg_split_pair(y[2], y[3], h1, l1);
                if (ok) {
                    char* dst = a.out + (((size_t)(m >> 4) * a.c8out + (o >> 3)) * 2) * 15360 + g * 256 + (m & 15) * 16 + half * 8;
                    *reinterpret_cast<uint2*>(dst) = uint2{h0, h1};
                    *reinterpret_cast<uint2*>(dst + 15360) = uint2{l0, l1};
                }
            }
        }
    }
    note_range_bits(a.rflag, top, FP16_MAX);
}

int cgemm_init() {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cgemm_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&cgemm_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS));
    return 0;
}

// the cone layer L (cin -> cout, 13 taps) on M matches: Bstages as launch_gft16_invg left them, out = cone1's input planes;
// slot[j * 13 + k] = cone slot of N[n_j][k], outg[j] = n_j; amax != null selects the fp8-correction kernel (needs L.wcg8)
int launch_cgemm(const Layer& L, const char* Bstages, int nslot, const unsigned char* slot, const unsigned char* outg, int nT32, int nTiles16,
                 char* out, hipStream_t s, int* rflag, const unsigned* amax) {
    const bool fp8 = amax != nullptr;
    if (!(fp8 ? L.wcg8 : L.wcg) || L.cin % 32 || L.cout % 256) { set_error("cgemm: needs the cone-GEMM weight pack, cin %% 32 == 0, cout %% 256 == 0"); return YOHO_EINVAL; }
    if (nT32 == 0) return 0;
    CGemmArgs a;
    a.A = reinterpret_cast<const char*>(fp8 ? L.wcg8 : L.wcg); a.B = Bstages; a.bias = L.bias; a.bn_s = L.bn_s; a.bn_t = L.bn_t; a.out = out;
    a.NTm = (nT32 + 7) / 8; a.MT = L.cout / 256; a.KSt = L.cin / 32; a.ntap = 13; a.nj = 13; a.nslot = nslot; a.c8out = L.cout / 8;
    a.nT32 = nT32; a.nTiles16 = nTiles16; a.descale = L.wcg_descale; a.rflag = rflag; a.amax = amax;
    std::memcpy(a.slot, slot, 169);
    std::memset(a.outg, 0, 16);
    std::memcpy(a.outg, outg, 13);
    const int CT = a.nj * a.NTm;
    const int grid = 8 * ((CT + 7) / 8) * a.MT;
    if (fp8) hipLaunchKernelGGL(cgemm_kernel<true>, dim3(grid), dim3(512), F3_LDS, s, a);
    else hipLaunchKernelGGL(cgemm_kernel<false>, dim3(grid), dim3(512), F3_LDS, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// fgemm3s: fgemm3's staging and K loop for layers with 32 output channels (32 d <= 160 live rows of the 256-row A tile).  In
// fgemm3 the waves are arranged 2 (rows) x 4 (columns): the lower row half is padding, so half the waves - two of the four
// SIMDs - idle and the other two issue 24 MFMAs per step on 128 rows of which 32 d are real.  Here the eight waves split the
// 256 columns (32 each) and every wave holds the d live 32-row tiles: 3 d MFMAs per wave and step on all four SIMDs, the same
// products in the same order per accumulator (bit-identical coefficients).
// ---------------------------------------------------------------------------------------------------------------
struct Frags3s {
    uintx4 ah[5], al[5], bh, bl;
};

// fragment read R (0..11) of a step, in the order of use: A lo x5, B hi, A hi x5, B lo
template <int R>
__device__ __forceinline__ void read_frag3s(const char* pa, const char* pb, Frags3s& f) {
    if constexpr (R < 5) f.al[R] = *reinterpret_cast<const uintx4*>(pa + 8192 + R * 512);
    else if constexpr (R == 5) f.bh = *reinterpret_cast<const uintx4*>(pb);
    else if constexpr (R < 11) f.ah[R - 6] = *reinterpret_cast<const uintx4*>(pa + (R - 6) * 512);
    else f.bl = *reinterpret_cast<const uintx4*>(pb + 8192);
}

// One K16 step: 3 d MFMAs on `f` (lo.hi, hi.lo, hi.hi per accumulator, as fgemm3); behind them the 12 fragment reads of the next
// step (rows beyond 32 d are padding: read, never used) and the wave's four DMA pieces of the step three ahead.
template <bool DMA, int D>
__device__ __forceinline__ void step3s(const Frags3s& f, floatx16 (&acc)[5], const char* ra, const char* rb, Frags3s& nf,
                                       const char* srcA, const char* srcB, char* dmabuf, int la, int ldst) {
    sfor<0, 5>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g < D) acc[g] = mfma_h(f.al[g], f.bh, acc[g]);
        if constexpr (g < D) read_frag3s<g>(ra, rb, nf);
        if constexpr (g == 4) read_frag3s<5>(ra, rb, nf);
        __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, 5>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g < D) acc[g] = mfma_h(f.ah[g], f.bl, acc[g]);
        if constexpr (g < D) read_frag3s<6 + g>(ra, rb, nf);
        if constexpr (g == 4) read_frag3s<11>(ra, rb, nf);
        __builtin_amdgcn_sched_barrier(0);
    });
    sfor<0, 5>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g < D) acc[g] = mfma_h(f.ah[g], f.bh, acc[g]);
        if constexpr (DMA && g < 4) dma_piece3<g>(srcA, srcB, dmabuf, la, ldst);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// K loop of fgemm3s for an irrep of dimension D (compile time: no branches around the MFMAs, no reads of padding rows)
template <int D>
__device__ __forceinline__ void fgemm3s_kloop(floatx16 (&acc)[5], char* smem, const char* Ag, const char* Bg, int KT, int la, int ldst,
                                              int lane_a, int lane_b) {
    Frags3s P, Q;
    sfor<0, 12>([&](auto rc) {
        constexpr int R = decltype(rc)::value;
        if constexpr (R == 5 || R == 11 || (R < 5 && R < D) || (R > 5 && R < 11 && R - 6 < D)) read_frag3s<R>(smem + lane_a, smem + lane_b, P);
    });
    // Every wave must have step 0 in registers before ANY wave overwrites buffer 0 with step 3 (the first DMA of the loop below):
    // inside the loop the barrier at the end of step s - 1 orders the reads of step s before the DMA of step s + 3, but between
    // the prologue's barrier and the first step there was nothing - a wave delayed by a few hundred cycles (cold instruction
    // cache at the head of a launch) read step 3's rows in place of step 0's.  Found in round 3 by the depth-first schedule
    // (many short launches), whose outputs differed from the breadth-first pass in whole 32-keypoint wave tiles.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    int cur = 0, nxt = F3_BUF;                   // ring offsets: step s lives in buffer s % 3
    int s = 0;
    auto advance = [&]() { cur = nxt; nxt = nxt == 2 * F3_BUF ? 0 : nxt + F3_BUF; ++s; };
    auto sync4 = [] {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto sync0 = [] {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // main loop: both steps of a pair still have a DMA to issue (s + 4 < KT)
    while (s + 4 < KT) {
        step3s<true, D>(P, acc, smem + nxt + lane_a, smem + nxt + lane_b, Q, uniform_ptr(Ag + step_off(s + 3)), uniform_ptr(Bg + step_off(s + 3)),
                     smem + cur, la, ldst);
        sync4();
        advance();
        step3s<true, D>(Q, acc, smem + nxt + lane_a, smem + nxt + lane_b, P, uniform_ptr(Ag + step_off(s + 3)), uniform_ptr(Bg + step_off(s + 3)),
                     smem + cur, la, ldst);
        sync4();
        advance();
    }
    // tail: the last four steps (two if KT == 2); only the first of them still has a DMA (step KT - 1) to issue
    if (KT >= 4) {
        step3s<true, D>(P, acc, smem + nxt + lane_a, smem + nxt + lane_b, Q, uniform_ptr(Ag + step_off(s + 3)), uniform_ptr(Bg + step_off(s + 3)),
                     smem + cur, la, ldst);
        sync4();
        advance();
        step3s<false, D>(Q, acc, smem + nxt + lane_a, smem + nxt + lane_b, P, nullptr, nullptr, nullptr, la, ldst);
        sync0();
        advance();
    }
    step3s<false, D>(P, acc, smem + nxt + lane_a, smem + nxt + lane_b, Q, nullptr, nullptr, nullptr, la, ldst);
    sync0();
    advance();
    // last step: its "next" fragments are read from a buffer that holds valid (unused) data
    step3s<false, D>(Q, acc, smem + cur + lane_a, smem + cur + lane_b, P, nullptr, nullptr, nullptr, la, ldst);

}

__global__ __launch_bounds__(512, 2) void fgemm3s_kernel(FGemmArgs a, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    int t = 0, local = 0, r = 0;
    if (!fg3_map(a, blockIdx.x & 7, blockIdx.x >> 3, t, local, r)) return;
    const int d = a.dim[t], qbase = a.qbase[t];
    const int KS = d * a.cin / 32, KT = 2 * KS;                        // MT = 1: the tile holds all 32 d rows
    const int ntile = r + 8 * local;
    const char* Ag = a.A + a.a_off[t];
    const char* Bg = a.B + a.b_off[t] + (size_t)ntile * KS * FG_STAGE;
    const int wkg = w8 >> 2, wcb = w8 & 3;                               // this wave's DMA pieces: k-group, 64-row / 64-column block
    const int la = wkg * 4096 + wcb * 1024 + lane * 16;
    const int ldst = wkg * 4096 + wcb * 1024;

    floatx16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    const int lane_a = (lane >> 5) * 4096 + (lane & 31) * 16;
    const int lane_b = 16384 + (lane >> 5) * 4096 + (w8 * 32 + (lane & 31)) * 16;

    auto dma = [&](int st, char* buf) {
        const char* sa = uniform_ptr(Ag + step_off(st));
        const char* sb = uniform_ptr(Bg + step_off(st));
        sfor<0, 4>([&](auto uc) { dma_piece3<decltype(uc)::value>(sa, sb, buf, la, ldst); });
    };
    dma(0, smem);
    dma(1, smem + F3_BUF);
    if (KT > 2) dma(2, smem + 2 * F3_BUF);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    switch (d) {
        case 1: fgemm3s_kloop<1>(acc, smem, Ag, Bg, KT, la, ldst, lane_a, lane_b); break;
        case 3: fgemm3s_kloop<3>(acc, smem, Ag, Bg, KT, la, ldst, lane_a, lane_b); break;
        case 4: fgemm3s_kloop<4>(acc, smem, Ag, Bg, KT, la, ldst, lane_a, lane_b); break;
        default: fgemm3s_kloop<5>(acc, smem, Ag, Bg, KT, la, ldst, lane_a, lane_b); break;
    }

    // ---- epilogue: D[row][col]: lane (col = lane & 31, half = lane >> 5), reg e -> row = (e & 3) + 8 * (e >> 2) + 4 * half
    const int half = lane >> 5, kp32 = lane & 31;
    const int cout8 = a.cout >> 3;                                       // = 4: row tile ai is coefficient row i = ai
    const bool addb = (d == 1);                                          // trivial irrep: coefficient 0 carries sqrt(60) * bias
    const int colbase = ntile * 256 + w8 * 32;
    const int jidx = colbase / a.kppad, kp0 = colbase - jidx * a.kppad;
    const int tile32 = kp0 >> 5;
    unsigned top = 0u;
    if (tile32 < a.nT32) {
#pragma unroll
        for (int ai = 0; ai < 5; ++ai) {
            if (ai >= d) continue;
            const int q = qbase + ai * d + jidx;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int o = q4 * 8 + half * 4;
                floatx4 val;
                val.x = acc[ai][4 * q4 + 0]; val.y = acc[ai][4 * q4 + 1];
                val.z = acc[ai][4 * q4 + 2]; val.w = acc[ai][4 * q4 + 3];
                val *= a.descale;
                if (addb) val += *reinterpret_cast<const floatx4*>(a.bias + o) * 7.745966692414834f;
                const size_t off = (((((size_t)tile32 * G + q) * cout8 + (o >> 3)) * 2 + half) * TILE + kp32) * 4;
                if (!(flags & F2_NOSTORE)) *reinterpret_cast<floatx4*>(a.out + off) = val;
                top = max(max(top, __float_as_uint(val.x) & 0x7FFFFFFFu), __float_as_uint(val.y) & 0x7FFFFFFFu);
                top = max(max(top, __float_as_uint(val.z) & 0x7FFFFFFFu), __float_as_uint(val.w) & 0x7FFFFFFFu);
            }
        }
    }
    note_range_bits(a.rflag, top, FP16_MAX / HF_ASCALE);
}

int fgemm3_init() {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm3s_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm3c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F3_LDS));
    return 0;
}

int launch_fgemm3(const FGemmArgs& a, int flags, hipStream_t s) {
    if (const char* dbg = experiment_env("YOHO_FGEMM_DEBUG")) {
        if (std::strstr(dbg, "nostore")) flags |= F2_NOSTORE;
        if (std::strstr(dbg, "sc1")) flags |= F2_ST_SC1;            // round 6: the coefficient stores write-through ...
        if (std::strstr(dbg, "stnt")) flags |= F2_ST_NT;            // ... or non-temporal: produced once, consumed once by gft16x
    }
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
    // 32 output channels, no residual: all 32 d live rows in every wave (fgemm3s); YOHO_FGEMM_DEBUG=nosmall keeps fgemm3
    static const bool small_ok = [] { const char* e = experiment_env("YOHO_FGEMM_DEBUG"); return !(e && std::strstr(e, "nosmall")); }();
    if (small_ok && a.cout == 32 && !(flags & EPI_RES)) hipLaunchKernelGGL(fgemm3s_kernel, dim3(tot), dim3(512), F3_LDS, s, a, flags);
    else if (a.amax) hipLaunchKernelGGL(fgemm3c_kernel, dim3(tot), dim3(512), F3_LDS, s, a, flags);      // gconv_mode 7: corrections on the fp8 pipe
    else hipLaunchKernelGGL(fgemm3_kernel, dim3(tot), dim3(512), F3_LDS, s, a, flags);
    HIPCHK(hipGetLastError());
    return 0;
}

int fgemm2_init() {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fgemm2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F2_LDS));
    return fgemm3_init();
}

int launch_fgemm2(const FGemmArgs& a, int flags, hipStream_t s) {
    if (const char* dbg = experiment_env("YOHO_FGEMM_DEBUG")) {          // kernel-timing experiments only (results are not valid with nostore)
        if (std::strstr(dbg, "nostore")) flags |= F2_NOSTORE;       // no coefficient stores at all (K loops alone)
        if (std::strstr(dbg, "mix")) flags |= F2_MIX;               // slots alternate between the d = 5 irrep and the others
        if (std::strstr(dbg, "sc1")) flags |= F2_ST_SC1;            // write-through stores that do not stay in the L2
        if (std::strstr(dbg, "nt")) flags |= F2_ST_NT;              // non-temporal stores
        if (std::strstr(dbg, "sparse4")) flags |= F2_SPARSE4;
        if (std::strstr(dbg, "sparse16")) flags |= F2_SPARSE16;
    }
    int tot = 0;
    for (int x = 0; x < 8; ++x) {
        int n = 0;
        for (int t = 0; t < NIR_ORD; ++t) {
            const int r = (x + a.rot[t]) & 7;
            if (a.NT[t] > r) n += ((a.NT[t] - 1 - r) / 8 + 1) * a.MT[t] * 2;
        }
        tot = n > tot ? n : tot;
    }
    tot *= 8;
    hipLaunchKernelGGL(fgemm2_kernel, dim3(tot), dim3(256), F2_LDS, s, a, flags);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
