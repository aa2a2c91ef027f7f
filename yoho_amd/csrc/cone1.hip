// PartII, last 13-tap layer (utils/network.py:272-274 evaluated at the identity only): the quaternion head reads
// group element 0 of this layer's output, so it is one dense product over (tap, channel):
//     out[o, m] = sum_k sum_c W[o,c,k] * act[c, N[0][k], m]            K = 13 * 512
// on the fp16x2 split MFMA.  Input = the activated 13-cone planes the previous layer left behind
// ([tile16][c8][plane][60][16 kp][8 ch] fp16, only the slabs N[0][k] are valid).  A workgroup owns 32 matches (two
// 16-match tiles = the two column halves of the MFMA) and 128 output channels (one 32-row block per wave); per
// 8-channel chunk only the 13 needed slabs of both tiles and planes are gathered into LDS (14 KiB, ring of three).  In front of the
// one-launch tail (gconv.hip: mlp_head_kernel) the 64 channel chunks are walked by TWO workgroups per tile (a fixed split) that
// leave raw partial sums; the tail adds them: a 1000-match pass then has 128 chains of 32 dependent chunks instead of 64 of 64
// (71 -> ~40 us); at 3233 matches (408 workgroups) it makes no difference.
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
    float* part;           // K split (gridDim.y = 2): raw partial sums [half of K][tile32][cout8][h][kp32][4], no descale / bias / residual
    float descale;
    int n0[14];            // N[0][k]; n0[13] = n0[0] (tap 13 has zero weights, its slab only has to be finite)
};

__device__ __forceinline__ floatx16 mfma_c1(uintx4 a, uintx4 b, floatx16 c) {
    union { uintx4 u; halfx8 h; } ca, cb;
    ca.u = a; cb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ca.h, cb.h, c, 0, 0, 0);
}

// compile-time loop
template <int I, int N, typename Fn>
__device__ __forceinline__ void sfor_c1(Fn&& fn) {
    if constexpr (I < N) { fn(std::integral_constant<int, I>{}); sfor_c1<I + 1, N>(fn); }
}

__global__ __launch_bounds__(256, 1) void cone1_kernel(Cone1Args a) {       // <= 204 workgroups for 3233 matches: one per CU
    __shared__ __attribute__((aligned(16))) char smem[3 * C1_STAGE];
    __shared__ int n0s[16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nog = a.cout8 / 16;                         // groups of 128 output channels
    // K split: workgroup (x, y) walks the channel chunks [y cn, (y + 1) cn).  The split is FIXED (never a function of the number of
    // matches), so a match's result does not depend on which other matches share its pass.
    const int cn = a.cin8 / (int)gridDim.y, c0 = (int)blockIdx.y * cn;
    const int tile32 = blockIdx.x / nog, og = blockIdx.x - tile32 * nog;
    const int ob = og * 4 + w;
    if (tid < 14) n0s[tid] = a.n0[tid];
    __syncthreads();

    // LDS-DMA gather: unit u (1 KiB) = slots 4u .. 4u+3 of [plane][half][tap]; this wave owns units w, w+4, w+8, w+12
    long long soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = w + 4 * i < C1_UNITS ? w + 4 * i : w + 4 * (i - 1);      // waves 2, 3 own three units: the fourth piece repeats the third
        const int slot = u * 4 + (lane >> 4);
        const int pl = slot / 28, r = slot - pl * 28, ht = r / 14, k = r - ht * 14;
        int t16 = 2 * tile32 + ht;
        t16 = t16 < a.nTiles16 ? t16 : a.nTiles16 - 1;
        soff[i] = (long long)t16 * a.cin8 * C1_CHUNK + pl * C1_PLANE + n0s[k] * 256 + (lane & 15) * 16;
    }
    auto stage = [&](int c8, char* dst) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (same bytes to the same place for the repeated piece: every wave issues four, so the wait counts are uniform and the
            // loop body has no branch)
            const int u = w + 4 * i < C1_UNITS ? w + 4 * i : w + 4 * (i - 1);
            __builtin_amdgcn_global_load_lds((gptr_t)(a.X + soff[i] + (long long)(c0 + c8) * C1_CHUNK), (lptr_t)(dst + u * 1024), 16, 0, 0);
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

    // Pipeline.  A workgroup walks 64 channel chunks of 21 MFMAs (0.3 us) each, and it is alone on its CU (<= 204 workgroups per
    // pair), so what it waits for is latency: the X slabs (LDS-DMA, from HBM) and the weight planes (registers, from the L2) are
    // requested TWO chunks ahead (rings of three).  In iteration c the wave issues W(c+2) and then DMA(c+2); at the top of
    // iteration c everything older than the requests of chunk c+1 has to have landed: a counted wait (vector-memory
    // operations complete in issue order), through the builtin so that the compiler's scoreboard sees it.  The B fragments are
    // read through asm: in front of a C++ LDS load the compiler waits for every pending LDS DMA (vmcnt(0)), which serialised
    // this loop before (128 us -> see DESIGN 3.1b).
    constexpr int RING = 3;
    uintx4 wreg[RING][7][2];
    auto loadW = [&](int c8, uintx4 (&wr)[7][2]) {
        const uintx4* Wn = Wl + (size_t)(c0 + c8) * (7 * 2 * 64);
#pragma unroll
        for (int tp = 0; tp < 7; ++tp) { wr[tp][0] = Wn[(tp * 2) * 64]; wr[tp][1] = Wn[(tp * 2 + 1) * 64]; }
    };
    auto compute = [&](const char* cur, const uintx4 (&wr)[7][2]) {
        // all 14 B fragments of the chunk are requested at once, one wait (tied to the registers, so no MFMA is scheduled in front
        // of it), then the 21 MFMAs
        const unsigned base = (unsigned)(size_t)(cur + lane_b);
        uintx4 bh[7], bl[7];
#pragma unroll
        for (int tp = 0; tp < 7; ++tp)
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                         : "=&v"(bh[tp]), "=&v"(bl[tp]) : "v"(base), "n"(tp * 512), "n"(tp * 512 + 2 * 14 * 256) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]), "+v"(bh[4]), "+v"(bh[5]), "+v"(bh[6]),
                       "+v"(bl[0]), "+v"(bl[1]), "+v"(bl[2]), "+v"(bl[3]), "+v"(bl[4]), "+v"(bl[5]), "+v"(bl[6]) :: "memory");
#pragma unroll
        for (int tp = 0; tp < 7; ++tp) {
            acc[0] = mfma_c1(wr[tp][1], bh[tp], acc[0]);  // three independent accumulation chains
            acc[1] = mfma_c1(wr[tp][0], bl[tp], acc[1]);
            acc[2] = mfma_c1(wr[tp][0], bh[tp], acc[2]);
        }
    };
    // the requests of one chunk stay in flight: 14 weight loads + 4 DMA pieces
    auto top = [&]() {
        __builtin_amdgcn_s_waitcnt(0x4F72);                              // vmcnt(18) = 0b010010: low nibble 2, bits 15:14 = 1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stage_of = [&](int c8) { return smem + (c8 % RING) * C1_STAGE; };
    const int last = cn - 1;
    auto clamp = [&](int c8) { return c8 <= last ? c8 : last; };
    // prologue: chunks 0, 1, 2 (in the order the loop keeps: W then DMA)
    loadW(clamp(0), wreg[0]); stage(clamp(0), stage_of(0));
    loadW(clamp(1), wreg[1]); stage(clamp(1), stage_of(1));
    const int nmain = (cn / RING) * RING;
    for (int c8 = 0; c8 < nmain; c8 += RING) {
        sfor_c1<0, RING>([&](auto jc) {
            constexpr int jj = decltype(jc)::value;
            const int c = c8 + jj;                                       // no branch in here: the compiler's wait bookkeeping stays
            top();                                                       // exact on straight-line code
            loadW(clamp(c + RING - 1), wreg[(jj + RING - 1) % RING]);
            stage(clamp(c + RING - 1), stage_of(c + RING - 1));
            compute(stage_of(c), wreg[jj]);
        });
    }
    // the cin8 % RING chunks left (their requests are already in flight; the extra requests keep the wait counts valid)
    sfor_c1<0, RING - 1>([&](auto jc) {
        constexpr int jj = decltype(jc)::value;
        if (nmain + jj <= last) {
            top();
            loadW(last, wreg[(jj + RING - 1) % RING]);
            stage(last, stage_of(nmain + jj + RING - 1));
            compute(stage_of(nmain + jj), wreg[jj]);
        }
    });
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
        if (gridDim.y > 1) {                                 // the consumer (mlp_head_kernel) adds the halves, then descale, bias, residual
            *reinterpret_cast<floatx4*>(a.part + ((((size_t)blockIdx.y * a.nTiles32 + tile32) * a.cout8 + ob * 4 + q) * 64 + lane) * 4) = val;
            continue;
        }
        val *= a.descale;
        val += *reinterpret_cast<const floatx4*>(a.bias + ch);
        const size_t off = (((((size_t)tile32 * a.cout8 + ob * 4 + q) * G + 0) * 2 + half) * TILE + kp32) * 4;
        if (a.res) val += *reinterpret_cast<const floatx4*>(a.res + off);
        *reinterpret_cast<floatx4*>(a.out + off) = val;
    }
}

// part != null: K over two workgroups, partial sums to `part` (2 * nTiles32 * cout8 * 256 floats) instead of the finished `out`
int launch_cone1(const Layer& L, const char* X, int nTiles32, int nTiles16, const float* res, float* out, const int* n0, hipStream_t s, float* part) {
    if (L.cout_pad % 128 || !L.wph || L.cin % 32) { set_error("cone1: needs cout %% 128 == 0, cin %% 32 == 0 and fp16x2 weights"); return YOHO_EINVAL; }
    Cone1Args a;
    a.X = X; a.Wp = reinterpret_cast<const char*>(L.wph); a.bias = L.bias; a.res = res; a.out = out; a.part = part;
    if (part && (L.cin / 8) % 2) { set_error("cone1: K split needs an even number of channel chunks"); return YOHO_EINVAL; }
    a.nTiles32 = nTiles32; a.nTiles16 = nTiles16; a.cin8 = L.cin / 8; a.cout8 = L.cout_pad / 8; a.descale = L.wph_descale;
    for (int k = 0; k < 13; ++k) a.n0[k] = n0[k];
    a.n0[13] = n0[0];
    if (nTiles32 == 0) return 0;
    hipLaunchKernelGGL(cone1_kernel, dim3(nTiles32 * (a.cout8 / 16), part ? 2 : 1), dim3(256), 0, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
