// Icosahedral group convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Replaces, per layer, the reference's  x[:,:,Nei] gather -> (B,C,60,13) -> Conv2d(C,Cout,(1,13))
// (utils/network.py:46-52,80-84 data_process; :12-21 Comb_Conv; :23-65 Residual_Comb_Conv).
//
//   out[b, o, g] = bias[o] + sum_k sum_c W[o, c, 0, k] * act[b, c, N[g, k]]
//
// MI355X formulation.  A tile is 32 keypoints.  For a tile, every group element g' owns a
// "slab" act[:, g', c-chunk] of shape (32 kp x 8 ch).  The conv is then, for every output g and
// tap k, a 32(o) x 32(kp) x 8(c) matrix product whose B operand is simply slab N[g,k]: the gather
// index is wave-uniform, no per-lane gather exists anywhere.  D = W_tap (32 o x K) * slab (K x 32 kp)
// so a lane ends up with 4 consecutive channels of one keypoint per accumulator quad, which is
// exactly the [c8][g][h][kp][4] layout the next layer stages into LDS with linear 16-B DMA.
//
// Workgroup = 4 waves, one per SIMD, 1 workgroup per CU (accumulators: 15 g x 16 regs per wave).
//   wave w owns output slots [w*GPW, (w+1)*GPW)  (GPW=15: all 60 group elements)
//   or, in OSPLIT mode (single output group element), wave w owns o-block 4*blk + w.
// LDS: two 60 KiB buffers holding one 8-channel chunk of all 60 slabs (double buffered,
// filled by global_load_lds DMA while the previous chunk is being multiplied).
// Weights are pre-packed in MFMA A-fragment order, one contiguous 1 KiB per (o-block, c8, tap),
// and stream L2 -> VGPR with a one-tap-ahead prefetch.
#include "common.h"

namespace yoho {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int CHUNK_BYTES = CHUNK_FLOATS * 4;         // 61440
constexpr int LDS_BYTES = 2 * CHUNK_BYTES;            // 122880

// Slot tables: device memory of the context, addressed through the constant address space so that the wave-uniform lookups are
// scalar (s_load) and never touch the vector-memory counter the LDS DMA and the weight prefetch depend on.
//   slabtab[cfg][tap][slot] = LDS byte offset of input slab N[g(slot), tap];  outg[cfg][slot] = g or -1
typedef const __attribute__((address_space(4))) int* cint_p;

int upload_slot_tables(const int* slab_h, const int* outg_h, SlotTables& t) {
    HIPCHK(hipMalloc((void**)&t.slabtab, sizeof(int) * NCFG * NTAP * G));
    HIPCHK(hipMalloc((void**)&t.outg, sizeof(int) * NCFG * G));
    HIPCHK(hipMemcpy(t.slabtab, slab_h, sizeof(int) * NCFG * NTAP * G, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(t.outg, outg_h, sizeof(int) * NCFG * G, hipMemcpyHostToDevice));
    return 0;
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// stage `npieces` 1 KiB pieces of a chunk: piece p -> LDS bytes [p*1024, p*1024+1024)
__device__ __forceinline__ void stage_chunk(const float* src, char* dst, int w, int lane, int npieces) {
    for (int p = w; p < npieces; p += 4) {
        const float* s = src + p * SLAB_FLOATS + lane * 4;
        char* d = dst + p * 1024;      // wave-uniform; hardware adds lane*16
        __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)d, 16, 0, 0);
    }
}

template <int GPW, bool OSPLIT>
__global__ __launch_bounds__(256, 1) void gconv_kernel(ConvArgs a, int flags, int nstage, int cfg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware bijective block remap: blocks b, b+8, b+16.. (same XCD) get consecutive work ids,
    // so the workgroups sharing one tile's activations (different o-blocks) share an L2.
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7, slot8 = b >> 3;
    const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot8;
    const int tile = v / a.nOB;
    const int obk = v - tile * a.nOB;
    const int ob = OSPLIT ? obk * 4 + w : obk;            // this wave's 32-channel output block
    const int ws = OSPLIT ? 0 : w;                        // this wave's slot group

    floatx16 acc[GPW];
#pragma unroll
    for (int j = 0; j < GPW; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    const float* Xt = a.X + (size_t)tile * a.cin8 * CHUNK_FLOATS;
    const int ntaps = a.ntaps;
    const int total = a.cin8 * ntaps;
    const floatx4* Wb = reinterpret_cast<const floatx4*>(a.Wp) + (size_t)ob * total * 64 + lane;

    stage_chunk(Xt, smem, w, lane, nstage);
    floatx4 wnext = Wb[0];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int it = 0;
    for (int c8 = 0; c8 < a.cin8; ++c8) {
        if (c8 + 1 < a.cin8)
            stage_chunk(Xt + (size_t)(c8 + 1) * CHUNK_FLOATS, smem + ((c8 + 1) & 1) * CHUNK_BYTES, w, lane, nstage);
        const char* xb = smem + (c8 & 1) * CHUNK_BYTES + lane * 16;
        for (int tap = 0; tap < ntaps; ++tap) {
            const floatx4 wc = wnext;
            ++it;
            if (it < total) wnext = Wb[(size_t)it * 64];
            cint_p st = (cint_p)a.slabtab + cfg * (NTAP * G) + (tap * 4 + ws) * GPW;
#pragma unroll
            for (int j = 0; j < GPW; ++j) {
                const int so = st[j];                                   // wave-uniform (scalar load)
                const floatx4 xf = *reinterpret_cast<const floatx4*>(xb + so);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.x, xf.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.y, xf.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.z, xf.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.w, xf.w, acc[j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: D[i=o][j=kp]; lane (kp = lane&31, half = lane>>5), reg r -> o = (r&3) + 8*(r>>2) + 4*half
    const int kp = lane & 31, half = lane >> 5;
    cint_p og = (cint_p)a.outg + cfg * G + ws * GPW;
#pragma unroll
    for (int j = 0; j < GPW; ++j) {
        const int g = og[j];
        if (g < 0) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = ob * 32 + q * 8 + half * 4;
            floatx4 val;
            val.x = acc[j][4 * q + 0]; val.y = acc[j][4 * q + 1];
            val.z = acc[j][4 * q + 2]; val.w = acc[j][4 * q + 3];
            val += *reinterpret_cast<const floatx4*>(a.bias + ch);
            const size_t off = (((((size_t)tile * a.cout8 + ob * 4 + q) * G + g) * 2 + half) * TILE + kp) * 4;
            if (flags & EPI_RES) val += *reinterpret_cast<const floatx4*>(a.res + off);
            if (flags & EPI_RAW) *reinterpret_cast<floatx4*>(a.out_raw + off) = val;
            if (flags & EPI_ACT) {
                const floatx4 s = *reinterpret_cast<const floatx4*>(a.bn_s + ch);
                const floatx4 t = *reinterpret_cast<const floatx4*>(a.bn_t + ch);
                floatx4 y = val * s + t;
                y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
                *reinterpret_cast<floatx4*>(a.out_act + off) = y;
            }
        }
    }
}

template <int GPW, bool OSPLIT>
static int launch_t(const ConvArgs& a, int flags, int nstage, int cfg, hipStream_t s) {
    const int grid = a.nTiles * a.nOB;
    hipLaunchKernelGGL((gconv_kernel<GPW, OSPLIT>), dim3(grid), dim3(256), LDS_BYTES, s, a, flags, nstage, cfg);
    HIPCHK(hipGetLastError());
    return 0;
}

template <int GPW, bool OSPLIT>
static int init_t() {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gconv_kernel<GPW, OSPLIT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    return 0;
}

// PartII's tail in one launch (utils/network.py:273-277: the three 1x1 convolutions on the feature at the identity, 256 -> 512 ->
// 128 -> 4, each followed by BN + ReLU except the last, then q / |q|): one workgroup per 32-match tile keeps the activations in
// LDS between the layers.  As three gconv_kernel<1, .> launches + quat_norm the tail was a chain of 80 workgroup barriers with a
// weight load behind each (32 + 64 + 16 eight-channel chunks): 80 us per 1000 matches for 0.4 GFLOP.  Same fp32 MFMAs in the same
// order per output (chunks ascending, x y z w inside a chunk) and the same epilogue expressions: given the same input, the bits of
// the staged path.  In the default PartII mode the input arrives as the two K halves of cone1_kernel and is finished while it is
// staged (the staged path's cone1 sums its K in one chain: the last bits of its quaternions differ).
struct MlpArgs {
    const float* X;                                        // [tile][cinA/8][60 slabs][256]: slab 0 = the group identity
    const float *WA, *biasA, *sA, *tA;                     // 256 -> 512, BN + ReLU
    const float *WB, *biasB, *sB, *tB;                     // 512 -> 128, BN + ReLU
    const float *WC, *biasC;                               // 128 -> 32 (4 used), raw
    int cinA8, obA, obB, M;
    float* quat;
    // X not finished yet: the producing layer (cone1_kernel with its K over two workgroups) left two partial sums
    // [2][tile][cinA/8][256]; X = (p0 + p1) * descale + bias + res (res in X's layout), formed while staging
    const float* part; const float* biasP; const float* resP; float descaleP; int nTiles;
};

__device__ __forceinline__ floatx4 quat_unit(floatx4 q) {
#pragma clang fp contract(off)                             // layout.hip:quat_norm_kernel's arithmetic (compiled without contraction)
    const float n = sqrtf(((q.x * q.x + q.y * q.y) + q.z * q.z) + q.w * q.w);
    floatx4 o;
    o.x = q.x / n; o.y = q.y / n; o.z = q.z / n; o.w = q.w / n;
    return o;
}

__global__ __launch_bounds__(256, 1) void mlp_head_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x;
    char* sX = smem;
    char* sA = sX + a.cinA8 * 1024;
    char* sB = sA + a.obA * 4 * 1024;
    const int half = lane >> 5;
    if (a.part) {
        for (int p = w; p < a.cinA8; p += 4) {
            const size_t o0 = (((size_t)tile * a.cinA8 + p) * 64 + lane) * 4, o1 = o0 + (size_t)a.nTiles * a.cinA8 * 256;
            floatx4 val = *reinterpret_cast<const floatx4*>(a.part + o0) + *reinterpret_cast<const floatx4*>(a.part + o1);
            val *= a.descaleP;
            val += *reinterpret_cast<const floatx4*>(a.biasP + p * 8 + half * 4);
            if (a.resP) val += *reinterpret_cast<const floatx4*>(a.resP + ((size_t)tile * a.cinA8 + p) * CHUNK_FLOATS + lane * 4);
            *reinterpret_cast<floatx4*>(sX + p * 1024 + lane * 16) = val;
        }
    } else {
        const float* Xt = a.X + (size_t)tile * a.cinA8 * CHUNK_FLOATS;
        for (int p = w; p < a.cinA8; p += 4)
            __builtin_amdgcn_global_load_lds((gptr_t)(Xt + (size_t)p * CHUNK_FLOATS + lane * 4), (lptr_t)(sX + p * 1024), 16, 0, 0);
    }
    // ---- layer A: wave w owns the output blocks w, w + 4, w + 8, w + 12 (one fragment of the input serves four accumulators)
    {
        floatx16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        const floatx4* W0 = reinterpret_cast<const floatx4*>(a.WA) + (size_t)w * a.cinA8 * 64 + lane;
        const size_t wstride = (size_t)4 * a.cinA8 * 64;
        floatx4 wn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) wn[j] = W0[j * wstride];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int c8 = 0; c8 < a.cinA8; ++c8) {
            floatx4 wc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) wc[j] = wn[j];
            if (c8 + 1 < a.cinA8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wn[j] = W0[j * wstride + (size_t)(c8 + 1) * 64];
            }
            const floatx4 xf = *reinterpret_cast<const floatx4*>(sX + c8 * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[j].x, xf.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[j].y, xf.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[j].z, xf.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[j].w, xf.w, acc[j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ob = w + 4 * j;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = ob * 32 + q * 8 + half * 4;
                floatx4 val;
                val.x = acc[j][4 * q + 0]; val.y = acc[j][4 * q + 1];
                val.z = acc[j][4 * q + 2]; val.w = acc[j][4 * q + 3];
                val += *reinterpret_cast<const floatx4*>(a.biasA + ch);
                const floatx4 sc = *reinterpret_cast<const floatx4*>(a.sA + ch);
                const floatx4 sh = *reinterpret_cast<const floatx4*>(a.tA + ch);
                floatx4 y = val * sc + sh;
                y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
                *reinterpret_cast<floatx4*>(sA + (ob * 4 + q) * 1024 + lane * 16) = y;
            }
        }
    }
    __syncthreads();
    // ---- layer B: wave w owns output block w
    {
        floatx16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int cin8 = a.obA * 4;
        const floatx4* W0 = reinterpret_cast<const floatx4*>(a.WB) + (size_t)w * cin8 * 64 + lane;
        floatx4 wr[4];                                      // weights run four chunks ahead
#pragma unroll
        for (int d = 0; d < 4; ++d) wr[d] = W0[(size_t)(d < cin8 ? d : cin8 - 1) * 64];
        for (int c0 = 0; c0 < cin8; c0 += 4) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int c8 = c0 + d;
                const floatx4 wc = wr[d];
                const int nx = c8 + 4 < cin8 ? c8 + 4 : cin8 - 1;
                wr[d] = W0[(size_t)nx * 64];
                const floatx4 xf = *reinterpret_cast<const floatx4*>(sA + c8 * 1024 + lane * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.x, xf.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.y, xf.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.z, xf.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.w, xf.w, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = w * 32 + q * 8 + half * 4;
            floatx4 val;
            val.x = acc[4 * q + 0]; val.y = acc[4 * q + 1];
            val.z = acc[4 * q + 2]; val.w = acc[4 * q + 3];
            val += *reinterpret_cast<const floatx4*>(a.biasB + ch);
            const floatx4 sc = *reinterpret_cast<const floatx4*>(a.sB + ch);
            const floatx4 sh = *reinterpret_cast<const floatx4*>(a.tB + ch);
            floatx4 y = val * sc + sh;
            y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f);
            *reinterpret_cast<floatx4*>(sB + (w * 4 + q) * 1024 + lane * 16) = y;
        }
    }
    __syncthreads();
    // ---- layer C (one output block, of which the first four channels are the quaternion) + normalisation: wave 0
    if (w == 0) {
        floatx16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const int cin8 = a.obB * 4;
        const floatx4* W0 = reinterpret_cast<const floatx4*>(a.WC) + lane;
        floatx4 wr[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) wr[d] = W0[(size_t)(d < cin8 ? d : cin8 - 1) * 64];
        for (int c0 = 0; c0 < cin8; c0 += 4) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int c8 = c0 + d;
                const floatx4 wc = wr[d];
                const int nx = c8 + 4 < cin8 ? c8 + 4 : cin8 - 1;
                wr[d] = W0[(size_t)nx * 64];
                const floatx4 xf = *reinterpret_cast<const floatx4*>(sB + c8 * 1024 + lane * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.x, xf.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.y, xf.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.z, xf.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.w, xf.w, acc, 0, 0, 0);
            }
        }
        // channels 0..3 of match kp: registers 0..3 of the lanes of half 0
        floatx4 val;
        val.x = acc[0]; val.y = acc[1]; val.z = acc[2]; val.w = acc[3];
        val += *reinterpret_cast<const floatx4*>(a.biasC + half * 4);
        const int m = tile * TILE + (lane & 31);
        if (half == 0 && m < a.M) *reinterpret_cast<floatx4*>(a.quat + (size_t)m * 4) = quat_unit(val);
    }
}

// A: 256 -> 512, B: 512 -> 128, C: 128 -> 4 (padded to 32); X = A's input [tile][32][60 slabs][256].  false: shapes this kernel
// does not take (the caller runs the staged launches).
bool mlp_head_supported(const Layer& A, const Layer& B, const Layer& C) {
    return A.ntaps == 1 && B.ntaps == 1 && C.ntaps == 1 && A.cin == 256 && A.cout_pad == 512 && B.cin == 512 && B.cout_pad == 128 &&
           C.cin == 128 && C.cout_pad == 32;
}

int launch_mlp_head(const Layer& A, const Layer& B, const Layer& C, const float* X, int nTiles, int M, float* quat, hipStream_t s,
                    const float* part, const Layer* P, const float* res) {
    MlpArgs a;
    a.X = X;
    a.part = part; a.biasP = P ? P->bias : nullptr; a.resP = res; a.descaleP = P ? P->wph_descale : 1.f; a.nTiles = nTiles;
    if (part && !P) { set_error("mlp head: partial sums without their layer"); return YOHO_EINVAL; }
    a.WA = A.wp; a.biasA = A.bias; a.sA = A.bn_s; a.tA = A.bn_t;
    a.WB = B.wp; a.biasB = B.bias; a.sB = B.bn_s; a.tB = B.bn_t;
    a.WC = C.wp; a.biasC = C.bias;
    a.cinA8 = A.cin / 8; a.obA = A.cout_pad / 32; a.obB = B.cout_pad / 32; a.M = M;
    a.quat = quat;
    const int lds = (a.cinA8 + a.obA * 4 + a.obB * 4) * 1024;
    hipLaunchKernelGGL(mlp_head_kernel, dim3(nTiles), dim3(256), lds, s, a);
    HIPCHK(hipGetLastError());
    return 0;
}

int gconv_init() {
    int rc;
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
    if ((rc = init_t<15, false>())) return rc;
    if ((rc = init_t<12, false>())) return rc;
    if ((rc = init_t<4, false>())) return rc;
    if ((rc = init_t<1, false>())) return rc;
    if ((rc = init_t<1, true>())) return rc;
    return 0;
}

// gpw: 15 / 12 / 4 / 1; gpw == -1 selects the o-split single-group-element variant.
int launch_gconv(const ConvArgs& a, int gpw, int flags, hipStream_t s) {
    if (!a.slabtab || !a.outg) { set_error("launch_gconv: the layer carries no slot tables"); return YOHO_EINVAL; }
    int nstage = G;
    switch (gpw) {
        case 15: return launch_t<15, false>(a, flags, nstage, CFG_FULL, s);
        case 12: return launch_t<12, false>(a, flags, nstage, CFG_C45, s);
        case 4:  return launch_t<4, false>(a, flags, nstage, CFG_C13, s);
        case 1:  return launch_t<1, false>(a, flags, a.ntaps == 1 ? 1 : nstage, CFG_C1, s);
        case -1: return launch_t<1, true>(a, flags, a.ntaps == 1 ? 1 : nstage, CFG_C1, s);
        default: set_error("launch_gconv: unsupported slots-per-wave %d", gpw); return YOHO_EINVAL;
    }
}

}  // namespace yoho
