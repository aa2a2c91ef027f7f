// Brute-force nearest neighbour, mutual matching and the coarse-rotation index.
//
//   nn_kernel        modified_knn_matcher.pdist / find_nn_gpu   utils/knn_search.py:17-20,26-66
//   mutual kernels   matcher_dual.match                         tests/matcher.py:37-48
//   des2r_kernel     extractor_dr_index.Batch_Des2R_torch       tests/extractor.py:74-78
//
// The reference takes argmin of sqrt(sum((a-b)^2) + 1e-7) over an explicit difference (not the
// GEMM expansion); indices must be bit-exact, so the arithmetic below is written with explicit
// round-to-nearest intrinsics (no FMA contraction) in the summation order torch's CPU reduction
// uses for a contiguous 32-float row: 8 lane sums over x[l], x[8+l], x[16+l], x[24+l], lanes then
// added 0..7 in sequence (pinned by tests/golden/pdist.npz).  The first minimum wins.
// This file is compiled with -ffp-contract=off (yoho_amd/build.py).
#include "common.h"
#include "nnmath.h"

namespace yoho {

constexpr int NN_ROWS = 16;     // source rows per workgroup
constexpr int NN_SPLIT = 16;    // target interleave per source row
constexpr int NN_TT = 256;      // target rows per LDS tile

// src (Ns,D) f32, tgt (Nt,D) f32 -> idx (Ns) int64, dist (Ns) f32 (optional)
template <int D, bool SQUARED>
__global__ __launch_bounds__(256) void nn_kernel(const float* __restrict__ src, int Ns, const float* __restrict__ tgt, int Nt,
                                                 int64_t* __restrict__ idx, float* __restrict__ dist) {
    __shared__ float tile[NN_TT * D];
    __shared__ float rd[NN_ROWS * NN_SPLIT];
    __shared__ int ri[NN_ROWS * NN_SPLIT];
    const int r = threadIdx.x % NN_ROWS, sp = threadIdx.x / NN_ROWS;
    const int row = blockIdx.x * NN_ROWS + r;
    float a[D];
    const int rowc = row < Ns ? row : Ns - 1;
#pragma unroll
    for (int k = 0; k < D; ++k) a[k] = src[(size_t)rowc * D + k];
    // best holds the squared distance.  For the 'L2' form the reference compares sqrt(D2 + 1e-7) (first minimum);
    // sqrt is monotone, so a later candidate can only win with a smaller D2, and the correctly rounded sqrt
    // (computed in f64) is evaluated only when the two D2 are close enough to round to the same distance.
    auto dist_of = [](float d2) -> float { return dist_of_f32(d2); };
    float best = __builtin_inff();
    int besti = 0;
    for (int t0 = 0; t0 < Nt; t0 += NN_TT) {
        const int nt = Nt - t0 < NN_TT ? Nt - t0 : NN_TT;
        __syncthreads();
        for (int i = threadIdx.x; i < nt * D; i += 256) tile[i] = tgt[(size_t)t0 * D + i];
        __syncthreads();
        for (int t = sp; t < nt; t += NN_SPLIT) {
            const float d2 = dist2_f32<D>(a, tile + t * D);
            if (d2 < best) {
                if (SQUARED || d2 < best * (1.0f - 1e-6f) || dist_of(d2) < dist_of(best)) { best = d2; besti = t0 + t; }
            }
        }
    }
    if (!SQUARED) best = dist_of(best);
    rd[r * NN_SPLIT + sp] = best;
    ri[r * NN_SPLIT + sp] = besti;
    __syncthreads();
    if (sp == 0 && row < Ns) {
        float bd = rd[r * NN_SPLIT];
        int bi = ri[r * NN_SPLIT];
        for (int k = 1; k < NN_SPLIT; ++k) {
            const float d = rd[r * NN_SPLIT + k];
            const int i = ri[r * NN_SPLIT + k];
            if (d < bd || (d == bd && i < bi)) { bd = d; bi = i; }
        }
        idx[row] = bi;
        if (dist) dist[row] = bd;
    }
}

int launch_nn(const float* src, int Ns, const float* tgt, int Nt, int D, int dist_type, int64_t* idx, float* dist, hipStream_t s) {
    const int grid = (Ns + NN_ROWS - 1) / NN_ROWS;
    const bool sq = dist_type == YOHO_DIST_SQUARE_L2;
    if (dist_type != YOHO_DIST_L2 && !sq) { set_error("yoho_nn_search: unknown dist_type %d", dist_type); return YOHO_EINVAL; }
    if (D == 32 && !sq) hipLaunchKernelGGL((nn_kernel<32, false>), dim3(grid), dim3(256), 0, s, src, Ns, tgt, Nt, idx, dist);
    else if (D == 32) hipLaunchKernelGGL((nn_kernel<32, true>), dim3(grid), dim3(256), 0, s, src, Ns, tgt, Nt, idx, dist);
    else if (D == 3 && !sq) hipLaunchKernelGGL((nn_kernel<3, false>), dim3(grid), dim3(256), 0, s, src, Ns, tgt, Nt, idx, dist);
    else if (D == 3) hipLaunchKernelGGL((nn_kernel<3, true>), dim3(grid), dim3(256), 0, s, src, Ns, tgt, Nt, idx, dist);
    else { set_error("yoho_nn_search: D must be 32 or 3 (got %d)", D); return YOHO_EINVAL; }
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// D = 32, large problems: the same arithmetic, blocked for the LDS return path and for load balance.
//   * a thread owns TWO source rows, so every target row read from LDS (8 x ds_read_b128) feeds two distance
//     evaluations (the single-row kernel is bound by the LDS return bandwidth, not by its packed fp32 math);
//   * the targets are cut into segments over blockIdx.y so that there are a few workgroups per CU whatever Ns is; the
//     per-segment winners meet in a packed (distance bits << 32 | index) key with atomicMin — distances are
//     non-negative, so the integer order is (distance, then lowest index) = the reference's first minimum.
// ---------------------------------------------------------------------------------------------------------------
template <int D, bool SQUARED>
__global__ __launch_bounds__(256) void nn32seg_kernel(const float* __restrict__ src, int Ns, const float* __restrict__ tgt, int Nt,
                                                      unsigned long long* __restrict__ keys, int segLen, const int* __restrict__ run_if = nullptr) {
    if (run_if && !*run_if) return;                  // fallback launch behind the MFMA pre-filter (matchf.hip): only if it declined
    __shared__ __attribute__((aligned(16))) float tile[NN_TT * D];
    __shared__ float rd[32 * NN_SPLIT];
    __shared__ int ri[32 * NN_SPLIT];
    const int r = threadIdx.x % 16, sp = threadIdx.x / 16;
    const int row0 = blockIdx.x * 32 + r, row1 = row0 + 16;
    float a0[D], a1[D];
    {
        const int rc0 = row0 < Ns ? row0 : Ns - 1, rc1 = row1 < Ns ? row1 : Ns - 1;
#pragma unroll
        for (int k = 0; k < D; ++k) { a0[k] = src[(size_t)rc0 * D + k]; a1[k] = src[(size_t)rc1 * D + k]; }
    }
    auto dist_of = [](float d2) -> float { return dist_of_f32(d2); };
    float best0 = __builtin_inff(), best1 = __builtin_inff();
    int bi0 = 0, bi1 = 0;
    const int tlo = blockIdx.y * segLen;
    const int thi = tlo + segLen < Nt ? tlo + segLen : Nt;
    for (int t0 = tlo; t0 < thi; t0 += NN_TT) {
        const int nt = thi - t0 < NN_TT ? thi - t0 : NN_TT;
        __syncthreads();
        if ((nt * D) % 4 == 0) {
            const float4* g4 = reinterpret_cast<const float4*>(tgt + (size_t)t0 * D);      // t0 is a multiple of 16 rows: 16-byte aligned
            float4* t4 = reinterpret_cast<float4*>(tile);
            for (int i = threadIdx.x; i < nt * D / 4; i += 256) t4[i] = g4[i];
        } else {
            for (int i = threadIdx.x; i < nt * D; i += 256) tile[i] = tgt[(size_t)t0 * D + i];
        }
        __syncthreads();
        for (int t = sp; t < nt; t += NN_SPLIT) {
            float b[D];
            if constexpr (D % 4 == 0) {
#pragma unroll
                for (int k = 0; k < D / 4; ++k) {
                    const float4 v = reinterpret_cast<const float4*>(tile + t * D)[k];
                    b[4 * k] = v.x; b[4 * k + 1] = v.y; b[4 * k + 2] = v.z; b[4 * k + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < D; ++k) b[k] = tile[t * D + k];
            }
            const float d0 = dist2_f32<D>(a0, b), d1 = dist2_f32<D>(a1, b);
            if (d0 < best0) {
                if (SQUARED || d0 < best0 * (1.0f - 1e-6f) || dist_of(d0) < dist_of(best0)) { best0 = d0; bi0 = t0 + t; }
            }
            if (d1 < best1) {
                if (SQUARED || d1 < best1 * (1.0f - 1e-6f) || dist_of(d1) < dist_of(best1)) { best1 = d1; bi1 = t0 + t; }
            }
        }
    }
    if (!SQUARED) { best0 = dist_of(best0); best1 = dist_of(best1); }
    rd[r * NN_SPLIT + sp] = best0; ri[r * NN_SPLIT + sp] = bi0;
    rd[(r + 16) * NN_SPLIT + sp] = best1; ri[(r + 16) * NN_SPLIT + sp] = bi1;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int rr = threadIdx.x, row = blockIdx.x * 32 + rr;
        if (row < Ns && tlo < thi) {
            float bd = rd[rr * NN_SPLIT];
            int bi = ri[rr * NN_SPLIT];
            for (int k = 1; k < NN_SPLIT; ++k) {
                const float d = rd[rr * NN_SPLIT + k];
                const int i = ri[rr * NN_SPLIT + k];
                if (d < bd || (d == bd && i < bi)) { bd = d; bi = i; }
            }
            const unsigned long long key = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bi;
            atomicMin(keys + row, key);
        }
    }
}

__global__ __launch_bounds__(256) void nn_unpack_kernel(const unsigned long long* __restrict__ keys, int Ns, int64_t* __restrict__ idx,
                                                        float* __restrict__ dist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Ns) return;
    const unsigned long long k = keys[i];
    idx[i] = (int64_t)(k & 0xFFFFFFFFull);
    if (dist) dist[i] = __uint_as_float((unsigned)(k >> 32));
}

// the same search into keys that are ALREADY initialised, run only if *run_if is non-zero (D = 32, 'L2' distance)
int launch_nn32seg_if(const float* src, int Ns, const float* tgt, int Nt, unsigned long long* keys, int nCU, const int* run_if, hipStream_t s) {
    const int rb = (Ns + 31) / 32;
    int nseg = (4 * nCU + rb - 1) / rb;
    const int maxseg = (Nt + 63) / 64;
    nseg = nseg < 1 ? 1 : (nseg > maxseg ? maxseg : nseg);
    int segLen = (Nt + nseg - 1) / nseg;
    segLen = (segLen + 15) / 16 * 16;
    nseg = (Nt + segLen - 1) / segLen;
    hipLaunchKernelGGL((nn32seg_kernel<32, false>), dim3(rb, nseg), dim3(256), 0, s, src, Ns, tgt, Nt, keys, segLen, run_if);
    HIPCHK(hipGetLastError());
    return 0;
}

// keys must hold Ns words; they are (re)initialised here
int launch_nn32seg(const float* src, int Ns, const float* tgt, int Nt, bool squared, unsigned long long* keys, int nCU, hipStream_t s,
                   int D = 32) {
    const int rb = (Ns + 31) / 32;
    int nseg = (4 * nCU + rb - 1) / rb;
    const int maxseg = (Nt + 63) / 64;
    nseg = nseg < 1 ? 1 : (nseg > maxseg ? maxseg : nseg);
    int segLen = (Nt + nseg - 1) / nseg;
    segLen = (segLen + 15) / 16 * 16;
    nseg = (Nt + segLen - 1) / segLen;
    HIPCHK(hipMemsetAsync(keys, 0xFF, sizeof(unsigned long long) * (size_t)Ns, s));
    if (D == 3 && squared) hipLaunchKernelGGL((nn32seg_kernel<3, true>), dim3(rb, nseg), dim3(256), 0, s, src, Ns, tgt, Nt, keys, segLen, (const int*)nullptr);
    else if (D == 3) hipLaunchKernelGGL((nn32seg_kernel<3, false>), dim3(rb, nseg), dim3(256), 0, s, src, Ns, tgt, Nt, keys, segLen, (const int*)nullptr);
    else if (squared) hipLaunchKernelGGL((nn32seg_kernel<32, true>), dim3(rb, nseg), dim3(256), 0, s, src, Ns, tgt, Nt, keys, segLen, (const int*)nullptr);
    else hipLaunchKernelGGL((nn32seg_kernel<32, false>), dim3(rb, nseg), dim3(256), 0, s, src, Ns, tgt, Nt, keys, segLen, (const int*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

int launch_nn_unpack(const unsigned long long* keys, int Ns, int64_t* idx, float* dist, hipStream_t s) {
    hipLaunchKernelGGL(nn_unpack_kernel, dim3((Ns + 255) / 256), dim3(256), 0, s, keys, Ns, idx, dist);
    HIPCHK(hipGetLastError());
    return 0;
}

// keep i with back[fwd[i]] == i, ascending i (tests/matcher.py:42-47).  Single workgroup scan.
// PACKED: fwd / back hold (distance bits << 32 | index) keys of the segmented search.
template <bool PACKED>
__global__ __launch_bounds__(1024) void mutual_compact_kernel(const int64_t* __restrict__ fwd, const int64_t* __restrict__ back,
                                                              int Na, int64_t* __restrict__ pairs, int* __restrict__ M_out) {
    __shared__ int wsum[16];
    __shared__ int base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < Na; i0 += 1024) {
        const int i = i0 + tid;
        bool keep = false;
        int64_t j = 0;
        if (i < Na) {
            j = PACKED ? (fwd[i] & 0xFFFFFFFFll) : fwd[i];
            const int64_t bk = PACKED ? (back[j] & 0xFFFFFFFFll) : back[j];
            keep = bk == (int64_t)i;
        }
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int k = 0; k < wv; ++k) off += wsum[k];
        if (keep) { pairs[2 * (size_t)(off + before)] = i; pairs[2 * (size_t)(off + before) + 1] = j; }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; base += t; }
        __syncthreads();
    }
    if (tid == 0) *M_out = base;
}

int launch_mutual_compact(const int64_t* fwd, const int64_t* back, int Na, int64_t* pairs, int* M_out, hipStream_t s, bool packed = false) {
    if (packed) hipLaunchKernelGGL(mutual_compact_kernel<true>, dim3(1), dim3(1024), 0, s, fwd, back, Na, pairs, M_out);
    else hipLaunchKernelGGL(mutual_compact_kernel<false>, dim3(1), dim3(1024), 0, s, fwd, back, Na, pairs, M_out);
    HIPCHK(hipGetLastError());
    return 0;
}

// cor[m, a] = sum_g sum_f d1[m, f, P[a, g]] * d2[m, f, g];  idx[m] = argmax_a (first maximum).
// One wave per match, two waves per workgroup, grid-stride over the matches.  The double sum is a gather of the 60 x 60 Gram
// matrix C = D1^T D2 (contraction over the 32 channels): cor[a] = sum_g C[P[a,g]][g].  C is computed on the fp32 MFMA
// (64 x v_mfma_f32_32x32x2_f32, exact fp32 products, fp32 accumulation).  The operands go from global memory straight into
// registers: lane (g' = lane & 31, half = lane >> 5) of step kk holds channel f = 2 kk + half of group elements g' and g' + 32
// (zero beyond 59), i.e. every load is two 128-byte runs per wave and all 64 of them are in flight at once.  C goes to a
// wave-private LDS tile (row stride 65: the gather's rows are a permutation, so stride 64 would put all lanes on one bank) and
// lane a < 60 adds its 60 entries in the order g = 0..59, the rows P[a][g] read as packed bytes (ctx->dPq, in LDS).
// Optional row indices address the descriptors in place.  (Round 3: the first version staged both descriptors in LDS - 33 KB per
// one-wave workgroup, i.e. one wave per SIMD - and read P with 60 dependent global loads per lane: 170-300 us per 3233 matches.)
typedef float floatx16m __attribute__((ext_vector_type(16)));
constexpr int D2R_C = 65;        // row stride of the Gram matrix
constexpr int D2R_W = 2;         // waves (matches in flight) per workgroup

__global__ __launch_bounds__(64 * D2R_W) void des2r_kernel(const float* __restrict__ e1, const int64_t* __restrict__ i1, const float* __restrict__ e2,
                                                          const int64_t* __restrict__ i2, int istride, const unsigned* __restrict__ Pq, int M,
                                                          int64_t* __restrict__ idx, float* __restrict__ cor) {
    __shared__ float Cls[D2R_W][G * D2R_C];
    __shared__ unsigned Pl[15 * 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 15 * 64; i += 64 * D2R_W) Pl[i] = Pq[i];
    __syncthreads();
    float* Cl = Cls[w];
    const int li = lane & 31, lk = lane >> 5;
    const bool hi_ok = li + 32 < G;
    for (int m = blockIdx.x * D2R_W + w; m < M; m += gridDim.x * D2R_W) {
        const size_t r1 = i1 ? (size_t)i1[(size_t)m * istride] : (size_t)m, r2 = i2 ? (size_t)i2[(size_t)m * istride] : (size_t)m;
        const float* p1 = e1 + r1 * F * G + lk * G + li;
        const float* p2 = e2 + r2 * F * G + lk * G + li;
        float a0[16], a1[16], b0[16], b1[16];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            a0[kk] = p1[2 * kk * G];
            b0[kk] = p2[2 * kk * G];
            a1[kk] = hi_ok ? p1[2 * kk * G + 32] : 0.f;
            b1[kk] = hi_ok ? p2[2 * kk * G + 32] : 0.f;
        }
        floatx16m acc[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b0[kk], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk], b1[kk], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b0[kk], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk], b1[kk], acc[1][1], 0, 0, 0);
        }
        // D[row g'][col g]: lane (col = lane & 31, half = lane >> 5), reg r -> row = (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * lk;
                    if (row < G && (cb == 0 || hi_ok)) Cl[row * D2R_C + 32 * cb + li] = acc[rb][cb][r];
                }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 15; ++q) {
            const unsigned pw = Pl[q * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += Cl[((pw >> (8 * j)) & 255u) * D2R_C + 4 * q + j];
        }
        if (cor && lane < G) cor[(size_t)m * G + lane] = sum;
        float bv = lane < G ? sum : -__builtin_inff();
        int bi = lane;
        for (int o = 32; o >= 1; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) idx[m] = bi;
        __builtin_amdgcn_wave_barrier();             // the next match's tile overwrites Cl
    }
}

int launch_des2r(const float* e1, const int64_t* i1, const float* e2, const int64_t* i2, int istride, const unsigned* Pq, int M, int64_t* idx,
                 float* cor, hipStream_t s) {
    if (M <= 0) return 0;
    static const int ncu = [] { int d = 0, n = 256; (void)hipGetDevice(&d); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d); return n; }();
    const int nwg = std::min((M + D2R_W - 1) / D2R_W, 4 * ncu);          // 35 KB of LDS per workgroup: four per CU
    hipLaunchKernelGGL(des2r_kernel, dim3(nwg), dim3(64 * D2R_W), 0, s, e1, i1, e2, i2, istride, Pq, M, idx, cor);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho

using namespace yoho;

extern "C" {

int yoho_nn_search(yoho_ctx* c, const float* src, int Ns, const float* tgt, int Nt, int D, int dist_type, int64_t* idx, float* dist,
                   void* stream) {
    if (!c || !src || !tgt || !idx || Ns < 0 || Nt < 1) { set_error("yoho_nn_search: bad argument"); return YOHO_EINVAL; }
    if (Ns == 0) return 0;
    YOHO_NEED_ALIGNED("yoho_nn_search", (D == 32 ? 15 : 3), src, tgt);
    YOHO_NEED_ALIGNED("yoho_nn_search", 7, idx);
    YOHO_NEED_ALIGNED("yoho_nn_search", 3, dist);
    HIPCHK(hipSetDevice(c->device));
    if (D == 3 && c->nn_cell > 0.0 && (size_t)Ns * Nt >= (1u << 20) && (dist_type == YOHO_DIST_L2 || dist_type == YOHO_DIST_SQUARE_L2)) {
        int rc;
        if ((rc = ensure_ws(c, grid_nn_ws_bytes(Ns, Nt), (hipStream_t)stream))) return rc;
        return launch_grid_nn(dist_type == YOHO_DIST_SQUARE_L2 ? 0 : 1, src, Ns, nullptr, tgt, Nt, c->nn_cell, c->ws.p, idx, dist, nullptr, nullptr,
                              c->nCU, (hipStream_t)stream);
    }
    if ((D == 32 || D == 3) && (size_t)Ns * Nt >= (1u << 20) && (dist_type == YOHO_DIST_L2 || dist_type == YOHO_DIST_SQUARE_L2)) {
        int rc;
        if ((rc = ensure_ws(c, sizeof(unsigned long long) * (size_t)Ns, (hipStream_t)stream))) return rc;
        unsigned long long* keys = (unsigned long long*)c->ws.p;
        if ((rc = launch_nn32seg(src, Ns, tgt, Nt, dist_type == YOHO_DIST_SQUARE_L2, keys, c->nCU, (hipStream_t)stream, D))) return rc;
        return launch_nn_unpack(keys, Ns, idx, dist, (hipStream_t)stream);
    }
    return launch_nn(src, Ns, tgt, Nt, D, dist_type, idx, dist, (hipStream_t)stream);
}

int yoho_mutual_nn(yoho_ctx* c, const float* a, int Na, const float* b, int Nb, int64_t* pairs, int* M_out, void* stream) {
    if (!c || !a || !b || !pairs || !M_out || Na < 1 || Nb < 1) { set_error("yoho_mutual_nn: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_mutual_nn", 15, a, b, pairs);
    YOHO_NEED_ALIGNED("yoho_mutual_nn", 3, M_out);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    // workspace: fwd (Na) + back (Nb) int64, placed after everything the descriptor passes use
    const size_t need = sizeof(int64_t) * ((size_t)Na + Nb);
    int rc;
    if ((rc = ensure_ws(c, need, s))) return rc;
    int64_t* fwd = (int64_t*)c->ws.p;
    int64_t* back = fwd + Na;
    if ((size_t)Na * Nb >= (1u << 20) && c->nn_prefilter) {
        // MFMA pre-filter + exact evaluation of the candidates (matchf.hip): the same packed keys as the segmented search below
        if ((rc = ensure_ws(c, mutual_prefilter_ws_bytes(Na, Nb), s))) return rc;
        unsigned long long *kA = nullptr, *kB = nullptr;
        if ((rc = launch_mutual_prefilter(a, Na, b, Nb, c->ws.p, &kA, &kB, c->nCU, s, c->env.nn_splits))) return rc;
        return launch_mutual_compact((const int64_t*)kA, (const int64_t*)kB, Na, pairs, M_out, s, true);
    }
    if ((size_t)Na * Nb >= (1u << 20)) {
        // segmented search: the workspace words are the packed keys, the compaction reads the index half
        if ((rc = launch_nn32seg(a, Na, b, Nb, false, (unsigned long long*)fwd, c->nCU, s))) return rc;
        if ((rc = launch_nn32seg(b, Nb, a, Na, false, (unsigned long long*)back, c->nCU, s))) return rc;
        return launch_mutual_compact(fwd, back, Na, pairs, M_out, s, true);
    }
    if ((rc = launch_nn(a, Na, b, Nb, 32, YOHO_DIST_L2, fwd, nullptr, s))) return rc;     // NN of every a-row in b  (KNN(feats1, feats0))
    if ((rc = launch_nn(b, Nb, a, Na, 32, YOHO_DIST_L2, back, nullptr, s))) return rc;    // NN of every b-row in a  (KNN(feats0, feats1))
    return launch_mutual_compact(fwd, back, Na, pairs, M_out, s);
}

int yoho_des2r(yoho_ctx* c, const float* d1, const float* d2, int M, int64_t* idx, float* cor, void* stream) {
    if (!c || !d1 || !d2 || !idx || M < 0) { set_error("yoho_des2r: bad argument"); return YOHO_EINVAL; }
    if (M == 0) return 0;
    YOHO_NEED_ALIGNED("yoho_des2r", 15, d1, d2);
    YOHO_NEED_ALIGNED("yoho_des2r", 7, idx);
    YOHO_NEED_ALIGNED("yoho_des2r", 3, cor);
    HIPCHK(hipSetDevice(c->device));
    return launch_des2r(d1, nullptr, d2, nullptr, 1, c->dPq, M, idx, cor, (hipStream_t)stream);
}

int yoho_des2r_indexed(yoho_ctx* c, const float* e1, const int64_t* i1, const float* e2, const int64_t* i2, int istride, int M,
                       int64_t* idx, float* cor, void* stream) {
    if (!c || !e1 || !e2 || !i1 || !i2 || !idx || M < 0 || istride < 1) { set_error("yoho_des2r_indexed: bad argument"); return YOHO_EINVAL; }
    if (M == 0) return 0;
    YOHO_NEED_ALIGNED("yoho_des2r_indexed", 15, e1, e2);
    YOHO_NEED_ALIGNED("yoho_des2r_indexed", 7, i1, i2, idx);
    YOHO_NEED_ALIGNED("yoho_des2r_indexed", 3, cor);
    HIPCHK(hipSetDevice(c->device));
    return launch_des2r(e1, i1, e2, i2, istride, c->dPq, M, idx, cor, (hipStream_t)stream);
}

}  // extern "C"
