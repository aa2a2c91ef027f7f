// Mutual nearest neighbours of two descriptor sets with an MFMA pre-filter - same answers as the brute-force kernels of match.hip.
//
// The reference takes, for every row of a, argmin_j sqrt(sum_f (a_f - b_jf)^2 + 1e-7) over an explicit difference and keeps the first
// minimum (utils/knn_search.py:17-20,36-44; tests/matcher.py:37-48), both ways.  Indices must be bit-exact, so the distance that
// DECIDES is always dist_of_f32(dist2_f32<32>(.,.)) of nnmath.h, the brute-force kernels' own arithmetic.  But 2 x 5000 x 5000 of
// those cost 0.2 ms of vector fp32 per pair, and almost all of them are nowhere near the minimum.  Here the Gram matrix G = a b^T is
// computed on the fp16 MFMA (one rounding of the inputs, fp32 accumulation: a few microseconds), twice:
//   pass 1   row minima of s_ij = |b_j|^2 - 2 G_ij and column minima of t_ij = |a_i|^2 - 2 G_ij (the squared distance up to the
//            term that is constant along the search): a workgroup serves ONE direction, owns 128 rows of its set and streams the
//            other set past them, so a minimum is a register until the end (one ordered-integer atomicMin per row and workgroup);
//   pass 2   every (i, j) whose score is within a PROVEN error band of its row (column) minimum is a candidate: its exact distance
//            is evaluated with dist2_f32 / dist_of_f32 and merged as the packed key (distance bits << 32 | index) the brute-force
//            path uses, so ties go to the lowest index as there.
// Band.  With u16 = 2^-11 (fp16 rounding), |G~ - G| <= (2 u16 + u16^2 + 32 * 2^-24) |a_i||b_j| <= 1.0e-3 |a_i||b_j|, the fp32 norms
// are off by <= 33 * 2^-24 |b_j|^2, the explicit-difference sums by <= 40 * 2^-24 (|a_i| + |b_j|)^2, fp16 subnormals add at most
// 2^-25 * sqrt(32) (|a_i| + |b_j|) to a dot product.  If j* is the exact winner and j^ the approximate one, s~(j*) <= s~(j^) + the
// sum of these errors at both indices, so with Bmax = max_j |b_j|
//   band_i = 4.5e-3 |a_i| Bmax + 2e-5 (|a_i| + Bmax)^2 + 4e-6 (|a_i| + Bmax)
// contains j* (all distances that round to the winner's distance included: they differ from it by ~1e-7 relative).  For unit-scale
// descriptors the band is 5e-3 of squared distance: a handful of candidates per row.  Non-finite inputs or magnitudes beyond the
// fp16 range raise a flag and the brute-force kernels run instead (they are launched behind the flag either way).
// Compiled with -ffp-contract=off like match.hip (the exact distances must not fuse).
#include <cstdlib>
#include "common.h"
#include "nnmath.h"

namespace yoho {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

struct MfArgs {
    const float* a; const float* b;
    int Na, Nb;
    float* na2; float* nb2;              // squared norms
    _Float16* a16; _Float16* b16;        // the rows rounded to fp16 (RNE), 32 halfs each: the MFMA operands of both Gram passes
    unsigned* rowmin; unsigned* colmin;  // ordered-integer images of the minima of s (per a row) and t (per b row)
    unsigned long long* keysA; unsigned long long* keysB;   // packed (distance bits << 32 | index) winners
    unsigned* maxn2;                     // [0] max |a_i|^2 bits, [1] max |b_j|^2 bits
    int* bad;                            // non-finite / out-of-range input: brute force instead
    int tilesPer;                        // 32-row tiles of the OTHER set per workgroup of the Gram passes
    unsigned long long* cand;            // pass 2: candidate pairs (i << 34 | j << 4 | for_row << 1 | for_col), evaluated exactly by mf_exact_kernel
    unsigned* ncand; unsigned cap;       // their number (may exceed cap: then `bad` is raised and brute force answers)
};

__device__ __forceinline__ unsigned ord_of(float f) {          // monotone map float -> unsigned
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_to(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

// squared norms, their maxima, the range check, the fp16 image of every row, and the initial values of minima and keys
__global__ __launch_bounds__(256) void mf_norms_kernel(MfArgs p) {
    __shared__ unsigned red[2][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = p.Na + p.Nb;
    const bool live = i < n, isa = i < p.Na;
    const int r = live ? (isa ? i : i - p.Na) : 0;
    float s = 0.f, m = 0.f;
    if (live) {
        const float4* x = reinterpret_cast<const float4*>((isa ? p.a : p.b) + (size_t)r * 32);
        halfx8* h16 = reinterpret_cast<halfx8*>((isa ? p.a16 : p.b16) + (size_t)r * 32);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 v = x[2 * k], u = x[2 * k + 1];
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            s += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
            m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
            m = fmaxf(fmaxf(fmaxf(m, fabsf(u.x)), fmaxf(fabsf(u.y), fabsf(u.z))), fabsf(u.w));
            halfx8 h;
            h[0] = (_Float16)v.x; h[1] = (_Float16)v.y; h[2] = (_Float16)v.z; h[3] = (_Float16)v.w;
            h[4] = (_Float16)u.x; h[5] = (_Float16)u.y; h[6] = (_Float16)u.z; h[7] = (_Float16)u.w;
            h16[k] = h;
        }
        (isa ? p.na2 : p.nb2)[r] = s;
        (isa ? p.rowmin : p.colmin)[r] = 0xFFFFFFFFu;
        (isa ? p.keysA : p.keysB)[r] = ~0ull;
    }
    const bool bad = live && (!(s <= 1e30f) || !(m <= 6.0e4f));          // NaN / inf / beyond the fp16 range
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(p.bad, 1);
    // maxima of the squared norms (s >= 0: integer order = float order): one atomic per workgroup and set
    unsigned ma = (live && isa && !bad) ? __float_as_uint(s) : 0u, mb = (live && !isa && !bad) ? __float_as_uint(s) : 0u;
    for (int o = 32; o >= 1; o >>= 1) { ma = max(ma, (unsigned)__shfl_xor((int)ma, o)); mb = max(mb, (unsigned)__shfl_xor((int)mb, o)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ma; red[1][threadIdx.x >> 6] = mb; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned v = max(max(red[threadIdx.x][0], red[threadIdx.x][1]), max(red[threadIdx.x][2], red[threadIdx.x][3]));
        if (v) atomicMax(p.maxn2 + threadIdx.x, v);
    }
}

// the candidate band of a row with squared norm n2 against a set whose largest squared norm is m2 (header comment)
__device__ __forceinline__ float band_of(float n2, float m2) {
    const float mine = sqrtf(n2) * 1.000001f, other = sqrtf(m2) * 1.000001f;
    const float sum = mine + other;
    return 4.5e-3f * mine * other + 2e-5f * sum * sum + 4e-6f * sum + 1e-30f;
}

// exact distances of the candidates of pass 2, one thread per pair, merged as packed keys (distance bits << 32 | index) like the
// brute-force kernels do.  In pass 2 a candidate used to be evaluated on the spot: one ~150-instruction divergent call per
// (column slab, register) slot with any candidate in the wave - that was most of the pass.
__global__ __launch_bounds__(256) void mf_exact_kernel(MfArgs p) {
    if (*p.bad) return;
    const unsigned n = *p.ncand < p.cap ? *p.ncand : p.cap;
    for (unsigned c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
        const unsigned long long e = p.cand[c];
        const int i = (int)(e >> 34), j = (int)((e >> 4) & 0x3FFFFFFFull);
        float av[32], bv[32];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 x = reinterpret_cast<const float4*>(p.a + (size_t)i * 32)[k], y = reinterpret_cast<const float4*>(p.b + (size_t)j * 32)[k];
            av[4 * k] = x.x; av[4 * k + 1] = x.y; av[4 * k + 2] = x.z; av[4 * k + 3] = x.w;
            bv[4 * k] = y.x; bv[4 * k + 1] = y.y; bv[4 * k + 2] = y.z; bv[4 * k + 3] = y.w;
        }
        // (a - b)^2 and (b - a)^2 are the same floats, so one evaluation serves both search directions
        const unsigned long long d = (unsigned long long)__float_as_uint(dist_of_f32(dist2_f32<32>(av, bv))) << 32;
        if (e & 2ull) atomicMin(p.keysA + i, d | (unsigned)j);
        if (e & 1ull) atomicMin(p.keysB + j, d | (unsigned)i);
    }
}

// The Gram passes, one search direction per workgroup (blockIdx.z = 0: every a row against b, minima of s = |b_j|^2 - 2 G; 1: every
// b row against a, minima of t = |a_i|^2 - 2 G).  A workgroup OWNS 128 rows of its set - 32 per wave, their fragments in registers
// for the whole kernel - and streams tilesPer 32-row tiles of the other set past them (fp16 rows written by mf_norms_kernel, two
// 16-byte loads per lane and tile, three tiles in flight).  MFMA 32x32x16 f16: A = own rows (lane & 31, k group lane >> 5), B =
// other rows; D[own][other]: lane -> other = lane & 31, half = lane >> 5, register e -> own row (e & 3) + 8 (e >> 2) + 4 half.
// The running minimum of an own row stays in a register across all tiles: one shuffle reduction and ONE atomicMin per row and
// workgroup at the end (blockIdx.y splits the other set, so a row meets gridDim.y of them).  Until round 3 one Gram tile served both
// directions: 128 columns in registers x 384 streamed rows per workgroup, which cost Na Nb / 128 + Na Nb / 384 = 260 k atomics per
// pass on 40 KB of minima - the passes were bound by those, not by their arithmetic (computing G twice is 3 GFLOP).
template <int PASS, int D>
__global__ __launch_bounds__(256) void mf_gram_kernel(MfArgs p) {
    // one reading of the flag per workgroup (another workgroup may raise it at any moment, and pass 2 ends in workgroup barriers)
    __shared__ int wg_bad;
    if (threadIdx.x == 0) wg_bad = *p.bad;
    __syncthreads();
    if (wg_bad) return;
    constexpr int WL = PASS == 2 ? 1024 : 1;                 // pass 2: per-wave list of the candidates found since the last flush
    __shared__ unsigned long long wl[4][WL];
    __shared__ unsigned wtot[4], wbase;
    int wcnt = 0;                                            // wave-uniform
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // A wave's list goes to the global list when it is full (rare: a handful of candidates per row) and, for all four waves
    // together, once at the end of the workgroup: ONE atomicAdd on the shared counter per workgroup.  (A flush per wave and row
    // step was 6000 returning atomics on one address per launch - they serialise in the L2 and were most of the pass.)
    auto copy_out = [&](unsigned base) {
        if (base + (unsigned)wcnt > p.cap) { if (lane == 0) *p.bad = 1; }       // list full: brute force answers instead
        else for (int k = lane; k < wcnt; k += 64) p.cand[base + k] = wl[w][k];
        wcnt = 0;
    };
    auto flush = [&]() {
        if (wcnt == 0) return;
        // the list was written by other lanes of this wave: all their LDS stores are issued and complete before the cross-lane
        // reads below (a wave's LDS operations retire in order; the fence keeps the compiler from moving the loads above the stores)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(p.ncand, (unsigned)wcnt);
        copy_out(__builtin_amdgcn_readfirstlane(base));
    };
    const int l31 = lane & 31, half = lane >> 5;
    const int dir = blockIdx.z;
    const int No = dir ? p.Nb : p.Na, Nt = dir ? p.Na : p.Nb;              // own rows, rows of the other set
    const _Float16* own16 = dir ? p.b16 : p.a16;
    const _Float16* oth16 = dir ? p.a16 : p.b16;
    const float* n2oth = dir ? p.na2 : p.nb2;                               // the score carries the OTHER row's squared norm
    const float* n2own = dir ? p.nb2 : p.na2;
    unsigned* omin = dir ? p.colmin : p.rowmin;
    const int r0 = blockIdx.x * 128 + w * 32;
    const int t0 = blockIdx.y * p.tilesPer;
    const int ntile_all = (Nt + 31) >> 5;
    const int t1 = t0 + p.tilesPer < ntile_all ? t0 + p.tilesPer : ntile_all;
    if (r0 < No && t0 < t1) {
        const int io = r0 + l31 < No ? r0 + l31 : No - 1;
        const halfx8 A0 = *reinterpret_cast<const halfx8*>(own16 + (size_t)io * 32 + 8 * half);
        const halfx8 A1 = *reinterpret_cast<const halfx8*>(own16 + (size_t)io * 32 + 16 + 8 * half);
        // per-register own-row data: rows r0 + (e & 3) + 8 (e >> 2) + 4 half
        float rth[16];
        bool rok[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * half;
            rok[e] = row < No;
            if (PASS == 1) rth[e] = __builtin_inff();                       // running minimum
            else {
                const int rr = rok[e] ? row : No - 1;
                rth[e] = ord_to(omin[rr]) + band_of(n2own[rr], __uint_as_float(p.maxn2[dir ? 0 : 1]));
            }
        }
        halfx8 B0[D], B1[D];
        float nj[D];
        auto load = [&](int t, int slot) {
            const int j = (t << 5) + l31;
            const int jc = j < Nt ? j : Nt - 1;
            B0[slot] = *reinterpret_cast<const halfx8*>(oth16 + (size_t)jc * 32 + 8 * half);
            B1[slot] = *reinterpret_cast<const halfx8*>(oth16 + (size_t)jc * 32 + 16 + 8 * half);
            nj[slot] = n2oth[jc];
        };
#pragma unroll
        for (int d = 0; d < D; ++d) load(t0 + d < t1 ? t0 + d : t1 - 1, d);
        for (int tb = t0; tb < t1; tb += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int t = tb + d;
                if (t < t1) {                                               // uniform
                    floatx16 g;
#pragma unroll
                    for (int e = 0; e < 16; ++e) g[e] = 0.f;
                    g = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, B0[d], g, 0, 0, 0);
                    g = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, B1[d], g, 0, 0, 0);
                    const float njd = nj[d];
                    const int j = (t << 5) + l31;
                    const bool jok = j < Nt;
                    load(t + D < t1 ? t + D : t1 - 1, d);                   // this slot's next tile (its operands are consumed above)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float sc = njd - 2.f * g[e];
                        if (PASS == 1) {
                            if (jok) rth[e] = fminf(rth[e], sc);
                        } else {
                            const bool c = jok && rok[e] && sc <= rth[e];
                            const unsigned long long m = __ballot(c);
                            if (m) {                                        // uniform
                                const int add = __popcll(m);
                                if (wcnt + add > WL) flush();
                                if (c) {
                                    const unsigned long long own = (unsigned long long)(r0 + (e & 3) + 8 * (e >> 2) + 4 * half), oth = (unsigned long long)j;
                                    wl[w][wcnt + __popcll(m & ((1ull << lane) - 1ull))] =
                                        dir ? ((oth << 34) | (own << 4) | 1ull) : ((own << 34) | (oth << 4) | 2ull);
                                }
                                wcnt += add;
                            }
                        }
                    }
                }
            }
        }
        if (PASS == 1) {
            // minimum of every own row over this workgroup's part of the other set: across the 32 lanes of a half, one atomic per row
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = rth[e];
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
                const int row = r0 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (l31 == 0 && rok[e] && v < __builtin_inff()) atomicMin(omin + row, ord_of(v));
            }
        }
    }
    if (PASS == 2) {
        if (lane == 0) wtot[w] = (unsigned)wcnt;
        __syncthreads();                                     // also orders every wave's list stores before the copies below
        if (threadIdx.x == 0) {
            const unsigned tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
            wbase = tot ? atomicAdd(p.ncand, tot) : 0u;
        }
        __syncthreads();
        unsigned base = wbase;
        for (int k = 0; k < w; ++k) base += wtot[k];
        if (wcnt) copy_out(base);
    }
}

int launch_nn32seg_if(const float* src, int Ns, const float* tgt, int Nt, unsigned long long* keys, int nCU, const int* run_if, hipStream_t s);

size_t mutual_prefilter_ws_bytes(int Na, int Nb) {
    const size_t n = (size_t)Na + Nb;
    return n * (sizeof(unsigned long long) + sizeof(float) + sizeof(unsigned) + 64) + 512 + (32 * n + 4096) * sizeof(unsigned long long);
}

// keysA (Na) / keysB (Nb) receive the packed winners (the layout launch_mutual_compact<PACKED> reads); ws as sized above
int launch_mutual_prefilter(const float* a, int Na, const float* b, int Nb, void* ws, unsigned long long** keysA, unsigned long long** keysB,
                            int nCU, hipStream_t s, int nn_splits) {
    MfArgs p;
    char* w = (char*)ws;
    p.a = a; p.b = b; p.Na = Na; p.Nb = Nb;
    p.keysA = (unsigned long long*)w; w += sizeof(unsigned long long) * (size_t)Na;
    p.keysB = (unsigned long long*)w; w += sizeof(unsigned long long) * (size_t)Nb;
    p.na2 = (float*)w; w += sizeof(float) * (size_t)Na;
    p.nb2 = (float*)w; w += sizeof(float) * (size_t)Nb;
    p.rowmin = (unsigned*)w; w += sizeof(unsigned) * (size_t)Na;
    p.colmin = (unsigned*)w; w += sizeof(unsigned) * (size_t)Nb;
    p.maxn2 = (unsigned*)w; w += 2 * sizeof(unsigned);
    p.bad = (int*)w; w += sizeof(int);
    p.ncand = (unsigned*)w; w += sizeof(unsigned);
    w = (char*)(((size_t)w + 63) & ~(size_t)63);
    p.a16 = (_Float16*)w; w += 64 * (size_t)Na;
    p.b16 = (_Float16*)w; w += 64 * (size_t)Nb;
    p.cand = (unsigned long long*)(((size_t)w + 15) & ~(size_t)15);
    p.cap = 32u * (unsigned)(Na + Nb) + 4000u;                 // a pair in the band of both its row and its column is listed once per direction
    *keysA = p.keysA; *keysB = p.keysB;
    HIPCHK(hipMemsetAsync(p.maxn2, 0, 16, s));                  // maxima, `bad`, candidate count
    const int n = Na + Nb;
    hipLaunchKernelGGL(mf_norms_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p);
    // grid: (128-row blocks of the larger set, splits of the other set, 2 directions), about two workgroups per CU
    const int nmax = Na > Nb ? Na : Nb, nmin = Na < Nb ? Na : Nb;
    const int rb = (nmax + 127) / 128, tiles = (nmax + 31) / 32;
    int splits = nn_splits > 0 ? nn_splits : (2 * nCU + 2 * rb - 1) / (2 * rb);
    splits = splits < 1 ? 1 : (splits > tiles ? tiles : splits);
    p.tilesPer = (tiles + splits - 1) / splits;
    splits = (tiles + p.tilesPer - 1) / p.tilesPer;
    (void)nmin;
    hipLaunchKernelGGL((mf_gram_kernel<1, 3>), dim3(rb, splits, 2), dim3(256), 0, s, p);
    hipLaunchKernelGGL((mf_gram_kernel<2, 3>), dim3(rb, splits, 2), dim3(256), 0, s, p);
    hipLaunchKernelGGL(mf_exact_kernel, dim3(nCU > 0 ? 2 * nCU : 512), dim3(256), 0, s, p);
    HIPCHK(hipGetLastError());
    // inputs the pre-filter cannot take: the brute-force kernels run (they return at once otherwise)
    int rc;
    if ((rc = launch_nn32seg_if(a, Na, b, Nb, p.keysA, nCU, p.bad, s))) return rc;
    return launch_nn32seg_if(b, Nb, a, Na, p.keysB, nCU, p.bad, s);
}

}  // namespace yoho
