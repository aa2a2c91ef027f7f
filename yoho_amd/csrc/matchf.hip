// Mutual nearest neighbours of two descriptor sets with an MFMA pre-filter - same answers as the brute-force kernels of match.hip.
//
// The reference takes, for every row of a, argmin_j sqrt(sum_f (a_f - b_jf)^2 + 1e-7) over an explicit difference and keeps the first
// minimum (utils/knn_search.py:17-20,36-44; tests/matcher.py:37-48), both ways.  Indices must be bit-exact, so the distance that
// DECIDES is always dist_of_f32(dist2_f32<32>(.,.)) of nnmath.h, the brute-force kernels' own arithmetic.  But 2 x 5000 x 5000 of
// those cost 0.2 ms of vector fp32 per pair, and almost all of them are nowhere near the minimum.  Here the Gram matrix G = a b^T is
// computed on the fp16 MFMA (one rounding of the inputs, fp32 accumulation: a few microseconds), twice:
//   pass 1   row minima of s_ij = |b_j|^2 - 2 G_ij and column minima of t_ij = |a_i|^2 - 2 G_ij (the squared distance up to the
//            term that is constant along the search), by ordered-integer atomicMin;
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
#include "common.h"
#include "nnmath.h"

namespace yoho {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

struct MfArgs {
    const float* a; const float* b;
    int Na, Nb;
    float* na2; float* nb2;              // squared norms
    float* bandA; float* bandB;          // candidate bands
    unsigned* rowmin; unsigned* colmin;  // ordered-integer images of the minima of s (per a row) and t (per b row)
    unsigned long long* keysA; unsigned long long* keysB;   // packed (distance bits << 32 | index) winners
    unsigned* maxn2;                     // [0] max |a_i|^2 bits, [1] max |b_j|^2 bits
    int* bad;                            // non-finite / out-of-range input: brute force instead
    int segRows;                         // a rows per workgroup of the Gram passes
    unsigned long long* cand;            // pass 2: candidate pairs (i << 34 | j << 4 | for_row << 1 | for_col), evaluated exactly by mf_exact_kernel
    unsigned* ncand; unsigned cap;       // their number (may exceed cap: then `bad` is raised and brute force answers)
};

__device__ __forceinline__ unsigned ord_of(float f) {          // monotone map float -> unsigned
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_to(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o); }

// squared norms, their maxima, the range check, and the initial values of minima and keys
__global__ __launch_bounds__(256) void mf_norms_kernel(MfArgs p) {
    __shared__ unsigned red[2][4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = p.Na + p.Nb;
    const bool live = i < n, isa = i < p.Na;
    const int r = live ? (isa ? i : i - p.Na) : 0;
    float s = 0.f, m = 0.f;
    if (live) {
        const float4* x = reinterpret_cast<const float4*>((isa ? p.a : p.b) + (size_t)r * 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 v = x[k];
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            m = fmaxf(fmaxf(fmaxf(m, fabsf(v.x)), fmaxf(fabsf(v.y), fabsf(v.z))), fabsf(v.w));
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

__global__ __launch_bounds__(256) void mf_band_kernel(MfArgs p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = p.Na + p.Nb;
    if (i >= n) return;
    const bool isa = i < p.Na;
    const int r = isa ? i : i - p.Na;
    const float mine = sqrtf((isa ? p.na2 : p.nb2)[r]) * 1.000001f;
    const float other = sqrtf(__uint_as_float(p.maxn2[isa ? 1 : 0])) * 1.000001f;
    const float sum = mine + other;
    (isa ? p.bandA : p.bandB)[r] = 4.5e-3f * mine * other + 2e-5f * sum * sum + 4e-6f * sum + 1e-30f;
}

__device__ __forceinline__ halfx8 frag_of(const float* row) {      // 8 consecutive floats -> 8 halfs (round to nearest even)
    const float4 lo = *reinterpret_cast<const float4*>(row), hi = *reinterpret_cast<const float4*>(row + 4);
    halfx8 h;
    h[0] = (_Float16)lo.x; h[1] = (_Float16)lo.y; h[2] = (_Float16)lo.z; h[3] = (_Float16)lo.w;
    h[4] = (_Float16)hi.x; h[5] = (_Float16)hi.y; h[6] = (_Float16)hi.z; h[7] = (_Float16)hi.w;
    return h;
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

// One workgroup: 128 b rows (columns of G, fragments kept in registers) x segRows a rows, 32 a rows per wave and step.
// MFMA 32x32x16 f16: A = a rows (lane & 31, k group lane >> 5), B = b rows; D[row][col]: lane -> col = lane & 31, half = lane >> 5,
// register e -> row (e & 3) + 8 (e >> 2) + 4 half.
template <int PASS>
__global__ __launch_bounds__(256) void mf_gram_kernel(MfArgs p) {
    if (*p.bad) return;
    constexpr int WL = PASS == 2 ? 1024 : 1;                 // pass 2: per-wave list of the candidates found since the last flush
    __shared__ unsigned long long wl[4][WL];
    int wcnt = 0;                                            // wave-uniform
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto flush = [&]() {
        if (wcnt == 0) return;
        // the list was written by other lanes of this wave: all their LDS stores are issued and complete before the cross-lane
        // reads below (a wave's LDS operations retire in order; the fence keeps the compiler from moving the loads above the stores)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(p.ncand, (unsigned)wcnt);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base + (unsigned)wcnt > p.cap) { if (lane == 0) *p.bad = 1; }       // list full: brute force answers instead
        else for (int k = lane; k < wcnt; k += 64) p.cand[base + k] = wl[w][k];
        wcnt = 0;
    };
    const int l31 = lane & 31, half = lane >> 5;
    const int col0 = blockIdx.x * 128;
    halfx8 Bf[4][2];
    float nbj[4], cbest[4], cband[4];
    int jcol[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = col0 + t * 32 + l31;
        jcol[t] = j;
        const int jc = j < p.Nb ? j : p.Nb - 1;
        const float* row = p.b + (size_t)jc * 32 + 8 * half;
        Bf[t][0] = frag_of(row);
        Bf[t][1] = frag_of(row + 16);
        nbj[t] = p.nb2[jc];
        if (PASS == 1) cbest[t] = __builtin_inff();
        else { cbest[t] = ord_to(p.colmin[jc]); cband[t] = p.bandB[jc]; }
    }
    const int rlo = blockIdx.y * p.segRows;
    const int rhi = rlo + p.segRows < p.Na ? rlo + p.segRows : p.Na;
    for (int r0 = rlo + w * 32; r0 < rhi; r0 += 128) {
        const int ia = r0 + l31 < p.Na ? r0 + l31 : p.Na - 1;
        const float* arow = p.a + (size_t)ia * 32 + 8 * half;
        const halfx8 A0 = frag_of(arow), A1 = frag_of(arow + 16);
        // per-register row data: rows r0 + (e & 3) + 8 (e >> 2) + 4 half
        float na[16], rth[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rb = r0 + 8 * q + 4 * half;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rr = rb + e < p.Na ? rb + e : p.Na - 1;
                na[4 * q + e] = p.na2[rr];
                if (PASS == 2) rth[4 * q + e] = ord_to(p.rowmin[rr]) + p.bandA[rr];
                else rth[4 * q + e] = __builtin_inff();             // running row minimum
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            floatx16 g;
#pragma unroll
            for (int e = 0; e < 16; ++e) g[e] = 0.f;
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(A0, Bf[t][0], g, 0, 0, 0);
            g = __builtin_amdgcn_mfma_f32_32x32x16_f16(A1, Bf[t][1], g, 0, 0, 0);
            const bool jok = jcol[t] < p.Nb;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = r0 + (e & 3) + 8 * (e >> 2) + 4 * half;
                const bool ok = jok && i < rhi;
                const float s = nbj[t] - 2.f * g[e], tt = na[e] - 2.f * g[e];
                if (PASS == 1) {
                    if (ok) { rth[e] = fminf(rth[e], s); cbest[t] = fminf(cbest[t], tt); }
                } else {
                    const bool cr = ok && s <= rth[e], cc = ok && tt <= cbest[t] + cband[t];
                    const unsigned long long m = __ballot(cr || cc);
                    if (m) {                                              // uniform
                        const int add = __popcll(m);
                        if (wcnt + add > WL) flush();
                        if (cr || cc)
                            wl[w][wcnt + __popcll(m & ((1ull << lane) - 1ull))] =
                                ((unsigned long long)i << 34) | ((unsigned long long)jcol[t] << 4) | (cr ? 2ull : 0ull) | (cc ? 1ull : 0ull);
                        wcnt += add;
                    }
                }
            }
        }
        if (PASS == 1) {
            // row minima over this workgroup's 128 columns: across the 32 lanes of a half, then one atomic per row
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = rth[e];
#pragma unroll
                for (int o = 16; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o));
                const int i = r0 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (l31 == 0 && i < rhi && v < __builtin_inff()) atomicMin(p.rowmin + i, ord_of(v));
            }
        }
        if (PASS == 2) flush();
    }
    if (PASS == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = fminf(cbest[t], __shfl_xor(cbest[t], 32));      // the two halves hold different rows of the same column
            if (half == 0 && jcol[t] < p.Nb && v < __builtin_inff()) atomicMin(p.colmin + jcol[t], ord_of(v));
        }
    }
}

int launch_nn32seg_if(const float* src, int Ns, const float* tgt, int Nt, unsigned long long* keys, int nCU, const int* run_if, hipStream_t s);

size_t mutual_prefilter_ws_bytes(int Na, int Nb) {
    const size_t n = (size_t)Na + Nb;
    return n * (sizeof(unsigned long long) + 3 * sizeof(float) + sizeof(unsigned)) + 256 + (16 * n + 4096) * sizeof(unsigned long long);
}

// keysA (Na) / keysB (Nb) receive the packed winners (the layout launch_mutual_compact<PACKED> reads); ws as sized above
int launch_mutual_prefilter(const float* a, int Na, const float* b, int Nb, void* ws, unsigned long long** keysA, unsigned long long** keysB,
                            int nCU, hipStream_t s) {
    MfArgs p;
    char* w = (char*)ws;
    p.a = a; p.b = b; p.Na = Na; p.Nb = Nb;
    p.keysA = (unsigned long long*)w; w += sizeof(unsigned long long) * (size_t)Na;
    p.keysB = (unsigned long long*)w; w += sizeof(unsigned long long) * (size_t)Nb;
    p.na2 = (float*)w; w += sizeof(float) * (size_t)Na;
    p.nb2 = (float*)w; w += sizeof(float) * (size_t)Nb;
    p.bandA = (float*)w; w += sizeof(float) * (size_t)Na;
    p.bandB = (float*)w; w += sizeof(float) * (size_t)Nb;
    p.rowmin = (unsigned*)w; w += sizeof(unsigned) * (size_t)Na;
    p.colmin = (unsigned*)w; w += sizeof(unsigned) * (size_t)Nb;
    p.maxn2 = (unsigned*)w; w += 2 * sizeof(unsigned);
    p.bad = (int*)w; w += sizeof(int);
    p.ncand = (unsigned*)w; w += sizeof(unsigned);
    p.cand = (unsigned long long*)(((size_t)w + 15) & ~(size_t)15);
    p.cap = 16u * (unsigned)(Na + Nb) + 4000u;
    *keysA = p.keysA; *keysB = p.keysB;
    HIPCHK(hipMemsetAsync(p.maxn2, 0, 16, s));                  // maxima, `bad`, candidate count
    const int n = Na + Nb;
    hipLaunchKernelGGL(mf_norms_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p);
    hipLaunchKernelGGL(mf_band_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p);
    const int cb = (Nb + 127) / 128;
    int segs = (2 * nCU + cb - 1) / cb;                         // about two workgroups per CU: the column fragments are loaded once per workgroup
    const int maxsegs = (Na + 127) / 128;
    segs = segs < 1 ? 1 : (segs > maxsegs ? maxsegs : segs);
    p.segRows = ((Na + segs - 1) / segs + 127) / 128 * 128;
    segs = (Na + p.segRows - 1) / p.segRows;
    hipLaunchKernelGGL(mf_gram_kernel<1>, dim3(cb, segs), dim3(256), 0, s, p);
    hipLaunchKernelGGL(mf_gram_kernel<2>, dim3(cb, segs), dim3(256), 0, s, p);
    hipLaunchKernelGGL(mf_exact_kernel, dim3(nCU > 0 ? 2 * nCU : 512), dim3(256), 0, s, p);
    HIPCHK(hipGetLastError());
    // inputs the pre-filter cannot take: the brute-force kernels run (they return at once otherwise)
    int rc;
    if ((rc = launch_nn32seg_if(a, Na, b, Nb, p.keysA, nCU, p.bad, s))) return rc;
    return launch_nn32seg_if(b, Nb, a, Na, p.keysB, nCU, p.bad, s);
}

}  // namespace yoho
