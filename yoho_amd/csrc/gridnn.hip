// Exact 1-nearest-neighbour in 3-D through a uniform hash grid - the same answers as the brute-force kernels
// (match.hip nn32seg_kernel<3,*>, estim.hip gather_nn_kernel), for callers that know the scale of the cloud:
//
//   60-fold feature transfer   YOHO_testset.py:153-166 (f64, 'L2'), simple_yoho/yoho_extract.py:33-39 (f32, 'SquareL2')
//
// The target cloud there is one point per voxel and every query is a point of the same cloud, so the winner sits within
// a voxel diagonal.  The grid (cell = the hint given to yoho_set_nn_grid; open-addressing table over packed cell
// coordinates, a linked list of points per cell) turns 5000 x n distance evaluations into 5000 x 125 cell probes.
//
// Exactness does not depend on the hint: a wave evaluates, with the brute-force kernels' own arithmetic, every point in
// the 5^3 cells around the query's cell and takes the minimum of (distance, index) - the reference's first minimum.
// Every point outside that block is more than 2 cells away, so the result is final when the winner is closer than
// 2 cells (with a 1e-4 relative slack that dwarfs every rounding involved).  Queries that fail the test (no point
// nearby, hint too small) go to a list and are redone by brute force, one workgroup per query.
// Compiled with -ffp-contract=off like match.hip / estim.hip.
#include "common.h"
#include "nnmath.h"

namespace yoho {

typedef unsigned long long u64;
constexpr u64 GN_EMPTY = ~0ull;
constexpr int GN_CLAMP = (1 << 20) - 1;

__device__ __forceinline__ int gn_cell(double x, double inv_cell) {
    double c = floor(x * inv_cell);                                   // monotone in x: |cell difference| >= 3 => distance > 2 cells
    c = fmin(fmax(c, -(double)GN_CLAMP), (double)GN_CLAMP);           // clamping keeps the map monotone (NaN -> -GN_CLAMP)
    return (int)c;
}
__device__ __forceinline__ int gn_clampi(int c) { return c < -GN_CLAMP ? -GN_CLAMP : (c > GN_CLAMP ? GN_CLAMP : c); }
__device__ __forceinline__ u64 gn_key(int cx, int cy, int cz) {
    return ((u64)(unsigned)(cx + (1 << 20)) << 42) | ((u64)(unsigned)(cy + (1 << 20)) << 21) | (u64)(unsigned)(cz + (1 << 20));
}
__device__ __forceinline__ unsigned gn_slot(u64 key, unsigned mask) { return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 33) & mask; }

__global__ void gn_clear_kernel(u64* keys, int* head, unsigned cap, int* ucount) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < cap) { keys[i] = GN_EMPTY; head[i] = -1; }
    if (i == 0) *ucount = 0;
}

__global__ void gn_build_kernel(const float* __restrict__ pts, int n, double inv_cell, u64* keys, int* head, int* next, unsigned mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 key = gn_key(gn_cell((double)pts[3 * (size_t)i], inv_cell), gn_cell((double)pts[3 * (size_t)i + 1], inv_cell),
                           gn_cell((double)pts[3 * (size_t)i + 2], inv_cell));
    unsigned s = gn_slot(key, mask);
    for (;;) {
        const u64 old = atomicCAS(&keys[s], GN_EMPTY, key);
        if (old == GN_EMPTY || old == key) break;
        s = (s + 1) & mask;
    }
    next[i] = atomicExch(&head[s], i);             // list order is arbitrary; the query takes a (distance, index) minimum
}

// MODE 0: f32 'SquareL2' (distance = squared), 1: f32 'L2', 2: f64 'L2' on keys rotated by Rg (group gather)
template <int MODE> struct GnMetric;
template <> struct GnMetric<0> {
    typedef float dist_t;
    float q[3];
    __device__ void load(const void* src, int i, const GnMat3&) { const float* s = (const float*)src + 3 * (size_t)i; q[0] = s[0]; q[1] = s[1]; q[2] = s[2]; }
    __device__ double pos(int a) const { return (double)q[a]; }
    __device__ float eval(const float* p) const { return dist2_f32<3>(q, p); }
    __device__ static bool resolved(float d, double lim2) { return (double)d < lim2; }
};
template <> struct GnMetric<1> {
    typedef float dist_t;
    float q[3];
    __device__ void load(const void* src, int i, const GnMat3&) { const float* s = (const float*)src + 3 * (size_t)i; q[0] = s[0]; q[1] = s[1]; q[2] = s[2]; }
    __device__ double pos(int a) const { return (double)q[a]; }
    __device__ float eval(const float* p) const { return dist_of_f32(dist2_f32<3>(q, p)); }
    __device__ static bool resolved(float d, double lim2) { return (double)d * (double)d - 1e-7 < lim2; }
};
template <> struct GnMetric<2> {
    typedef double dist_t;
    double q[3];
    __device__ void load(const void* src, int i, const GnMat3& R) {
        const double* kk = (const double*)src + 3 * (size_t)i;
#pragma unroll
        for (int a = 0; a < 3; ++a) q[a] = rotate_key_f64(kk, R.m + 3 * a);
    }
    __device__ double pos(int a) const { return q[a]; }
    __device__ double eval(const float* p) const { return sqrt(__dadd_rn(dist2_key_f64(q, (double)p[0], (double)p[1], (double)p[2]), 1e-7)); }
    __device__ static bool resolved(double d, double lim2) { return d * d - 1e-7 < lim2; }
};

struct GnArgs {
    const void* src; int Ns;
    const float* tgt; int Nt;
    GnMat3 R;
    double inv_cell, lim2;          // lim2 = (2 cell)^2 (1 - 1e-4)
    const u64* keys; const int* head; const int* next; unsigned mask;
    int64_t* idx; float* dist;      // MODE 0/1 outputs
    double* part_d; int* part_i;    // MODE 2 outputs (one "slice" for gather_merge_kernel)
    int* ulist; int* ucount;
};

template <typename T> __device__ __forceinline__ T gn_inf();
template <> __device__ __forceinline__ float gn_inf<float>() { return __builtin_inff(); }
template <> __device__ __forceinline__ double gn_inf<double>() { return __builtin_inf(); }

template <int MODE>
__device__ __forceinline__ void gn_store(const GnArgs& a, int q, typename GnMetric<MODE>::dist_t d, int i) {
    if (MODE == 2) { a.part_d[q] = (double)d; a.part_i[q] = i; }
    else { a.idx[q] = i; if (a.dist) a.dist[q] = (float)d; }
}

template <int MODE>
__global__ __launch_bounds__(256) void gn_query_kernel(GnArgs a) {
    typedef typename GnMetric<MODE>::dist_t D;
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= a.Ns) return;                                           // wave-uniform
    GnMetric<MODE> m;
    m.load(a.src, q, a.R);
    const int cx = gn_cell(m.pos(0), a.inv_cell), cy = gn_cell(m.pos(1), a.inv_cell), cz = gn_cell(m.pos(2), a.inv_cell);
    D bd = gn_inf<D>();
    int bi = 0x7FFFFFFF;
    for (int c = lane; c < 125; c += 64) {
        const u64 key = gn_key(gn_clampi(cx + c % 5 - 2), gn_clampi(cy + (c / 5) % 5 - 2), gn_clampi(cz + c / 25 - 2));
        unsigned s = gn_slot(key, a.mask);
        int i = -1;
        for (;;) {
            const u64 k = a.keys[s];
            if (k == key) { i = a.head[s]; break; }
            if (k == GN_EMPTY) break;
            s = (s + 1) & a.mask;
        }
        for (; i >= 0; i = a.next[i]) {
            const D d = m.eval(a.tgt + 3 * (size_t)i);
            if (d < bd || (d == bd && i < bi)) { bd = d; bi = i; }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const D od = __shfl_xor(bd, o);
        const int oi = __shfl_xor(bi, o);
        if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if (lane == 0) {
        if (bi != 0x7FFFFFFF && GnMetric<MODE>::resolved(bd, a.lim2)) gn_store<MODE>(a, q, bd, bi);
        else a.ulist[atomicAdd(a.ucount, 1)] = q;
    }
}

// brute force for the queries the grid could not settle: one workgroup per query, same arithmetic, (distance, index) minimum
template <int MODE>
__global__ __launch_bounds__(256) void gn_fallback_kernel(GnArgs a) {
    typedef typename GnMetric<MODE>::dist_t D;
    __shared__ D rd[256];
    __shared__ int ri[256];
    const int count = *a.ucount;
    for (int u = blockIdx.x; u < count; u += gridDim.x) {
        const int q = a.ulist[u];
        GnMetric<MODE> m;
        m.load(a.src, q, a.R);
        D bd = gn_inf<D>();
        int bi = 0x7FFFFFFF;
        for (int i = threadIdx.x; i < a.Nt; i += 256) {
            const D d = m.eval(a.tgt + 3 * (size_t)i);
            if (d < bd || (d == bd && i < bi)) { bd = d; bi = i; }
        }
        __syncthreads();
        rd[threadIdx.x] = bd; ri[threadIdx.x] = bi;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if (threadIdx.x < o) {
                const D od = rd[threadIdx.x + o];
                const int oi = ri[threadIdx.x + o];
                if (od < rd[threadIdx.x] || (od == rd[threadIdx.x] && oi < ri[threadIdx.x])) { rd[threadIdx.x] = od; ri[threadIdx.x] = oi; }
            }
            __syncthreads();
        }
        // all distances NaN (NaN query): the brute-force kernels answer index 0
        if (threadIdx.x == 0) gn_store<MODE>(a, q, rd[0], ri[0] == 0x7FFFFFFF ? 0 : ri[0]);
    }
}

size_t grid_nn_ws_bytes(int Ns, int Nt) {
    unsigned cap = 1024;
    while (cap < 2u * (unsigned)Nt) cap <<= 1;
    return (size_t)cap * 12 + ((size_t)Nt + Ns) * 4 + 256;
}

// ws: grid_nn_ws_bytes(Ns, Nt) bytes of scratch.  mode 0/1: idx/dist outputs; mode 2: part_d/part_i (Ns entries each).
int launch_grid_nn(int mode, const void* src, int Ns, const GnMat3* R, const float* tgt, int Nt, double cell, void* ws, int64_t* idx, float* dist,
                   double* part_d, int* part_i, int nCU, hipStream_t s) {
    unsigned cap = 1024;
    while (cap < 2u * (unsigned)Nt) cap <<= 1;
    char* p = (char*)ws;
    GnArgs a;
    u64* keys = (u64*)p; p += (size_t)cap * 8;
    int* head = (int*)p; p += (size_t)cap * 4;
    int* next = (int*)p; p += (size_t)Nt * 4;
    a.ulist = (int*)p; p += (size_t)Ns * 4;
    a.ucount = (int*)p;
    a.src = src; a.Ns = Ns; a.tgt = tgt; a.Nt = Nt;
    if (R) a.R = *R; else for (int i = 0; i < 9; ++i) a.R.m[i] = i % 4 == 0 ? 1.0 : 0.0;
    a.inv_cell = 1.0 / cell; a.lim2 = 4.0 * cell * cell * (1.0 - 1e-4);
    a.keys = keys; a.head = head; a.next = next; a.mask = cap - 1;
    a.idx = idx; a.dist = dist; a.part_d = part_d; a.part_i = part_i;
    hipLaunchKernelGGL(gn_clear_kernel, dim3(cap / 256), dim3(256), 0, s, keys, head, cap, a.ucount);
    hipLaunchKernelGGL(gn_build_kernel, dim3((Nt + 255) / 256), dim3(256), 0, s, tgt, Nt, a.inv_cell, keys, head, next, a.mask);
    const dim3 qg((Ns + 3) / 4), fg(nCU > 0 ? nCU : 256);
    switch (mode) {
        case 0: hipLaunchKernelGGL(gn_query_kernel<0>, qg, dim3(256), 0, s, a); hipLaunchKernelGGL(gn_fallback_kernel<0>, fg, dim3(256), 0, s, a); break;
        case 1: hipLaunchKernelGGL(gn_query_kernel<1>, qg, dim3(256), 0, s, a); hipLaunchKernelGGL(gn_fallback_kernel<1>, fg, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(gn_query_kernel<2>, qg, dim3(256), 0, s, a); hipLaunchKernelGGL(gn_fallback_kernel<2>, fg, dim3(256), 0, s, a); break;
    }
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- the feature transfer of one backbone pass in four launches (copy = blockIdx.y) -------------------------------------------
// simple_yoho/yoho_extract.py:33-39 per rotated copy b: q = (R_b keypoint).float(), nearest down-sampled point of copy b in fp32
// 'SquareL2', out[k, :, g0 + b] = feat_b[nn].  Per copy that was rotate -> clear -> build -> query -> fallback -> scatter, six
// launches of 5-15 us on 5000 queries: 90 launches per 15-copy pass.  Here the copies share the launches; the query kernel rotates
// its keypoint itself (the rotate kernel's expression: f64 fma chain, then the cast) and writes the feature row as soon as it knows
// the winner.  Same arithmetic, same (distance, index) minimum: the same rows.
constexpr int GT_BATCH = 16;
struct GtBatch {
    const double* pts; const int64_t* kidx; int K, nb;
    double R[GT_BATCH][9];
    const float* ds[GT_BATCH]; const float* feat[GT_BATCH]; int m[GT_BATCH];
    double inv_cell, lim2;
    u64* keys; int* head; unsigned cap;          // copy b: + b * cap
    int* next; int mmax;                         // copy b: + b * mmax
    int* ulist; int* ucount;                     // copy b: ulist + b * K, ucount[b]
    int g0; float* out;
};

__global__ void gt_clear_kernel(u64* keys, int* head, size_t total, int* ucount, int nb) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) { keys[i] = GN_EMPTY; head[i] = -1; }
    if (i < (size_t)nb) ucount[i] = 0;
}

__global__ void gt_build_kernel(GtBatch a) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= a.m[b]) return;
    const float* pts = a.ds[b];
    u64* keys = a.keys + (size_t)b * a.cap;
    int* head = a.head + (size_t)b * a.cap;
    const unsigned mask = a.cap - 1;
    const u64 key = gn_key(gn_cell((double)pts[3 * (size_t)i], a.inv_cell), gn_cell((double)pts[3 * (size_t)i + 1], a.inv_cell),
                           gn_cell((double)pts[3 * (size_t)i + 2], a.inv_cell));
    unsigned s = gn_slot(key, mask);
    for (;;) {
        const u64 old = atomicCAS(&keys[s], GN_EMPTY, key);
        if (old == GN_EMPTY || old == key) break;
        s = (s + 1) & mask;
    }
    a.next[(size_t)b * a.mmax + i] = atomicExch(&head[s], i);
}

// the rotated keypoint as the rotate kernel (sparse.hip rotate_sel_kernel / point_of) forms it: f64 fma chain per coordinate, then float
__device__ __forceinline__ void gt_query_point(const GtBatch& a, int b, int k, float (&q)[3]) {
    const size_t r = (size_t)a.kidx[k];
    const double p0 = a.pts[3 * r], p1 = a.pts[3 * r + 1], p2 = a.pts[3 * r + 2];
    const double* R = a.R[b];
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = (float)fma(p2, R[3 * c + 2], fma(p1, R[3 * c + 1], p0 * R[3 * c]));
}

// out[k, :, g] = feat[row, :] by the first 32 threads of the caller's wave / workgroup (group_scatter_kernel's clamp)
__device__ __forceinline__ void gt_scatter(const GtBatch& a, int b, int k, int row, int t) {
    row = row < 0 ? 0 : (row >= a.m[b] ? a.m[b] - 1 : row);
    if (t < F) a.out[((size_t)k * F + t) * G + a.g0 + b] = a.feat[b][(size_t)row * F + t];
}

__global__ __launch_bounds__(256) void gt_query_kernel(GtBatch a) {
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= a.K) return;                                            // wave-uniform
    GnMetric<0> m;
    gt_query_point(a, b, k, m.q);
    const u64* keys = a.keys + (size_t)b * a.cap;
    const int* head = a.head + (size_t)b * a.cap;
    const int* next = a.next + (size_t)b * a.mmax;
    const float* tgt = a.ds[b];
    const unsigned mask = a.cap - 1;
    const int cx = gn_cell(m.pos(0), a.inv_cell), cy = gn_cell(m.pos(1), a.inv_cell), cz = gn_cell(m.pos(2), a.inv_cell);
    float bd = gn_inf<float>();
    int bi = 0x7FFFFFFF;
    for (int c = lane; c < 125; c += 64) {
        const u64 key = gn_key(gn_clampi(cx + c % 5 - 2), gn_clampi(cy + (c / 5) % 5 - 2), gn_clampi(cz + c / 25 - 2));
        unsigned s = gn_slot(key, mask);
        int i = -1;
        for (;;) {
            const u64 kk = keys[s];
            if (kk == key) { i = head[s]; break; }
            if (kk == GN_EMPTY) break;
            s = (s + 1) & mask;
        }
        for (; i >= 0; i = next[i]) {
            const float d = m.eval(tgt + 3 * (size_t)i);
            if (d < bd || (d == bd && i < bi)) { bd = d; bi = i; }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float od = __shfl_xor(bd, o);
        const int oi = __shfl_xor(bi, o);
        if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
    }
    if (bi != 0x7FFFFFFF && GnMetric<0>::resolved(bd, a.lim2)) gt_scatter(a, b, k, bi, lane);       // every lane holds the winner
    else if (lane == 0) a.ulist[(size_t)b * a.K + atomicAdd(a.ucount + b, 1)] = k;
}

__global__ __launch_bounds__(256) void gt_fallback_kernel(GtBatch a) {
    __shared__ float rd[256];
    __shared__ int ri[256];
    const int b = blockIdx.y;
    const int count = a.ucount[b];
    const float* tgt = a.ds[b];
    for (int u = blockIdx.x; u < count; u += gridDim.x) {
        const int k = a.ulist[(size_t)b * a.K + u];
        GnMetric<0> m;
        gt_query_point(a, b, k, m.q);
        float bd = gn_inf<float>();
        int bi = 0x7FFFFFFF;
        for (int i = threadIdx.x; i < a.m[b]; i += 256) {
            const float d = m.eval(tgt + 3 * (size_t)i);
            if (d < bd || (d == bd && i < bi)) { bd = d; bi = i; }
        }
        __syncthreads();
        rd[threadIdx.x] = bd; ri[threadIdx.x] = bi;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if (threadIdx.x < o) {
                const float od = rd[threadIdx.x + o];
                const int oi = ri[threadIdx.x + o];
                if (od < rd[threadIdx.x] || (od == rd[threadIdx.x] && oi < ri[threadIdx.x])) { rd[threadIdx.x] = od; ri[threadIdx.x] = oi; }
            }
            __syncthreads();
        }
        gt_scatter(a, b, k, ri[0] == 0x7FFFFFFF ? 0 : ri[0], threadIdx.x);       // all distances NaN: the brute-force kernels answer index 0
    }
}

size_t grid_transfer_ws_bytes(int K, int nb, int mmax) {
    unsigned cap = 1024;
    while (cap < 2u * (unsigned)mmax) cap <<= 1;
    return ((size_t)cap * 12 + ((size_t)mmax + K) * 4) * nb + 1024;
}

// nb <= 64 copies (launched GT_BATCH at a time); ws: grid_transfer_ws_bytes(K, nb, max m) bytes
int launch_grid_transfer_batch(const double* pts, const int64_t* kidx, int K, const double* R_host, int nb, const float* const* ds,
                               const float* const* feat, const int* m, int g0, float* out, double cell, void* ws, int nCU, hipStream_t s) {
    int mmax = 1;
    for (int b = 0; b < nb; ++b) mmax = m[b] > mmax ? m[b] : mmax;
    unsigned cap = 1024;
    while (cap < 2u * (unsigned)mmax) cap <<= 1;
    char* p = (char*)ws;
    u64* keys = (u64*)p; p += (size_t)cap * 8 * nb;
    int* head = (int*)p; p += (size_t)cap * 4 * nb;
    int* next = (int*)p; p += (size_t)mmax * 4 * nb;
    int* ulist = (int*)p; p += (size_t)K * 4 * nb;
    int* ucount = (int*)p;
    {
        const size_t total = (size_t)cap * nb;
        hipLaunchKernelGGL(gt_clear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, keys, head, total, ucount, nb);
    }
    for (int b0 = 0; b0 < nb; b0 += GT_BATCH) {
        const int nbc = nb - b0 < GT_BATCH ? nb - b0 : GT_BATCH;
        GtBatch a;
        a.pts = pts; a.kidx = kidx; a.K = K; a.nb = nbc;
        for (int b = 0; b < GT_BATCH; ++b) {
            const int sb = b < nbc ? b0 + b : b0;
            for (int i = 0; i < 9; ++i) a.R[b][i] = R_host[9 * (size_t)sb + i];
            a.ds[b] = ds[sb]; a.feat[b] = feat[sb]; a.m[b] = m[sb];
        }
        a.inv_cell = 1.0 / cell; a.lim2 = 4.0 * cell * cell * (1.0 - 1e-4);
        a.keys = keys + (size_t)b0 * cap; a.head = head + (size_t)b0 * cap; a.cap = cap;
        a.next = next + (size_t)b0 * mmax; a.mmax = mmax;
        a.ulist = ulist + (size_t)b0 * K; a.ucount = ucount + b0;
        a.g0 = g0 + b0; a.out = out;
        hipLaunchKernelGGL(gt_build_kernel, dim3((mmax + 255) / 256, nbc), dim3(256), 0, s, a);
        hipLaunchKernelGGL(gt_query_kernel, dim3((K + 3) / 4, nbc), dim3(256), 0, s, a);
        hipLaunchKernelGGL(gt_fallback_kernel, dim3(nCU > 64 ? 64 : (nCU > 0 ? nCU : 64), nbc), dim3(256), 0, s, a);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
