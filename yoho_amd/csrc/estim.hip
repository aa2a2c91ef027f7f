// Transformation estimators (f64 end to end, as the reference: keypoints come from open3d as f64).
//
//   hyp_kernel      quat -> [R|t] per match      tests/extractor.py:187-199, utils/r_eval.py:94-110
//   score_kernel    overlap_cal per hypothesis   tests/estimator.py:286-290 (yohoo), :66-70 (yohoc)
//   argbest_kernel  strict '>' running best      tests/estimator.py:330-336 / :131-137
//   kabsch_kernel   Threepps2Tran                tests/estimator.py:55-63
//   gather_kernel   60-fold FCGF feature gather  YOHO_testset.py:153-166
// This file is compiled with -ffp-contract=off (yoho_amd/build.py): only explicit fma() fuses.
#include "common.h"
#include "nnmath.h"

namespace yoho {

// ---- quaternion -> rotation, evaluated in fp32 in the reference's operation order (the
// quaternion is a float32 numpy vector, so utils/r_eval.py:94-110 runs in fp32) ----------------
__device__ __forceinline__ void quat2mat_f32(const float* q, double* m) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float two = 2.0f;
#define MUL(a, b) __fmul_rn(a, b)
#define SUB(a, b) __fsub_rn(a, b)
#define ADD(a, b) __fadd_rn(a, b)
    m[0] = (double)SUB(SUB(1.0f, MUL(MUL(two, y), y)), MUL(MUL(two, z), z));
    m[1] = (double)SUB(MUL(MUL(two, x), y), MUL(MUL(two, z), w));
    m[2] = (double)ADD(MUL(MUL(two, x), z), MUL(MUL(two, y), w));
    m[3] = (double)ADD(MUL(MUL(two, x), y), MUL(MUL(two, z), w));
    m[4] = (double)SUB(SUB(1.0f, MUL(MUL(two, x), x)), MUL(MUL(two, z), z));
    m[5] = (double)SUB(MUL(MUL(two, y), z), MUL(MUL(two, x), w));
    m[6] = (double)SUB(MUL(MUL(two, x), z), MUL(MUL(two, y), w));
    m[7] = (double)ADD(MUL(MUL(two, y), z), MUL(MUL(two, x), w));
    m[8] = (double)SUB(SUB(1.0f, MUL(MUL(two, x), x)), MUL(MUL(two, y), y));
#undef MUL
#undef SUB
#undef ADD
}

__global__ __launch_bounds__(256) void hyp_kernel(const float* __restrict__ quat, const int64_t* __restrict__ idx,
                                                  const double* __restrict__ k0, const double* __restrict__ k1,
                                                  const double* __restrict__ Rg64, int M, double* __restrict__ T) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    double rr[9];
    quat2mat_f32(quat + (size_t)m * 4, rr);
    long long gi = idx[m];
    gi = gi < 0 ? 0 : (gi > 59 ? 59 : gi);
    const double* A = Rg64 + gi * 9;
    double R[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = fma(rr[i * 3 + 2], A[6 + j], fma(rr[i * 3 + 1], A[3 + j], rr[i * 3] * A[j]));
    const double* p0 = k0 + (size_t)m * 3;
    const double* p1 = k1 + (size_t)m * 3;
    double* o = T + (size_t)m * 12;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double d = fma(p1[2], R[i * 3 + 2], fma(p1[1], R[i * 3 + 1], p1[0] * R[i * 3]));
        o[i * 4 + 0] = R[i * 3 + 0]; o[i * 4 + 1] = R[i * 3 + 1]; o[i * 4 + 2] = R[i * 3 + 2];
        o[i * 4 + 3] = __dsub_rn(p0[i], d);
    }
}

// inlier test of one match under one [R|t] (3x4 row-major)
__device__ __forceinline__ bool inlier(const double* T, const double* a, const double* b, double d2thr) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double p = __dadd_rn(fma(b[2], T[i * 4 + 2], fma(b[1], T[i * 4 + 1], b[0] * T[i * 4])), T[i * 4 + 3]);
        const double e = __dsub_rn(a[i], p);
        const double q = __dmul_rn(e, e);
        s = i == 0 ? q : __dadd_rn(s, q);
    }
    return s < d2thr;
}

__device__ __forceinline__ int block_count(bool flag_acc_unused, int local, int* red) {
    // sum `local` over a 256-thread workgroup
    for (int o = 32; o >= 1; o >>= 1) local += __shfl_xor(local, o);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wv] = local;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// one workgroup per hypothesis h: counts[h] = #{m : |k0[m] - (R k1[m] + t)|^2 < d^2}, T taken at order[h]
__global__ __launch_bounds__(256) void score_kernel(const double* __restrict__ k0, const double* __restrict__ k1, int M,
                                                    const double* __restrict__ T, const int64_t* __restrict__ order,
                                                    double d2thr, int32_t* __restrict__ counts) {
    __shared__ double Ts[12];
    __shared__ int red[4];
    const int h = blockIdx.x;
    const size_t ti = order ? (size_t)order[h] : (size_t)h;
    if (threadIdx.x < 12) Ts[threadIdx.x] = T[ti * 12 + threadIdx.x];
    __syncthreads();
    int local = 0;
    for (int m = threadIdx.x; m < M; m += 256) local += inlier(Ts, k0 + (size_t)m * 3, k1 + (size_t)m * 3, d2thr) ? 1 : 0;
    const int tot = block_count(false, local, red);
    if (threadIdx.x == 0) counts[h] = tot;
}

// first strict maximum over counts[0..H): reproduces `if overlap > best_overlap` with best = 0 initially
__global__ __launch_bounds__(256) void argbest_kernel(const int32_t* __restrict__ counts, int H, int* __restrict__ best_h,
                                                      int* __restrict__ best_count) {
    __shared__ int sv[4];
    __shared__ int si[4];
    int bv = 0, bi = -1;
    for (int h = threadIdx.x; h < H; h += 256) {
        const int c = counts[h];
        if (c > bv) { bv = c; bi = h; }
    }
    // larger count wins, equal (positive) counts go to the lower index: butterfly inside the wave, then the four waves
    for (int o = 32; o >= 1; o >>= 1) {
        const int ov = __shfl_xor(bv, o), oi = __shfl_xor(bi, o);
        if (ov > bv || (ov == bv && ov > 0 && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int v = 0, i = -1;
        for (int k = 0; k < 4; ++k) {
            if (sv[k] > v) { v = sv[k]; i = si[k]; }
            else if (sv[k] == v && v > 0 && si[k] < i) i = si[k];
        }
        *best_h = i < 0 ? 0 : i;
        *best_count = v;
    }
}

// ---- 3-point Kabsch (rank <= 2 covariance) --------------------------------------------------
// m = (k1 - c1)^T (k0 - c0) = U S V^T;  reference R = V U^T with no determinant fix.  One-sided
// Jacobi (Hestenes) on the columns of m gives  m V = U S;  the third singular direction is taken
// as the cross product so that det(U) = det(V) = +1 (proper rotation).  `reflect` reproduces the
// reference's det = -1 answer:  R_ref = R (I - 2 n n^T), n = normal of the centred k1 triangle.
__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ __forceinline__ void any_orthogonal(const double* a, double* o) {
    // unit vector orthogonal to unit vector a
    double e[3] = {0.0, 0.0, 0.0};
    const double ax = fabs(a[0]), ay = fabs(a[1]), az = fabs(a[2]);
    e[(ax <= ay && ax <= az) ? 0 : (ay <= az ? 1 : 2)] = 1.0;
    cross3(a, e, o);
    const double n = sqrt(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    o[0] /= n; o[1] /= n; o[2] /= n;
}

__device__ void kabsch3(const double* a0, const double* a1, bool reflect, double* T) {
    // a0: 3 points of fragment 0 (target), a1: 3 points of fragment 1 (source), row-major (3,3)
    double c0[3], c1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        c0[j] = ((a0[j] + a0[3 + j]) + a0[6 + j]) / 3.0;
        c1[j] = ((a1[j] + a1[3 + j]) + a1[6 + j]) / 3.0;
    }
    double A[9];       // columns of A are rotated by Jacobi; A = m
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
#pragma unroll
            for (int p = 0; p < 3; ++p) s += (a1[p * 3 + i] - c1[i]) * (a0[p * 3 + j] - c0[j]);
            A[i * 3 + j] = s;
        }
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 30; ++sweep) {
        double offmax = 0.0;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double al = A[p] * A[p] + A[3 + p] * A[3 + p] + A[6 + p] * A[6 + p];
            const double be = A[q] * A[q] + A[3 + q] * A[3 + q] + A[6 + q] * A[6 + q];
            const double ga = A[p] * A[q] + A[3 + p] * A[3 + q] + A[6 + p] * A[6 + q];
            const double lim = 1e-30 + 1e-16 * sqrt(al * be);
            if (fabs(ga) > lim) {
                offmax = fmax(offmax, fabs(ga) / (sqrt(al * be) + 1e-300));
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double ap = A[r * 3 + p], aq = A[r * 3 + q];
                    A[r * 3 + p] = cs * ap - sn * aq;
                    A[r * 3 + q] = sn * ap + cs * aq;
                    const double vp = V[r * 3 + p], vq = V[r * 3 + q];
                    V[r * 3 + p] = cs * vp - sn * vq;
                    V[r * 3 + q] = sn * vp + cs * vq;
                }
            }
        }
        if (offmax < 1e-15) break;
    }
    // singular values = column norms; order the two largest
    double sg[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) sg[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
    int i1 = 0;
    if (sg[1] > sg[i1]) i1 = 1;
    if (sg[2] > sg[i1]) i1 = 2;
    int i2 = (i1 + 1) % 3, i3 = (i1 + 2) % 3;
    if (sg[i3] > sg[i2]) { const int t = i2; i2 = i3; i3 = t; }
    double u1[3], u2[3], u3[3], v1[3], v2[3], v3[3];
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (sg[i1] > 1e-300) {
#pragma unroll
        for (int r = 0; r < 3; ++r) { u1[r] = A[r * 3 + i1] / sg[i1]; v1[r] = V[r * 3 + i1]; }
        if (sg[i2] > 1e-13 * sg[i1]) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { u2[r] = A[r * 3 + i2] / sg[i2]; v2[r] = V[r * 3 + i2]; }
            // re-orthogonalise u2 against u1 (cheap insurance for nearly collinear triangles)
            const double d = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
            u2[0] -= d * u1[0]; u2[1] -= d * u1[1]; u2[2] -= d * u1[2];
            const double n = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
            u2[0] /= n; u2[1] /= n; u2[2] /= n;
        } else {            // rank 1 (two sampled matches coincide): roll about u1/v1 is arbitrary
            any_orthogonal(u1, u2);
            any_orthogonal(v1, v2);
        }
        cross3(u1, u2, u3);
        cross3(v1, v2, v3);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) R[i * 3 + j] = v1[i] * u1[j] + v2[i] * u2[j] + v3[i] * u3[j];
        if (reflect) {      // R <- R (I - 2 u3 u3^T)
            double Ru[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) Ru[i] = R[i * 3] * u3[0] + R[i * 3 + 1] * u3[1] + R[i * 3 + 2] * u3[2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) R[i * 3 + j] -= 2.0 * Ru[i] * u3[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        T[i * 4] = R[i * 3]; T[i * 4 + 1] = R[i * 3 + 1]; T[i * 4 + 2] = R[i * 3 + 2];
        T[i * 4 + 3] = c0[i] - (c1[0] * R[i * 3] + c1[1] * R[i * 3 + 1] + c1[2] * R[i * 3 + 2]);
    }
}

// one workgroup per RANSAC iteration: Kabsch on the sampled triple, then the inlier vote
__global__ __launch_bounds__(256) void kabsch_score_kernel(const double* __restrict__ k0, const double* __restrict__ k1, int M,
                                                           const int64_t* __restrict__ triples, const uint8_t* __restrict__ reflect,
                                                           double d2thr, double* __restrict__ T_out, int32_t* __restrict__ counts) {
    __shared__ double Ts[12];
    __shared__ int red[4];
    const int it = blockIdx.x;
    if (threadIdx.x == 0) {
        double a0[9], a1[9];
        for (int p = 0; p < 3; ++p) {
            long long mi = triples[(size_t)it * 3 + p];
            mi = mi < 0 ? 0 : (mi >= M ? M - 1 : mi);
            for (int j = 0; j < 3; ++j) { a0[p * 3 + j] = k0[(size_t)mi * 3 + j]; a1[p * 3 + j] = k1[(size_t)mi * 3 + j]; }
        }
        double T[12];
        kabsch3(a0, a1, reflect ? reflect[it] != 0 : false, T);
        for (int i = 0; i < 12; ++i) { Ts[i] = T[i]; T_out[(size_t)it * 12 + i] = T[i]; }
    }
    __syncthreads();
    int local = 0;
    for (int m = threadIdx.x; m < M; m += 256) local += inlier(Ts, k0 + (size_t)m * 3, k1 + (size_t)m * 3, d2thr) ? 1 : 0;
    const int tot = block_count(false, local, red);
    if (threadIdx.x == 0) counts[it] = tot;
}

__global__ void pick_T_kernel(const double* __restrict__ T_all, const int* __restrict__ best_h, const int* __restrict__ best_count,
                              double* __restrict__ best_T, int* __restrict__ best_iter) {
    const int i = threadIdx.x;
    const bool any = *best_count > 0;
    if (i < 12) best_T[i] = any ? T_all[(size_t)(*best_h) * 12 + i] : ((i % 5 == 0) ? 1.0 : 0.0);   // eye(4)[:3]
    if (i == 0) *best_iter = any ? *best_h + 1 : 0;
}

// ---- YOHO-C with the sampling on the device (yoho_c_ransac_device) -----------------------------
// tests/estimator.py:34-51 (DR_statictic) + :119-128 (the two np.random.choice draws of an iteration) without the host:
//   1. cstat_kernel (one workgroup): histogram of the coarse-rotation index over the matches, the bucket lists in
//      ascending match order (R_index_pre_statistic), the weights p_b = n (n - 0.01)(n - 0.02), n = count / 100 for
//      count >= 2 (else 0) as a running sum (cdf, f64, summed b = 0..59), and the matched keypoints gathered once.
//      sum p < 1e-4 -> no estimate (the reference saves eye(4) with recalltime 50001, :103-107).
//   2. kabsch_sample_kernel (one workgroup per iteration it): counter-based Philox4x32-10 keyed by the caller's seed,
//      counter = it -> four 32-bit words: (w0, w1) -> u in [0,1) with 53 bits -> bucket b = first with cdf[b] > u * total
//      (np.random.choice(range(60), p) is a cdf search as well), w1', w2', w3' of a second block -> three members of the
//      bucket, with replacement (np.random.choice(bucket, 3)).  Every bucket with p > 0 has >= 2 members, so each draw
//      is an accepted iteration (:123-125 never skips).  Then Kabsch + vote as kabsch_score_kernel.
// The random stream is Philox, not numpy's MT19937: for a given seed the sampled triples are those of
// oracle/yoho_oracle.py:yohoc_device_triples (bit-exact, tested), not those of np.random.seed(seed).
struct CStat {
    int count[G];
    int start[G];
    double cdf[G];
    int valid;
    int pad;
};

// every match in parallel: its two keypoints gathered into contiguous (M,3) arrays, its coarse rotation clamped to a byte
__global__ __launch_bounds__(256) void cprep_kernel(const int64_t* __restrict__ dr, int M, const double* __restrict__ keys0,
                                                    const double* __restrict__ keys1, const int64_t* __restrict__ i0,
                                                    const int64_t* __restrict__ i1, int istride, unsigned char* __restrict__ dr8,
                                                    double* __restrict__ k0m, double* __restrict__ k1m) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    long long v = dr[m];
    dr8[m] = (unsigned char)(v < 0 ? 0 : (v > G - 1 ? G - 1 : v));
    const size_t r0 = i0 ? (size_t)i0[(size_t)m * istride] : (size_t)m;
    const size_t r1 = i1 ? (size_t)i1[(size_t)m * istride] : (size_t)m;
#pragma unroll
    for (int j = 0; j < 3; ++j) { k0m[(size_t)m * 3 + j] = keys0[r0 * 3 + j]; k1m[(size_t)m * 3 + j] = keys1[r1 * 3 + j]; }
}

// one workgroup of 16 waves: histogram, weights / running sum, bucket lists in ascending match order (a stable counting sort).
// Wave w owns the contiguous segment [w * seg, (w + 1) * seg) of the matches.  Pass 1: per-wave bucket counts (the lanes of a
// 64-match step that share a bucket find each other with six ballots; the lowest of them adds their number to the wave's own
// LDS row - no atomics).  Then 60 threads turn the counts into per-wave cursors (bucket start + the counts of the waves before),
// thread 0 into the weights and their running sum.  Pass 2: every wave walks its segment again and places match m at
// cursor[bucket] + (number of lower lanes of the step in the same bucket): ascending match order inside every bucket, exactly
// the order of the reference's R_index_pre_statistic lists (tests/estimator.py:34-51).
constexpr int CS_WAVES = 16;

__device__ __forceinline__ unsigned long long same_bucket_lanes(int b, bool valid) {
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
        const bool on = (b >> bit) & 1;
        const unsigned long long bal = __ballot(on);
        same &= on ? bal : ~bal;
    }
    return valid ? same : 0ull;
}

__global__ __launch_bounds__(64 * CS_WAVES) void cstat_kernel(const unsigned char* __restrict__ dr8, int M, CStat* __restrict__ st, int* __restrict__ members) {
    __shared__ volatile int wcnt[CS_WAVES][64];      // pass 1: counts of wave w; pass 2: its cursors
    __shared__ int tot[G];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int seg = ((M + CS_WAVES - 1) / CS_WAVES + 63) / 64 * 64;
    const int m0 = w * seg, m1 = min(M, m0 + seg);
    wcnt[w][lane] = 0;
    __builtin_amdgcn_wave_barrier();
    const unsigned long long below = (1ull << lane) - 1ull;
    for (int base = m0; base < m1; base += 64) {
        const int m = base + lane;
        const bool valid = m < m1;
        const int b = valid ? dr8[m] : 0;
        const unsigned long long same = same_bucket_lanes(b, valid);
        if (valid && (same & below) == 0ull) wcnt[w][b] += __popcll(same);       // one writer per bucket and wave
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (tid < G) {
        int n = 0;
        for (int k = 0; k < CS_WAVES; ++k) n += wcnt[k][tid];
        tot[tid] = n;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        double run = 0.0;
        for (int j = 0; j < G; ++j) {
            const int n = tot[j];
            st->start[j] = acc;
            st->count[j] = n;
            tot[j] = acc;
            acc += n;
            double p = 0.0;
            if (n >= 2) {
                const double num = (double)n / 100.0;
                p = __dmul_rn(__dmul_rn(num, __dsub_rn(num, 0.01)), __dsub_rn(num, 0.02));
            }
            run = __dadd_rn(run, p);
            st->cdf[j] = run;
        }
        st->valid = run < 1e-4 ? 0 : 1;
    }
    __syncthreads();
    if (tid < G) {                                   // counts -> cursors: bucket start + the counts of the waves before
        int acc = tot[tid];
        for (int k = 0; k < CS_WAVES; ++k) { const int n = wcnt[k][tid]; wcnt[k][tid] = acc; acc += n; }
    }
    __syncthreads();
    for (int base = m0; base < m1; base += 64) {
        const int m = base + lane;
        const bool valid = m < m1;
        const int b = valid ? dr8[m] : 0;
        const unsigned long long same = same_bucket_lanes(b, valid);
        int cur = 0;
        if (valid) cur = wcnt[w][b];
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            members[cur + __popcll(same & below)] = m;
            if ((same & below) == 0ull) wcnt[w][b] = cur + __popcll(same);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ __launch_bounds__(256) void kabsch_sample_kernel(const double* __restrict__ k0, const double* __restrict__ k1, int M,
                                                            const CStat* __restrict__ st, const int* __restrict__ members,
                                                            unsigned seed_lo, unsigned seed_hi, double d2thr, double* __restrict__ T_out,
                                                            int32_t* __restrict__ counts, int64_t* __restrict__ triples_out) {
    __shared__ double Ts[12];
    __shared__ int red[4];
    const int it = blockIdx.x;
    if (!st->valid) {                                       // no bucket with two matches: nothing to sample
        if (threadIdx.x == 0) counts[it] = 0;
        return;
    }
    if (threadIdx.x == 0) {
        unsigned w[4], v[4];
        philox4x32_10((unsigned)it, 0u, 0u, 0u, seed_lo, seed_hi, w);
        philox4x32_10((unsigned)it, 1u, 0u, 0u, seed_lo, seed_hi, v);
        const double u = (double)(((unsigned long long)w[0] << 21) | (w[1] >> 11)) * 0x1p-53;
        const double thr = __dmul_rn(u, st->cdf[G - 1]);
        int b = G - 1;
        for (int j = G - 1; j >= 0; --j) if (st->cdf[j] > thr) b = j;          // first bucket whose running sum exceeds thr
        const int n = st->count[b], s0 = st->start[b];
        double a0[9], a1[9];
        for (int p = 0; p < 3; ++p) {
            const int mi = members[s0 + (int)(((unsigned long long)v[p] * (unsigned)n) >> 32)];
            if (triples_out) triples_out[(size_t)it * 3 + p] = mi;
            for (int j = 0; j < 3; ++j) { a0[p * 3 + j] = k0[(size_t)mi * 3 + j]; a1[p * 3 + j] = k1[(size_t)mi * 3 + j]; }
        }
        double T[12];
        kabsch3(a0, a1, false, T);
        for (int i = 0; i < 12; ++i) { Ts[i] = T[i]; T_out[(size_t)it * 12 + i] = T[i]; }
    }
    __syncthreads();
    int local = 0;
    for (int m = threadIdx.x; m < M; m += 256) local += inlier(Ts, k0 + (size_t)m * 3, k1 + (size_t)m * 3, d2thr) ? 1 : 0;
    const int tot = block_count(false, local, red);
    if (threadIdx.x == 0) counts[it] = tot;
}

__global__ void pick_T_dev_kernel(const double* __restrict__ T_all, const int* __restrict__ best_h, const int* __restrict__ best_count,
                                  const CStat* __restrict__ st, double* __restrict__ best_T, int* __restrict__ best_iter) {
    const int i = threadIdx.x;
    const bool any = st->valid && *best_count > 0;
    if (i < 12) best_T[i] = any ? T_all[(size_t)(*best_h) * 12 + i] : ((i % 5 == 0) ? 1.0 : 0.0);   // eye(4)[:3]
    if (i == 0) *best_iter = !st->valid ? 50001 : (any ? *best_h + 1 : 0);                               // tests/estimator.py:107
}

// ---- group-feature gather (one group element) ------------------------------------------------
// kr = keys @ Rg^T (f64); NN among pts (f32, widened) with sqrt(D2 + 1e-7) in f64; copy feature row.
// N-body style: a thread owns GG_KR rotated keys in registers and walks a slice of the cloud that the workgroup
// streams through LDS (every lane reads the same point: one broadcast LDS read feeds 64 x GG_KR distance
// evaluations).  The cloud is cut into `nslice` slices across blockIdx.y; a second kernel merges the per-slice
// winners (distance, then lower index) and copies the feature row.
constexpr int GG_KR = 2, GG_TT = 2048;
typedef GnMat3 Mat3;

__global__ __launch_bounds__(256) void gather_nn_kernel(const double* __restrict__ keys, int K, const float* __restrict__ pts, int n,
                                                        Mat3 Rg, int slice_len, double* __restrict__ part_d, int* __restrict__ part_i) {
    __shared__ double tile[GG_TT * 3];            // cloud points widened to f64 once per tile
    const int k0 = (blockIdx.x * 256 + threadIdx.x) * GG_KR;
    double kr[GG_KR][3], best[GG_KR];
    int besti[GG_KR];
#pragma unroll
    for (int q = 0; q < GG_KR; ++q) {
        const int kc = k0 + q < K ? k0 + q : K - 1;
        const double* kk = keys + (size_t)kc * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) kr[q][i] = rotate_key_f64(kk, Rg.m + 3 * i);
        best[q] = __builtin_inf();
        besti[q] = 0;
    }
    const int s0 = blockIdx.y * slice_len;
    const int s1 = s0 + slice_len < n ? s0 + slice_len : n;
    // The reference takes argmin of sqrt(D2 + 1e-7) (first minimum).  sqrt is monotone, so a later candidate can
    // only win with a smaller D2; the f64 sqrt is evaluated only when two D2 are so close that rounding could map
    // them to the same distance - then the earlier index must stay.
    for (int t0 = s0; t0 < s1; t0 += GG_TT) {
        const int nt = s1 - t0 < GG_TT ? s1 - t0 : GG_TT;
        __syncthreads();
        for (int i = threadIdx.x; i < nt * 3; i += 256) tile[i] = (double)pts[(size_t)t0 * 3 + i];
        __syncthreads();
        for (int t = 0; t < nt; ++t) {
            const double px = tile[t * 3], py = tile[t * 3 + 1], pz = tile[t * 3 + 2];
#pragma unroll
            for (int q = 0; q < GG_KR; ++q) {
                const double s = dist2_key_f64(kr[q], px, py, pz);
                if (s < best[q]) {
                    if (s < best[q] * (1.0 - 1e-12) || sqrt(__dadd_rn(s, 1e-7)) < sqrt(__dadd_rn(best[q], 1e-7))) { best[q] = s; besti[q] = t0 + t; }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < GG_KR; ++q)
        if (k0 + q < K) {
            part_d[(size_t)blockIdx.y * K + k0 + q] = sqrt(__dadd_rn(best[q], 1e-7));
            part_i[(size_t)blockIdx.y * K + k0 + q] = besti[q];
        }
}

// merge the slices of one key (16 threads per key: each also copies 2 of the 32 feature channels)
__global__ __launch_bounds__(256) void gather_merge_kernel(const double* __restrict__ part_d, const int* __restrict__ part_i, int K,
                                                           int nslice, const float* __restrict__ feat, int g,
                                                           float* __restrict__ out, int64_t* __restrict__ nn_idx) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4), sp = threadIdx.x & 15;
    if (row >= K) return;
    double bd = part_d[row];
    int bi = part_i[row];
    for (int k = 1; k < nslice; ++k) {                 // slices are in ascending index order: strict < keeps the first minimum
        const double d = part_d[(size_t)k * K + row];
        if (d < bd) { bd = d; bi = part_i[(size_t)k * K + row]; }
    }
    if (sp == 0 && nn_idx) nn_idx[row] = bi;
    const float* fr = feat + (size_t)bi * F;
    float* o = out + (size_t)row * F * G + g;
    o[(size_t)(2 * sp) * G] = fr[2 * sp];
    o[(size_t)(2 * sp + 1) * G] = fr[2 * sp + 1];
}

// out[row, :, g] = feat[idx[row], :]: the feature transfer of one group element once the nearest neighbours are known
// (simple_yoho/yoho_extract.py:38-39, :52); 16 threads per row, 2 channels each
__global__ __launch_bounds__(256) void group_scatter_kernel(const float* __restrict__ feat, const int64_t* __restrict__ idx, int K, int n, int g,
                                                            float* __restrict__ out) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4), sp = threadIdx.x & 15;
    if (row >= K) return;
    long long bi = idx[row];
    bi = bi < 0 ? 0 : (bi >= n ? n - 1 : bi);
    const float2 v = *reinterpret_cast<const float2*>(feat + (size_t)bi * F + 2 * sp);
    float* o = out + (size_t)row * F * G + g;
    o[(size_t)(2 * sp) * G] = v.x;
    o[(size_t)(2 * sp + 1) * G] = v.y;
}

}  // namespace yoho

using namespace yoho;

extern "C" {

int yoho_group_scatter(yoho_ctx* c, const float* feat, int n, const int64_t* idx, int K, int g, float* out, void* stream) {
    if (!c || K < 0 || n < 1 || g < 0 || g >= G) { set_error("yoho_group_scatter: bad argument"); return YOHO_EINVAL; }
    if (K == 0) return 0;
    if (!feat || !idx || !out) { set_error("yoho_group_scatter: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_group_scatter", 15, feat, out);
    YOHO_NEED_ALIGNED("yoho_group_scatter", 7, idx);
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(group_scatter_kernel, dim3((K + 15) / 16), dim3(256), 0, (hipStream_t)stream, feat, idx, K, n, g, out);
    HIPCHK(hipGetLastError());
    return 0;
}

int yoho_hyp_from_quat(yoho_ctx* c, const float* quat, const int64_t* idx, const double* k0, const double* k1, int M, double* T,
                       void* stream) {
    if (!c || !quat || !idx || !k0 || !k1 || !T || M < 0) { set_error("yoho_hyp_from_quat: bad argument"); return YOHO_EINVAL; }
    if (M == 0) return 0;
    YOHO_NEED_ALIGNED("yoho_hyp_from_quat", 15, quat);
    YOHO_NEED_ALIGNED("yoho_hyp_from_quat", 7, idx, k0, k1, T);
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(hyp_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, quat, idx, k0, k1, c->dR64, M, T);
    HIPCHK(hipGetLastError());
    return 0;
}

int yoho_o_score(yoho_ctx* c, const double* k0, const double* k1, int M, const double* T, const int64_t* order, int H, double d,
                 int* best_h, int* best_count, int32_t* counts, void* stream) {
    if (!c || !k0 || !k1 || !T || !best_h || !best_count || M < 1 || H < 1) { set_error("yoho_o_score: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_o_score", 7, k0, k1, T, order);
    YOHO_NEED_ALIGNED("yoho_o_score", 3, best_h, best_count, counts);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    int rc;
    int32_t* cnt = counts;
    if (!cnt) {
        if ((rc = ensure_ws(c, sizeof(int32_t) * (size_t)H, s))) return rc;
        cnt = (int32_t*)c->ws.p;
    }
    hipLaunchKernelGGL(score_kernel, dim3(H), dim3(256), 0, s, k0, k1, M, T, order, d * d, cnt);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(argbest_kernel, dim3(1), dim3(256), 0, s, cnt, H, best_h, best_count);
    HIPCHK(hipGetLastError());
    return 0;
}

int yoho_c_ransac(yoho_ctx* c, const double* k0, const double* k1, int M, const int64_t* triples, const uint8_t* reflect, int I,
                  double d, double* best_T, int* best_iter, int* best_count, double* T_out, int32_t* counts, void* stream) {
    if (!c || !k0 || !k1 || !triples || !best_T || !best_iter || !best_count || M < 1 || I < 1) {
        set_error("yoho_c_ransac: bad argument"); return YOHO_EINVAL;
    }
    YOHO_NEED_ALIGNED("yoho_c_ransac", 7, k0, k1, triples, best_T, T_out);
    YOHO_NEED_ALIGNED("yoho_c_ransac", 3, best_iter, best_count, counts);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // workspace: T (I,12) f64 | counts (I) i32 | best_h i32
    const size_t need = sizeof(double) * 12 * (size_t)I + sizeof(int32_t) * ((size_t)I + 4);
    if ((rc = ensure_ws(c, need, s))) return rc;
    double* Tall = T_out ? T_out : (double*)c->ws.p;
    int32_t* cnt = counts ? counts : (int32_t*)((char*)c->ws.p + sizeof(double) * 12 * (size_t)I);
    int* bh = (int*)((char*)c->ws.p + sizeof(double) * 12 * (size_t)I + sizeof(int32_t) * (size_t)I);
    hipLaunchKernelGGL(kabsch_score_kernel, dim3(I), dim3(256), 0, s, k0, k1, M, triples, reflect, d * d, Tall, cnt);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(argbest_kernel, dim3(1), dim3(256), 0, s, cnt, I, bh, best_count);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(pick_T_kernel, dim3(1), dim3(64), 0, s, Tall, bh, best_count, best_T, best_iter);
    HIPCHK(hipGetLastError());
    return 0;
}

int yoho_c_ransac_device(yoho_ctx* c, const double* keys0, const int64_t* i0, const double* keys1, const int64_t* i1, int istride,
                         const int64_t* dr_index, int M, int max_iter, uint64_t seed, double d, double* best_T, int* best_iter,
                         int* best_count, int64_t* triples_out, void* stream) {
    if (!c || !keys0 || !keys1 || !dr_index || !best_T || !best_iter || !best_count || M < 1 || max_iter < 1 || istride < 1) {
        set_error("yoho_c_ransac_device: bad argument"); return YOHO_EINVAL;
    }
    YOHO_NEED_ALIGNED("yoho_c_ransac_device", 7, keys0, keys1, i0, i1, dr_index, best_T, triples_out);
    YOHO_NEED_ALIGNED("yoho_c_ransac_device", 3, best_iter, best_count);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const size_t I = (size_t)max_iter;
    // workspace: T (I,12) f64 | matched keys 2 x (M,3) f64 | CStat | counts (I) i32 | members (M) i32 | best_h
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t oT = take(sizeof(double) * 12 * I), oK0 = take(sizeof(double) * 3 * (size_t)M), oK1 = take(sizeof(double) * 3 * (size_t)M);
    const size_t oS = take(sizeof(CStat)), oC = take(sizeof(int32_t) * I), oM = take(sizeof(int) * (size_t)M), oB = take(16), oD = take((size_t)M);
    if ((rc = ensure_ws(c, off, s))) return rc;
    char* w = (char*)c->ws.p;
    double* Tall = (double*)(w + oT);
    double *k0m = (double*)(w + oK0), *k1m = (double*)(w + oK1);
    CStat* st = (CStat*)(w + oS);
    int32_t* cnt = (int32_t*)(w + oC);
    int* members = (int*)(w + oM);
    int* bh = (int*)(w + oB);
    unsigned char* dr8 = (unsigned char*)(w + oD);
    hipLaunchKernelGGL(cprep_kernel, dim3((M + 255) / 256), dim3(256), 0, s, dr_index, M, keys0, keys1, i0, i1, istride, dr8, k0m, k1m);
    hipLaunchKernelGGL(cstat_kernel, dim3(1), dim3(64 * CS_WAVES), 0, s, dr8, M, st, members);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(kabsch_sample_kernel, dim3(max_iter), dim3(256), 0, s, k0m, k1m, M, st, members, (unsigned)(seed & 0xFFFFFFFFu),
                       (unsigned)(seed >> 32), d * d, Tall, cnt, triples_out);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(argbest_kernel, dim3(1), dim3(256), 0, s, cnt, max_iter, bh, best_count);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(pick_T_dev_kernel, dim3(1), dim3(64), 0, s, Tall, bh, best_count, st, best_T, best_iter);
    HIPCHK(hipGetLastError());
    return 0;
}

int yoho_group_gather(yoho_ctx* c, const double* keys, int K, const float* pts, const float* feat, int n, int g,
                      const double* Rg_host, float* out, int64_t* nn_idx, void* stream) {
    if (!c || !keys || !pts || !feat || !Rg_host || !out || K < 1 || n < 1 || g < 0 || g >= G) {
        set_error("yoho_group_gather: bad argument"); return YOHO_EINVAL;
    }
    YOHO_NEED_ALIGNED("yoho_group_gather", 7, keys, nn_idx);
    YOHO_NEED_ALIGNED("yoho_group_gather", 15, feat, out);
    YOHO_NEED_ALIGNED("yoho_group_gather", 3, pts);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    Mat3 R;
    for (int i = 0; i < 9; ++i) R.m[i] = Rg_host[i];
    if (c->nn_cell > 0.0 && (size_t)K * n >= (1u << 20)) {
        // hash-grid search (gridnn.hip): same winners as the brute-force slices below
        int rc;
        const size_t head = ((size_t)K * (sizeof(double) + sizeof(int)) + 255) & ~(size_t)255;
        if ((rc = ensure_ws(c, head + grid_nn_ws_bytes(K, n), s))) return rc;
        double* pd = (double*)c->ws.p;
        int* pi = (int*)(pd + K);
        if ((rc = launch_grid_nn(2, keys, K, &R, pts, n, c->nn_cell, (char*)c->ws.p + head, nullptr, nullptr, pd, pi, c->nCU, s))) return rc;
        hipLaunchKernelGGL(gather_merge_kernel, dim3((K + 15) / 16), dim3(256), 0, s, pd, pi, K, 1, feat, g, out, nn_idx);
        HIPCHK(hipGetLastError());
        return 0;
    }
    const int kblocks = (K + 256 * GG_KR - 1) / (256 * GG_KR);
    int nslice = (2048 + kblocks - 1) / kblocks;                       // aim for ~2048 workgroups
    const int max_slices = (n + 255) / 256;                             // at least 256 cloud points per slice
    if (nslice > max_slices) nslice = max_slices;
    if (nslice < 1) nslice = 1;
    const int slice_len = ((n + nslice - 1) / nslice + 255) / 256 * 256;
    nslice = (n + slice_len - 1) / slice_len;
    int rc;
    if ((rc = ensure_ws(c, (size_t)nslice * K * (sizeof(double) + sizeof(int)), s))) return rc;
    double* pd = (double*)c->ws.p;
    int* pi = (int*)(pd + (size_t)nslice * K);
    hipLaunchKernelGGL(gather_nn_kernel, dim3(kblocks, nslice), dim3(256), 0, s, keys, K, pts, n, R, slice_len, pd, pi);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(gather_merge_kernel, dim3((K + 15) / 16), dim3(256), 0, s, pd, pi, K, nslice, feat, g, out, nn_idx);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // extern "C"
