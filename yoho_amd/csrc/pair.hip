// One scene pair in one library call: yoho_register_pair = the stage sequence of the reference's evaluator for a pair whose two
// fragments are already described (tests/evaluator.py:112-117 for YOHO-O, :41-47 for YOHO-C), composed from the entries of
// match.hip / api.hip / estim.hip exactly as yoho_amd/pipeline.py:run_pair composes them from Python:
//
//   mutual NN (tests/matcher.py:35-48) -> match count to the host -> Des2R (tests/extractor.py:97-99)
//     YOHO-O: vote order (tests/estimator.py:321-323: np.random.shuffle of arange(M)) -> PartII (utils/network.py:259-278) on the
//             matches the vote reads (or on all of them) -> [R|t] per match (tests/extractor.py:142-201) -> inlier vote
//             (tests/estimator.py:321-336) -> winner to the host
//     YOHO-C: yoho_c_ransac_device (tests/estimator.py:28-141, sampling on the device) -> winner to the host
//
// Why it exists: from Python a pair is ~25 ctypes calls, a dozen small tensor allocations and two read-backs - 0.6 ms of
// interpreter time per pair, serialised over the dataset driver's worker threads by the GIL, against 0.3-0.6 ms of device time.
// One foreign call releases the GIL for the whole pair, so the workers of run_dataset.ScenePairRunner overlap for real.
//
// The vote order is numpy's: RandomState(seed).shuffle(arange(M)) restated in C (MT19937 seeded by init_genrand, Fisher-Yates
// from the top with masked rejection sampling - numpy/random/mtrand.pyx:_shuffle_raw, _common/distributions.c:random_interval),
// checked against numpy itself in tests/test_host_cpu.py; with it the fused call returns the bits of the Python composition.
#include <cstring>
#include <vector>
#include "common.h"

namespace yoho {

struct NpMt {
    uint32_t key[624];
    int pos;
};

static void np_mt_seed(NpMt& s, uint32_t seed) {
    for (int pos = 0; pos < 624; ++pos) {
        s.key[pos] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
    }
    s.pos = 624;
}

static inline uint32_t np_mt_next(NpMt& s) {
    if (s.pos == 624) {
        constexpr uint32_t A = 0x9908b0dfu, UP = 0x80000000u, LO = 0x7fffffffu;
        uint32_t* mt = s.key;
        int kk = 0;
        for (; kk < 624 - 397; ++kk) {
            const uint32_t y = (mt[kk] & UP) | (mt[kk + 1] & LO);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        for (; kk < 623; ++kk) {
            const uint32_t y = (mt[kk] & UP) | (mt[kk + 1] & LO);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        }
        const uint32_t y = (mt[623] & UP) | (mt[0] & LO);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
        s.pos = 0;
    }
    uint32_t y = s.key[s.pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static void np_shuffle_arange(uint32_t seed, int M, int64_t* out) {
    for (int i = 0; i < M; ++i) out[i] = i;
    NpMt s;
    np_mt_seed(s, seed);
    for (int i = M - 1; i >= 1; --i) {
        uint32_t mask = (uint32_t)i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        uint32_t v;
        while ((v = np_mt_next(s) & mask) > (uint32_t)i) {}
        const int64_t t = out[i]; out[i] = out[v]; out[v] = t;
    }
}

// The sampling half of the reference's YOHO-C loop (tests/estimator.py:113-128) on numpy's legacy stream, restated from
// numpy/random/mtrand.pyx (RandomState.choice, .random_sample, .randint) and _common/distributions.c
// (random_bounded_uint64_fill -> buffered_bounded_masked_uint32):
//   choice(range(60), p = prob)   one 53-bit double u = ((w0 >> 5) * 2^26 + (w1 >> 6)) / 2^53 from two MT words, then
//                                 cdf = cumsum(prob) / cumsum(prob)[-1]; index = searchsorted(cdf, u, side='right')
//   choice(bucket, 3)             randint(0, n, size=3): per element words masked to the smallest 2^k - 1 >= n - 1 until <= n - 1
// A rotation whose bucket holds fewer than two matches consumes its draw and is skipped (the reference's `continue`); the loop
// ends after max_iter accepted iterations or 50001 draws (:118 `if exec_time > max_time: break` is tested before the increment).
static void np_yohoc_draws(NpMt& s, const double* prob, const int64_t* start, const int64_t* members, int max_iter, int64_t* triples,
                           int* n_triples, int* n_draws) {
    double cdf[60];
    double acc = 0.0;
    for (int i = 0; i < 60; ++i) { acc = i ? acc + prob[i] : prob[0]; cdf[i] = acc; }      // ndarray.cumsum: sequential adds
    const double last = cdf[59];
    for (int i = 0; i < 60; ++i) cdf[i] = cdf[i] / last;
    int it = 0, draws = 0;
    while (it < max_iter) {
        if (draws > 50000) break;
        ++draws;
        const uint32_t a = np_mt_next(s) >> 5, b = np_mt_next(s) >> 6;
        const double u = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
        int lo = 0, hi = 60;                                  // searchsorted(side='right'): the number of cdf entries <= u
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
        const int rot = lo;
        // numpy would raise IndexError for rot == 60 (u >= cdf[59] = 1 cannot happen: u < 1); an empty answer keeps the ABI total
        const int64_t n = rot < 60 ? start[rot + 1] - start[rot] : 0;
        if (n < 2) continue;
        const uint32_t rng = (uint32_t)(n - 1);
        uint32_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
        for (int e = 0; e < 3; ++e) {
            uint32_t v;
            while ((v = np_mt_next(s) & mask) > rng) {}
            triples[3 * it + e] = members[start[rot] + v];
        }
        ++it;
    }
    *n_triples = it;
    *n_draws = draws;
}

// k0m[m] = keys0[pairs[m][0]], k1m[m] = keys1[pairs[m][1]]
__global__ void pair_keys_kernel(const double* __restrict__ keys0, const double* __restrict__ keys1, const int64_t* __restrict__ pairs, int M,
                                 double* __restrict__ k0m, double* __restrict__ k1m) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 3 * M) return;
    const int m = i / 3, a = i - 3 * m;
    k0m[i] = keys0[pairs[2 * m] * 3 + a];
    k1m[i] = keys1[pairs[2 * m + 1] * 3 + a];
}

// the rows the vote reads: row h of every selected array = row order[h] of the full one
__global__ void pair_select_kernel(const int64_t* __restrict__ order, int H, const int64_t* __restrict__ pairs, const int64_t* __restrict__ dr,
                                   const double* __restrict__ k0m, const double* __restrict__ k1m, int64_t* __restrict__ ms,
                                   int64_t* __restrict__ drs, double* __restrict__ k0s, double* __restrict__ k1s) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    const int64_t m = order[h];
    ms[2 * h] = pairs[2 * m]; ms[2 * h + 1] = pairs[2 * m + 1];
    drs[h] = dr[m];
#pragma unroll
    for (int a = 0; a < 3; ++a) { k0s[3 * h + a] = k0m[3 * m + a]; k1s[3 * h + a] = k1m[3 * m + a]; }
}

struct PairDev {            // what the host reads at the end of a pair
    double trans[12];
    int best_h, best_count, range, pad;
};

// YOHO-O: the winner's [R|t] (row res[0] of T, through the vote order unless T is already in vote order) + the PartII range word
__global__ void pair_pick_kernel(const double* __restrict__ T, const int* __restrict__ res, const int64_t* __restrict__ order, int* rflag,
                                 PairDev* __restrict__ out) {
    const int t = threadIdx.x;
    const int bh = res[0];
    const int64_t row = order ? order[bh] : bh;
    if (t < 12) out->trans[t] = T[row * 12 + t];
    if (t == 12) { out->best_h = bh; out->best_count = res[1]; out->range = rflag ? atomicExch(rflag + 1, 0) : 0; out->pad = 0; }
}

// YOHO-C: best_T / best_iter / best_count as yoho_c_ransac_device leaves them
__global__ void pair_pick_c_kernel(const double* __restrict__ bestT, const int* __restrict__ res, PairDev* __restrict__ out) {
    const int t = threadIdx.x;
    if (t < 12) out->trans[t] = bestT[t];
    if (t == 12) { out->best_h = res[0]; out->best_count = res[1]; out->range = 0; out->pad = 0; }
}

struct PairScratch {
    int64_t *pairs, *dr, *order, *ms, *drs;
    double *k0m, *k1m, *k0s, *k1s, *T, *bestT;
    float* quat;
    int* res;               // [0] best_h / best_iter, [1] best_count, [2] M
    PairDev* out;
};

static int pair_scratch(yoho_ctx* c, int n, PairScratch& p, hipStream_t s) {
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t N = (size_t)n;
    const size_t o_pairs = take(16 * N), o_dr = take(8 * N), o_order = take(8 * N), o_ms = take(16 * N), o_drs = take(8 * N);
    const size_t o_k0m = take(24 * N), o_k1m = take(24 * N), o_k0s = take(24 * N), o_k1s = take(24 * N), o_T = take(96 * N), o_bT = take(96);
    const size_t o_q = take(16 * N), o_res = take(16), o_out = take(sizeof(PairDev));
    if (c->pair_ws_bytes < off) {
        if (c->pair_ws) {
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipFree(c->pair_ws));
            c->pair_ws = nullptr; c->pair_ws_bytes = 0;
        }
        hipError_t e = hipMalloc(&c->pair_ws, off);
        if (e != hipSuccess) { set_error("pair scratch of %zu bytes: %s", off, hipGetErrorString(e)); return YOHO_ENOMEM; }
        c->pair_ws_bytes = off;
    }
    const size_t host_need = 8 * N + 256;
    if (c->pair_host_bytes < host_need) {
        if (c->pair_host) { HIPCHK(hipStreamSynchronize(s)); HIPCHK(hipHostFree(c->pair_host)); c->pair_host = nullptr; c->pair_host_bytes = 0; }
        hipError_t e = hipHostMalloc(&c->pair_host, host_need, hipHostMallocDefault);
        if (e != hipSuccess) { set_error("pinned pair scratch of %zu bytes: %s", host_need, hipGetErrorString(e)); return YOHO_ENOMEM; }
        c->pair_host_bytes = host_need;
    }
    char* w = (char*)c->pair_ws;
    p.pairs = (int64_t*)(w + o_pairs); p.dr = (int64_t*)(w + o_dr); p.order = (int64_t*)(w + o_order); p.ms = (int64_t*)(w + o_ms);
    p.drs = (int64_t*)(w + o_drs); p.k0m = (double*)(w + o_k0m); p.k1m = (double*)(w + o_k1m); p.k0s = (double*)(w + o_k0s);
    p.k1s = (double*)(w + o_k1s); p.T = (double*)(w + o_T); p.bestT = (double*)(w + o_bT); p.quat = (float*)(w + o_q);
    p.res = (int*)(w + o_res); p.out = (PairDev*)(w + o_out);
    return 0;
}

}  // namespace yoho

using namespace yoho;

extern "C" {

int yoho_vote_order(uint32_t seed, int M, int64_t* order) {
    if (M < 0 || (M > 0 && !order)) { set_error("yoho_vote_order: bad argument"); return YOHO_EINVAL; }
    np_shuffle_arange(seed, M, order);
    return 0;
}

int yoho_c_draw_np(uint32_t* mt_key, int* mt_pos, const double* prob, const int64_t* bucket_start, const int64_t* bucket_members,
                   int max_iter, int64_t* triples, int* n_triples, int* n_draws) {
    if (!mt_key || !mt_pos || !prob || !bucket_start || !bucket_members || !triples || !n_triples || !n_draws || max_iter < 0 ||
        *mt_pos < 0 || *mt_pos > 624) {
        set_error("yoho_c_draw_np: bad argument"); return YOHO_EINVAL;
    }
    for (int i = 0; i < 60; ++i)
        if (bucket_start[i + 1] < bucket_start[i] || bucket_start[i] < 0 || !(prob[i] >= 0.0)) {
            set_error("yoho_c_draw_np: bucket_start must be non-decreasing and prob non-negative (rotation %d)", i); return YOHO_EINVAL;
        }
    if (!(prob[59] >= 0.0)) { set_error("yoho_c_draw_np: prob must be non-negative"); return YOHO_EINVAL; }
    NpMt s;
    std::memcpy(s.key, mt_key, sizeof(s.key));
    s.pos = *mt_pos;
    np_yohoc_draws(s, prob, bucket_start, bucket_members, max_iter, triples, n_triples, n_draws);
    std::memcpy(mt_key, s.key, sizeof(s.key));
    *mt_pos = s.pos;
    return 0;
}

int yoho_register_pair(yoho_ctx* c, const float* feat0, const float* feat1, const float* eqv0, const float* eqv1, const float* inv0,
                       const float* inv1, const double* keys0, const double* keys1, int n0, int n1, int estimator, int max_iter,
                       double inlier_dist, uint64_t seed, int selected, yoho_pair_result* out, void* stream) {
    if (!c || !eqv0 || !eqv1 || !inv0 || !inv1 || !keys0 || !keys1 || !out || n0 < 1 || n1 < 1 || max_iter < 1 ||
        (estimator != YOHO_ESTIMATOR_O && estimator != YOHO_ESTIMATOR_C) || (estimator == YOHO_ESTIMATOR_O && (!feat0 || !feat1))) {
        set_error("yoho_register_pair: bad argument"); return YOHO_EINVAL;
    }
    if (estimator == YOHO_ESTIMATOR_O && !c->has_partII) { set_error("yoho_register_pair: PartII weights not loaded"); return YOHO_ENOWEIGHTS; }
    if (estimator == YOHO_ESTIMATOR_O && c->partII_mode < 2) {
        set_error("yoho_register_pair: YOHO-O runs in the fp16x2 PartII arithmetic modes (2, 3, 4) only (the caller composes the staged entries otherwise)");
        return YOHO_EINVAL;
    }
    YOHO_NEED_ALIGNED("yoho_register_pair", 15, feat0, feat1, eqv0, eqv1, inv0, inv1);
    YOHO_NEED_ALIGNED("yoho_register_pair", 7, keys0, keys1);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    std::memset(out, 0, sizeof(*out));
    PairScratch p;
    int rc;
    if ((rc = pair_scratch(c, n0, p, s))) return rc;
    int* hostM = (int*)c->pair_host;                                  // [0] M; the result struct follows at +64, the order at +256
    PairDev* hostR = (PairDev*)((char*)c->pair_host + 64);
    int64_t* hostOrder = (int64_t*)((char*)c->pair_host + 256);

    if ((rc = yoho_mutual_nn(c, inv0, n0, inv1, n1, p.pairs, p.res + 2, s))) return rc;
    HIPCHK(hipMemcpyAsync(hostM, p.res + 2, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));                                  // first host wait of the pair: the match count
    const int M = *hostM;
    out->matches = M;
    if (M == 0) return 0;
    // Batch_Des2R_torch(feats1, feats0): rows addressed in place through the match list
    if ((rc = yoho_des2r_indexed(c, eqv1, p.pairs + 1, eqv0, p.pairs, 2, M, p.dr, nullptr, s))) return rc;

    if (estimator == YOHO_ESTIMATOR_C) {
        if ((rc = yoho_c_ransac_device(c, keys0, p.pairs, keys1, p.pairs + 1, 2, p.dr, M, max_iter, seed, inlier_dist, p.bestT, p.res, p.res + 1,
                                       nullptr, s))) return rc;
        hipLaunchKernelGGL(pair_pick_c_kernel, dim3(1), dim3(64), 0, s, p.bestT, p.res, p.out);
        HIPCHK(hipGetLastError());
        out->hypotheses = max_iter;
    } else {
        const int H = max_iter < M ? max_iter : M;
        out->hypotheses = H;
        np_shuffle_arange((uint32_t)(seed & 0xFFFFFFFFu), M, hostOrder);
        HIPCHK(hipMemcpyAsync(p.order, hostOrder, sizeof(int64_t) * (size_t)M, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(pair_keys_kernel, dim3((3 * M + 255) / 256), dim3(256), 0, s, keys0, keys1, p.pairs, M, p.k0m, p.k1m);
        HIPCHK(hipGetLastError());
        const bool sel = selected && H < M;
        if (sel) {
            hipLaunchKernelGGL(pair_select_kernel, dim3((H + 255) / 256), dim3(256), 0, s, p.order, H, p.pairs, p.dr, p.k0m, p.k1m, p.ms, p.drs,
                               p.k0s, p.k1s);
            HIPCHK(hipGetLastError());
        }
        const int64_t* mlist = sel ? p.ms : p.pairs;
        const int64_t* dr = sel ? p.drs : p.dr;
        const int nh = sel ? H : M;
        // batch_create's 0 <-> 1 exchange (tests/extractor.py:125-138): the network sees fragment 1 first
        if ((rc = yoho_partII_forward_indexed(c, feat1, mlist + 1, feat0, mlist, eqv1, mlist + 1, eqv0, mlist, 2, dr, nh, p.quat, s))) return rc;
        if ((rc = yoho_hyp_from_quat(c, p.quat, dr, sel ? p.k0s : p.k0m, sel ? p.k1s : p.k1m, nh, p.T, s))) return rc;
        if ((rc = yoho_o_score(c, p.k0m, p.k1m, M, p.T, sel ? nullptr : p.order, H, inlier_dist, p.res, p.res + 1, nullptr, s))) return rc;
        hipLaunchKernelGGL(pair_pick_kernel, dim3(1), dim3(64), 0, s, p.T, p.res, sel ? nullptr : p.order, c->d_rflag, p.out);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(hostR, p.out, sizeof(PairDev), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));                                  // second host wait: the winner
    std::memcpy(out->trans, hostR->trans, sizeof(out->trans));
    out->best_h = hostR->best_h;
    out->best_count = hostR->best_count;
    out->range_flag = hostR->range;
    return 0;
}

}  // extern "C"
