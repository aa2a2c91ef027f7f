// Group-Fourier formulation of the icosahedral group convolution (fp32 MFMA), gfx950.
//
// The reference's 13-tap conv (utils/network.py:46-52,12-21) is a cross-correlation on the rotation group:
// N[g,k] = idx(R_{N[0,k]} R_g) = n_k * g, so   y(g) = sum_k W_k x(n_k g).
// The icosahedral rotation group (A5, order 60) has real irreps rho of dimension 1, 3, 3', 4, 5
// (1 + 9 + 9 + 16 + 25 = 60).  With  x^(rho) = sum_g x(g) rho(g)  (a d x d matrix per channel):
//     y^(rho) = [sum_k W_k rho(n_k)^T] x^(rho)
// i.e. in the orthonormal coefficient basis  F[(rho,i,j), g] = sqrt(d/60) rho(g)[i,j]  every output
// coefficient slab (rho,i,j) needs only d input slabs (rho,m,j), m < d:
//     Y[(rho,i,j)] = sum_{m<d} What(rho,i,m) (Cout x Cin) * X[(rho,m,j)]
// -> sum_rho d^3 = 244 slab products per 8-channel chunk instead of 60 * 13 = 780 (3.2x fewer MFMAs),
// same slab layout and same MFMA tile as gconv.hip, no gather at all.  BN+ReLU act in the group domain,
// so layers are separated by a transform kernel  X^ -> F^T -> (+bias, BN, ReLU) -> F -> X^  (two 60x60
// matmuls per (keypoint, channel), also on MFMA).  Bias and the residual add are linear and stay in the
// Fourier domain (a constant b is sqrt(60) b on the trivial-irrep coefficient).
//
// The irreps are not hard-coded: they are computed at yoho_ctx_create from the 60_60 multiplication
// table (eigen-spaces of a random symmetric element of the right-regular commutant), so any consistent
// set of group tables works.  Results equal the direct formulation up to fp32 rounding (orthogonal
// transforms); parity with the oracle is asserted in tests/test_gpu_kernels.py.
#include "common.h"
#include <cmath>
#include <cstring>
#include <type_traits>
#include <vector>
#include <algorithm>

namespace yoho {

// ---------------------------------------------------------------------------------------------------
// compile-time structure of the coefficient basis and the per-wave MFMA programs
// ---------------------------------------------------------------------------------------------------
constexpr int NIR = 5;
constexpr int IR_D[NIR] = {1, 3, 3, 4, 5};
constexpr int IR_BASE[NIR] = {0, 1, 10, 19, 35};          // first coefficient index of irrep r; coeff = base + i*d + j

struct FStep { short frag, in_slab, acc, newfrag; };
struct FProg {
    int nsteps[4];
    FStep step[4][64];
    int nout[4];
    short out_slab[4][16];
};

constexpr FProg build_fprog() {
    FProg p{};
    // ownership of output slabs: cost of slab (r,i,j) is d_r.  Balanced hand assignment (61 per wave):
    //   wave 0: 7 slabs of rho5, rho4 row 0, rho3 row 0, rho1;  waves 1..3: 6 of rho5, a rho4 row, 5 of the threes
    short owner[60] = {};
    // rho5 (base 35, 25 slabs, row-major): 7 / 6 / 6 / 6
    {
        const int cnt[4] = {7, 6, 6, 6};
        int s = 35;
        for (int w = 0; w < 4; ++w) for (int k = 0; k < cnt[w]; ++k) owner[s++] = (short)w;
    }
    // rho4 (base 19): one row per wave
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) owner[19 + i * 4 + j] = (short)i;
    // the 18 three-dimensional slabs (rho3 base 1, rho3' base 10) in order: 3 / 5 / 5 / 5
    {
        const int cnt[4] = {3, 5, 5, 5};
        int s = 1;
        for (int w = 0; w < 4; ++w) for (int k = 0; k < cnt[w]; ++k) owner[s++] = (short)w;
    }
    owner[0] = 0;
    for (int w = 0; w < 4; ++w) {
        int ns = 0, no = 0;
        short accof[60] = {};
        for (int s = 0; s < 60; ++s) accof[s] = -1;
        for (int s = 0; s < 60; ++s)
            if (owner[s] == w) { accof[s] = (short)no; p.out_slab[w][no++] = (short)s; }
        p.nout[w] = no;
        for (int r = 0; r < NIR; ++r) {
            const int d = IR_D[r], base = IR_BASE[r];
            for (int i = 0; i < d; ++i) {
                bool any = false;
                for (int j = 0; j < d; ++j) any = any || owner[base + i * d + j] == w;
                if (!any) continue;
                for (int m = 0; m < d; ++m) {
                    bool first = true;
                    for (int j = 0; j < d; ++j) {
                        const int so = base + i * d + j;
                        if (owner[so] != w) continue;
                        p.step[w][ns].frag = (short)(base + i * d + m);          // What(r,i,m)
                        p.step[w][ns].in_slab = (short)(base + m * d + j);        // X[(r,m,j)]
                        p.step[w][ns].acc = accof[so];
                        p.step[w][ns].newfrag = first ? 1 : 0;
                        first = false;
                        ++ns;
                    }
                }
            }
        }
        p.nsteps[w] = ns;
    }
    return p;
}

constexpr FProg FPROG = build_fprog();
static_assert(FPROG.nsteps[0] == 61 && FPROG.nsteps[1] == 61 && FPROG.nsteps[2] == 61 && FPROG.nsteps[3] == 61, "unbalanced program");
static_assert(FPROG.nout[0] == 15 && FPROG.nout[1] == 15 && FPROG.nout[2] == 15 && FPROG.nout[3] == 15, "15 output slabs per wave");

// ---------------------------------------------------------------------------------------------------
// host: irreps from the multiplication table
// ---------------------------------------------------------------------------------------------------
static void jacobi_eigh(std::vector<double>& A, int n, std::vector<double>& V, std::vector<double>& w) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[(size_t)p * n + q] * A[(size_t)p * n + q];
        if (off < 1e-26) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq;
                    A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk;
                    A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq;
                    V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = A[(size_t)i * n + i];
}

// rho[r][g][a*d+b]; returns 0 or YOHO_EINVAL when the tables are not the icosahedral group
int build_fourier(const uint8_t* N, const uint8_t* P, FourierBasis& fb) {
    auto mul = [&](int a, int b) { return (int)P[b * G + a]; };          // element of R_a R_b  (P[i,g] = idx(R_g R_i))
    int inv[G];
    for (int g = 0; g < G; ++g) {
        inv[g] = -1;
        for (int h = 0; h < G; ++h) if (mul(g, h) == 0) inv[g] = h;
        if (inv[g] < 0) { set_error("60_60 table: element %d has no inverse (element 0 must be the identity)", g); return YOHO_EINVAL; }
    }
    for (int g = 0; g < G; ++g)
        for (int k = 0; k < NTAP; ++k)
            if ((int)N[g * NTAP + k] != mul((int)N[k], g)) { set_error("Nei table is not left multiplication by N[0,k]"); return YOHO_EINVAL; }
    // random symmetric element of the commutant of the left-regular representation
    std::vector<double> S((size_t)G * G, 0.0);
    unsigned long long lcg = 0x9E3779B97F4A7C15ull;
    for (int h = 0; h < G; ++h) {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        const double c = ((double)(lcg >> 11) / 9007199254740992.0) - 0.5;
        for (int g = 0; g < G; ++g) {
            const int r = mul(g, inv[h]);          // right multiplication by h^-1
            S[(size_t)r * G + g] += c;
            S[(size_t)g * G + r] += c;
        }
    }
    std::vector<double> V, w;
    jacobi_eigh(S, G, V, w);
    std::vector<int> ord(G);
    for (int i = 0; i < G; ++i) ord[i] = i;
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return w[a] < w[b]; });
    bool have[NIR] = {false, false, false, false, false};
    const int h72 = N[1];                                                  // a 72-degree rotation (N[0,1])
    int i0 = 0, ngroups = 0;
    while (i0 < G) {
        int i1 = i0 + 1;
        while (i1 < G && std::fabs(w[ord[i1]] - w[ord[i0]]) < 1e-7) ++i1;
        const int d = i1 - i0;
        ++ngroups;
        if (d != 1 && d != 3 && d != 4 && d != 5) { set_error("group tables: unexpected invariant subspace of dimension %d", d); return YOHO_EINVAL; }
        // rho(h)[a][b] = sum_g B[h g][a] B[g][b]
        std::vector<double> rho((size_t)G * d * d, 0.0);
        for (int h = 0; h < G; ++h)
            for (int a = 0; a < d; ++a)
                for (int b = 0; b < d; ++b) {
                    double s = 0.0;
                    for (int g = 0; g < G; ++g) s += V[(size_t)mul(h, g) * G + ord[i0 + a]] * V[(size_t)g * G + ord[i0 + b]];
                    rho[((size_t)h * d + a) * d + b] = s;
                }
        int r = -1;
        if (d == 1) r = 0;
        else if (d == 4) r = 3;
        else if (d == 5) r = 4;
        else {
            double ch = 0.0;
            for (int a = 0; a < 3; ++a) ch += rho[((size_t)h72 * 3 + a) * 3 + a];
            r = ch > 0.5 ? 1 : 2;                                          // character (1+sqrt5)/2 vs (1-sqrt5)/2
        }
        if (!have[r]) {
            have[r] = true;
            for (int h = 0; h < G; ++h)
                for (int k = 0; k < d * d; ++k) fb.rho[r][h][k] = rho[(size_t)h * d * d + k];
        }
        i0 = i1;
    }
    for (int r = 0; r < NIR; ++r) if (!have[r]) { set_error("group tables: irrep %d of the icosahedral group not found", r); return YOHO_EINVAL; }
    if (ngroups != 16) { set_error("group tables: %d invariant subspaces instead of 16", ngroups); return YOHO_EINVAL; }
    // checks: homomorphism and orthogonality of F
    for (int r = 0; r < NIR; ++r) {
        const int d = IR_D[r];
        for (int a = 0; a < d; ++a)
            for (int b = 0; b < d; ++b) {
                double s = 0.0;
                for (int k = 0; k < d; ++k) s += fb.rho[r][7][a * d + k] * fb.rho[r][13][k * d + b];
                if (std::fabs(s - fb.rho[r][mul(7, 13)][a * d + b]) > 1e-9) { set_error("irrep %d is not a homomorphism", r); return YOHO_EINVAL; }
            }
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j)
                for (int g = 0; g < G; ++g) fb.F[(IR_BASE[r] + i * d + j) * G + g] = std::sqrt((double)d / G) * fb.rho[r][g][i * d + j];
    }
    for (int a = 0; a < G; ++a)
        for (int b = 0; b < G; ++b) {
            double s = 0.0;
            for (int g = 0; g < G; ++g) s += fb.F[a * G + g] * fb.F[b * G + g];
            if (std::fabs(s - (a == b ? 1.0 : 0.0)) > 1e-9) { set_error("Fourier basis is not orthonormal"); return YOHO_EINVAL; }
        }
    for (int k = 0; k < NTAP; ++k) fb.n0[k] = N[k];
    return 0;
}

// What(r,i,m)[o][c] = sum_k W[o][c][k] rho_r(n_k)[m][i], packed as MFMA A fragments
//   Wpf[ob][c8][frag 60][lane = 32h + i32][s]  = What(frag)[ob*32 + i32][c8*8 + 4h + s]
void pack_fourier_weights(const FourierBasis& fb, const float* W, int cin, int cout, int cout_pad, std::vector<float>& out) {
    const int nob = cout_pad / 32, c8n = cin / 8;
    out.assign((size_t)nob * c8n * 60 * 256, 0.f);
    std::vector<double> coef(60 * NTAP);                                  // coef[frag][k] = rho(n_k)[m][i]
    for (int r = 0; r < NIR; ++r) {
        const int d = IR_D[r];
        for (int i = 0; i < d; ++i)
            for (int m = 0; m < d; ++m)
                for (int k = 0; k < NTAP; ++k) coef[(IR_BASE[r] + i * d + m) * NTAP + k] = fb.rho[r][fb.n0[k]][m * d + i];
    }
    for (int ob = 0; ob < nob; ++ob)
        for (int c8 = 0; c8 < c8n; ++c8)
            for (int f = 0; f < 60; ++f) {
                float* dst = &out[(((size_t)ob * c8n + c8) * 60 + f) * 256];
                for (int lane = 0; lane < 64; ++lane) {
                    const int h = lane >> 5, o = ob * 32 + (lane & 31);
                    if (o >= cout) continue;
                    for (int s = 0; s < 4; ++s) {
                        const int c = c8 * 8 + 4 * h + s;
                        const float* wk = W + ((size_t)o * cin + c) * NTAP;
                        double acc = 0.0;
                        for (int k = 0; k < NTAP; ++k) acc += (double)wk[k] * coef[f * NTAP + k];
                        dst[lane * 4 + s] = (float)acc;
                    }
                }
            }
}

// ---------------------------------------------------------------------------------------------------
// Fourier-domain group conv kernel (fp32 MFMA).  Same tile / slab / staging scheme as gconv_kernel.
// ---------------------------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int CHUNK_BYTES_F = CHUNK_FLOATS * 4;           // 61440
constexpr int LDS_BYTES_F = 2 * CHUNK_BYTES_F;            // 122880
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int I, int N, typename Fn>
__device__ __forceinline__ void static_for_f(Fn&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_f<I + 1, N>(f);
    }
}

__device__ __forceinline__ void stage_chunk_f(const float* src, char* dst, int w, int lane) {
    for (int p = w; p < G; p += 4) {
        const float* s = src + p * SLAB_FLOATS + lane * 4;
        char* d = dst + p * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)s, (lptr_t)d, 16, 0, 0);
    }
}

struct ConvFArgs {
    const float* X;        // Fourier coefficients of the activated input [tile][cin8][60][256]
    const float* Wp;       // [ob][cin8][60 frags][64][4]
    const float* bias;     // [cout_pad]
    const float* res;      // raw Fourier residual (layout of the output), EPI_RES
    float* out;            // raw Fourier output
    int nTiles, cin8, cout8, nOB;
};

template <int W>
__device__ __forceinline__ void gconvf_wave(const ConvFArgs& a, int flags, char* smem, int lane, int tile, int ob) {
    constexpr int NS = FPROG.nsteps[W];
    constexpr int NO = FPROG.nout[W];
    floatx16 acc[NO];
#pragma unroll
    for (int j = 0; j < NO; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    const float* Xt = a.X + (size_t)tile * a.cin8 * CHUNK_FLOATS;
    const floatx4* Wb = reinterpret_cast<const floatx4*>(a.Wp) + (size_t)ob * a.cin8 * 60 * 64 + lane;

    // This wave's weight fragments in program order.  A fragment feeds only d <= 5 slab products (~1300 cycles of
    // MFMA), far less than an L2-miss round trip, so NPF fragments are kept in flight in a register ring.
    // Slot of the q-th fragment of a chunk is q % NPF; for that to hold across chunk boundaries the sequence is
    // padded (virtually) to NFRP, a multiple of NPF - padded positions are never loaded.
    constexpr int NPF = 6;
    constexpr int NFR = []() constexpr { int n = 0; for (int k = 0; k < NS; ++k) n += FPROG.step[W][k].newfrag; return n; }();
    constexpr int NFRP = (NFR + NPF - 1) / NPF * NPF;
    // frag_at(q): fragment id of the q-th distinct fragment of the program
    auto frag_ptr = [&](int c8, auto qc) -> const floatx4* {
        constexpr int q = decltype(qc)::value;
        constexpr int f = []() constexpr { int n = -1; for (int k = 0; k < NS; ++k) { n += FPROG.step[W][k].newfrag; if (n == q) return (int)FPROG.step[W][k].frag; } return 0; }();
        return Wb + ((size_t)c8 * 60 + f) * 64;
    };
    floatx4 wq[NPF];

    // Activation chunks are staged through registers (global_load -> ds_write) rather than by LDS DMA: next to an
    // in-flight global_load_lds hipcc drains the whole vector-memory queue (vmcnt(0)) at every use of an ordinary
    // load, which would defeat the weight ring.  A wave moves pieces W, W+4, .. (15 x 1 KiB) in three batches of
    // five, issued at fixed steps of the MFMA program and written to the other LDS buffer 18 steps later.
    floatx4 xr[5];
    auto xload = [&](const float* chunk, int batch) {
#pragma unroll
        for (int k = 0; k < 5; ++k) xr[k] = *reinterpret_cast<const floatx4*>(chunk + (W + 4 * (batch * 5 + k)) * SLAB_FLOATS + lane * 4);
    };
    auto xstore = [&](char* buf, int batch) {
#pragma unroll
        for (int k = 0; k < 5; ++k) *reinterpret_cast<floatx4*>(buf + (W + 4 * (batch * 5 + k)) * 1024 + lane * 16) = xr[k];
    };
#pragma unroll
    for (int bt = 0; bt < 3; ++bt) { xload(Xt, bt); xstore(smem, bt); }
    static_for_f<0, NPF>([&](auto qc) { wq[decltype(qc)::value] = *frag_ptr(0, qc); });
    __syncthreads();

    for (int c8 = 0; c8 < a.cin8; ++c8) {
        const bool last = c8 + 1 == a.cin8;
        const float* xnext = Xt + (size_t)(c8 + 1) * CHUNK_FLOATS;
        char* bnext = smem + ((c8 + 1) & 1) * CHUNK_BYTES_F;
        const char* xb = smem + (c8 & 1) * CHUNK_BYTES_F + lane * 16;
        floatx4 wc = wq[0];
        static_for_f<0, NS>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            constexpr FStep st = FPROG.step[W][I];
            // sched_barrier: hipcc would otherwise sink each batch of loads down to its ds_write (register pressure
            // heuristic) and expose the full global-memory latency three times per chunk
            if constexpr (I == 0) { if (!last) xload(xnext, 0); __builtin_amdgcn_sched_barrier(0); }
            if constexpr (I == 18) { if (!last) { xstore(bnext, 0); xload(xnext, 1); } __builtin_amdgcn_sched_barrier(0); }
            if constexpr (I == 38) { if (!last) { xstore(bnext, 1); xload(xnext, 2); } __builtin_amdgcn_sched_barrier(0); }
            if constexpr (I == 58) { if (!last) xstore(bnext, 2); }
            if constexpr (st.newfrag) {
                // q = index of this fragment among the wave's distinct fragments
                constexpr int q = []() constexpr { int n = -1; for (int k = 0; k <= I; ++k) n += FPROG.step[W][k].newfrag; return n; }();
                wc = wq[q % NPF];
                // refill the slot with its next real occupant: NPF positions ahead, skipping padded positions,
                // wrapping into the next chunk
                constexpr int v = []() constexpr { int x = q + NPF; while (x >= NFR && x < NFRP) x += NPF; return x; }();
                if constexpr (v < NFR) wq[q % NPF] = *frag_ptr(c8, std::integral_constant<int, v>{});
                else if (!last) wq[q % NPF] = *frag_ptr(c8 + 1, std::integral_constant<int, v - NFRP>{});
            }
            const floatx4 xf = *reinterpret_cast<const floatx4*>(xb + st.in_slab * 1024);
            acc[st.acc] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.x, xf.x, acc[st.acc], 0, 0, 0);
            acc[st.acc] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.y, xf.y, acc[st.acc], 0, 0, 0);
            acc[st.acc] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.z, xf.z, acc[st.acc], 0, 0, 0);
            acc[st.acc] = __builtin_amdgcn_mfma_f32_32x32x2f32(wc.w, xf.w, acc[st.acc], 0, 0, 0);
        });
        __syncthreads();
    }

    // epilogue: raw Fourier coefficients (+ bias on the trivial-irrep coefficient, + residual)
    const int kp = lane & 31, half = lane >> 5;
    static_for_f<0, NO>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr int so = FPROG.out_slab[W][j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = ob * 32 + q * 8 + half * 4;
            floatx4 val;
            val.x = acc[j][4 * q + 0]; val.y = acc[j][4 * q + 1];
            val.z = acc[j][4 * q + 2]; val.w = acc[j][4 * q + 3];
            if constexpr (so == 0) val += *reinterpret_cast<const floatx4*>(a.bias + ch) * 7.745966692414834f;   // sqrt(60)
            const size_t off = (((((size_t)tile * a.cout8 + ob * 4 + q) * G + so) * 2 + half) * TILE + kp) * 4;
            if (flags & EPI_RES) val += *reinterpret_cast<const floatx4*>(a.res + off);
            *reinterpret_cast<floatx4*>(a.out + off) = val;
        }
    });
}

__global__ __launch_bounds__(256, 1) void gconvf_kernel(ConvFArgs a, int flags) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7, slot8 = b >> 3;
    const int v = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot8;
    // The 32 workgroups an XCD runs concurrently cover 4 tiles x 8 o-blocks: both the activation stream (shared by
    // the o-blocks of a tile) and the weight stream (shared by the tiles of an o-block) are then reused in L2,
    // instead of 2 tiles x 16 o-blocks where every weight block has only two readers.
    const int obg = a.nOB < 8 ? a.nOB : 8;                 // o-blocks per group
    const int ngrp = a.nOB / obg;
    const int per_quad = 4 * a.nOB;                        // work items per 4 tiles
    const int tq = v / per_quad, rq = v - tq * per_quad;
    const int og = rq / (4 * obg), r2 = rq - og * (4 * obg);
    const int tile = tq * 4 + r2 / obg;
    const int ob = og * obg + (r2 % obg);
    (void)ngrp;
    if (tile >= a.nTiles) return;
    if (w == 0) gconvf_wave<0>(a, flags, smem, lane, tile, ob);
    else if (w == 1) gconvf_wave<1>(a, flags, smem, lane, tile, ob);
    else if (w == 2) gconvf_wave<2>(a, flags, smem, lane, tile, ob);
    else gconvf_wave<3>(a, flags, smem, lane, tile, ob);
}

int launch_gconvf(const Layer& L, const float* X, int nTiles, const float* res, float* out, int flags, hipStream_t s) {
    ConvFArgs a;
    a.X = X; a.Wp = L.wpf; a.bias = L.bias; a.res = res; a.out = out;
    a.nTiles = nTiles; a.cin8 = L.cin / 8; a.cout8 = L.cout_pad / 8; a.nOB = L.cout_pad / 32;
    hipLaunchKernelGGL(gconvf_kernel, dim3((nTiles + 3) / 4 * 4 * a.nOB), dim3(256), LDS_BYTES_F, s, a, flags);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// transform kernel: one workgroup per (tile, c8) chunk = 60 coefficient slabs x 256 columns.
//   GFT_FWD : out = F * in                                   (group domain -> Fourier)
//   GFT_INV : out = F^T * in                                 (Fourier -> group domain)
//   GFT_ACT : out = F * relu(s * (F^T * in) + t)             (the BN+ReLU between two convs)
// Both 60x60 matmuls run on v_mfma_f32_32x32x2_f32 with the transform matrix as the A operand (held in
// registers, padded to 64x64 with zeros) and the chunk, staged in LDS as [row 64][256 cols], as B.
// ---------------------------------------------------------------------------------------------------
enum { GFT_FWD = 0, GFT_INV = 1, GFT_ACT = 2 };

constexpr int GFT_ROWS = 64;
constexpr int GFT_LDS = GFT_ROWS * 256 * 4;              // 65536: the chunk is transformed in place, 2 workgroups per CU

// D(64 x 64 cols of this wave) = T(64x64) * B(64 x 256)[:, cols];  afrag[rb][t] = T[rb*32 + (lane&31)][2t + (lane>>5)]
__device__ __forceinline__ void gft_mm(const float (&afrag)[2][32], const float* lds, int lane, int cb0, floatx16 (&acc)[2][2]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[rb][c][i] = 0.f;
    const float* bp = lds + (lane >> 5) * 256 + cb0 * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < 32; ++t) {      // fully unrolled: afrag must stay in registers (static indices)
        const float b0 = bp[t * 512], b1 = bp[t * 512 + 32];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[0][t], b0, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[1][t], b0, acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[0][t], b1, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[1][t], b1, acc[1][1], 0, 0, 0);
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void gft_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ Fpad,
                                                     const float* __restrict__ bn_s, const float* __restrict__ bn_t, int C8) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ldsA = reinterpret_cast<float*>(smem);                     // [64][256] input chunk
    // A wave only ever touches its own 64 columns of the chunk (as B operand and as result), so the activated
    // group-domain values overwrite the coefficients in place and no barrier is needed after the initial load.
    float* ldsB = ldsA;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t chunk = blockIdx.x;                                  // = tile * C8 + c8
    const int c8 = (int)(chunk % (size_t)C8);
    stage_chunk_f(in + chunk * CHUNK_FLOATS, smem, w, lane);          // 60 x 1 KiB LDS DMA
    float4* l4 = reinterpret_cast<float4*>(ldsA);
    for (int i = tid; i < 4 * 256 / 4; i += 256) l4[CHUNK_FLOATS / 4 + i] = make_float4(0.f, 0.f, 0.f, 0.f);   // rows 60..63

    // transform matrices as MFMA A fragments.  Fpad holds F padded to 64x64 (row = coefficient, col = group
    // element) followed by its transpose, so that both fragment loads are coalesced (lane index = fastest axis).
    float aF[2][32], aFt[2][32];
    const int ai = lane & 31, ak = lane >> 5;
    const float* FpadT = Fpad + 64 * 64;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (MODE != GFT_INV) aF[rb][t] = FpadT[(2 * t + ak) * 64 + rb * 32 + ai];         // T = F:   F[row][k] = F^T[k][row]
            if (MODE != GFT_FWD) aFt[rb][t] = Fpad[(2 * t + ak) * 64 + rb * 32 + ai];         // T = F^T: F^T[row][k] = F[k][row]
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int cb0 = 2 * w;                                            // this wave's two 32-column blocks
    floatx16 acc[2][2];
    const int colj = lane & 31, half = lane >> 5;
    if (MODE == GFT_ACT) {
        gft_mm(aFt, ldsA, lane, cb0, acc);                            // group domain = F^T * coefficients
        // column -> channel: col = h*128 + kp*4 + e, channel = c8*8 + h*4 + e
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = (cb0 + c) * 32 + colj;
            const int ch = c8 * 8 + (col >> 7) * 4 + (col & 3);
            const float s = bn_s[ch], t = bn_t[ch];
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float y = fmaxf(acc[rb][c][r] * s + t, 0.f);
                    ldsB[row * 256 + col] = row < G ? y : 0.f;
                }
        }
        gft_mm(aF, ldsB, lane, cb0, acc);                             // coefficients = F * activated
    } else if (MODE == GFT_FWD) {
        gft_mm(aF, ldsA, lane, cb0, acc);
    } else {
        gft_mm(aFt, ldsA, lane, cb0, acc);
    }
    float* dst = out + chunk * CHUNK_FLOATS;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int col = (cb0 + c) * 32 + colj;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < G) dst[row * 256 + col] = acc[rb][c][r];
            }
    }
}

int gft_init() {
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gconvf_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_F));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gft_kernel<GFT_FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, GFT_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gft_kernel<GFT_INV>), hipFuncAttributeMaxDynamicSharedMemorySize, GFT_LDS));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gft_kernel<GFT_ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, GFT_LDS));
    return 0;
}

int launch_gft(int mode, const float* in, float* out, const float* Fpad, const float* bn_s, const float* bn_t, int nTiles, int C8, hipStream_t s) {
    const dim3 grid(nTiles * C8), blk(256);
    if (mode == GFT_FWD) hipLaunchKernelGGL(gft_kernel<GFT_FWD>, grid, blk, GFT_LDS, s, in, out, Fpad, bn_s, bn_t, C8);
    else if (mode == GFT_INV) hipLaunchKernelGGL(gft_kernel<GFT_INV>, grid, blk, GFT_LDS, s, in, out, Fpad, bn_s, bn_t, C8);
    else hipLaunchKernelGGL(gft_kernel<GFT_ACT>, grid, blk, GFT_LDS, s, in, out, Fpad, bn_s, bn_t, C8);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
