// Timing + numerics probe for the lever NOTEBOOK.md section 8 names for the next round: the two CORRECTION products of the fp16x2 split on the
// fp8 matrix pipe.
//
// The irrep GEMMs evaluate a * w as a_h w_h + a_h w_l + a_l w_h on v_mfma_f32_32x32x16_f16 (3 MFMAs per term).  The corrections are
// 2^-11 of the main product; tools/fp8_correction_study.py (CPU emulation of PartI) says they survive fp8 e4m3: 1.5e-5 worst relative error of
// the descriptor against 1.4e-6 today and a tolerance of 1e-4.  gfx950's v_mfma_scale_f32_32x32x64_f8f6f4 does K = 64 per instruction at
// twice the fp16 rate, and one such MFMA takes the corrections of TWO K16 steps at once by K-concatenation:
//     A' = [a_h(s0) | a_l(s0) | a_h(s1) | a_l(s1)]   (32 fp8 per lane),   B' = [w_l(s0) | w_h(s0) | w_l(s1) | w_h(s1)]
// so a K16 step costs 1 fp16 MFMA + half an fp8 MFMA of twice the length = 2/3 of today's matrix time, plus the conversion of the fp16
// fragments to fp8 in registers (v_cvt_scalef32_pk_fp8_f16: two values per instruction, 48 per wave and K16 step).
//
// This file measures what that buys BEFORE touching fgemm3: a K loop with fgemm3's shape (256 x 256 tile, eight waves of 128 x 64, K16 steps
// of 16 KiB A + 16 KiB B staged by LDS DMA through a ring of three buffers, one barrier per step, two workgroups' worth of registers per CU) in
// arithmetic variants - 0: three fp16 products (shipped), 1: fp16 main + fp8 corrections, 2: main product only (the floor), 3: as 1 with the
// weight operand's fp8 fragments read ready-made instead of converted (timing only: what a pre-packed weight plane would buy) - timed back
// to back under sustained load, and the accumulators of one tile compared with an fp64 evaluation on the host.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fp8_corr_probe.hip -o tools/_fp8_corr_probe && tools/_fp8_corr_probe [row tiles 16] [col tiles 32] [K16 steps 416]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef int intx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
typedef short shortx2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int PBUF = 32768, RING = 3, OPB = 16384;      // one K16 step: A planes 16 KiB | B planes 16 KiB
constexpr float LO_SCALE = 2048.f;                      // the lo planes enter the fp8 conversion times 2^11 (they are 2^-11 of the hi planes)

struct Args {
    const char* A;        // [row tile][K16 step][plane 2][row block 8][lane 64] x 16 B
    const char* B;        // [col tile][K16 step][plane 2][col block 8][lane 64] x 16 B
    float* out;           // accumulators of tile (0, 0): [wave 8][rb 4][cb 2][lane 64][16]
    int ntm, ntn, nsteps;
    float cvt_hi, cvt_lo; // scale operands of the fp8 conversion for hi / lo planes (set from the semantics probe)
};

__device__ __forceinline__ floatx16 mfma16(uintx4 a, uintx4 b, floatx16 c) {
    union { uintx4 u; halfx8 h; } ca, cb;
    ca.u = a; cb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ca.h, cb.h, c, 0, 0, 0);
}
// eight fp16 (one fragment) -> eight fp8 e4m3 (two dwords)
struct I2 { int x, y; };
__device__ __forceinline__ I2 to_fp8(uintx4 f, float scale) {
    union { unsigned u; halfx2 h; } p0, p1, p2, p3;
    p0.u = f[0]; p1.u = f[1]; p2.u = f[2]; p3.u = f[3];
    shortx2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, p0.h, scale, false);
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, p1.h, scale, true);
    shortx2 q = {0, 0};
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, p2.h, scale, false);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, p3.h, scale, true);
    I2 o;
    __builtin_memcpy(&o.x, &r, 4);
    __builtin_memcpy(&o.y, &q, 4);
    return o;
}
__device__ __forceinline__ void wait_frags(uintx4 (&f)[12]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                 "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]) :: "memory");
}
__device__ __forceinline__ constexpr int vm_wait(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

template <int V>
__global__ __launch_bounds__(512, 2) void kloop(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = w >> 2, wc = w & 3;
    const int mt = blockIdx.x / a.ntn, nt = blockIdx.x % a.ntn;
    const char* Ag = a.A + (size_t)mt * a.nsteps * OPB + lane * 16;
    const char* Bg = a.B + (size_t)nt * a.nsteps * OPB + lane * 16;
    floatx16 acc[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][c][e] = 0.f;
    auto stage = [&](int step, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int piece = w + 8 * i;                 // 0..15: A, 16..31: B
            const char* src = (piece >= 16 ? Bg + (size_t)(piece - 16) * 1024 : Ag + (size_t)piece * 1024) + (size_t)step * OPB;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * PBUF + piece * 1024), 16, 0, 0);
        }
    };
    uintx4 f[12];                                        // A hi x4, A lo x4, B hi x2, B lo x2
    intx8 a8[4], b8[2];                                  // V == 1: the fp8 operands of a pair of steps
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[r][e] = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) b8[c][e] = 0;
    stage(0, 0);
    stage(1, 1);
    auto one_step = [&](int step, auto parity) {
        constexpr int P = decltype(parity)::value;
        __builtin_amdgcn_s_waitcnt(vm_wait(4));          // this step has landed; the four pieces of the next one may stay in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        stage(step + 2 < a.nsteps ? step + 2 : a.nsteps - 1, (step + 2) % RING);
        const unsigned base = (unsigned)(size_t)(smem + (step % RING) * PBUF) + lane * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned ia = base + (4 * wr + r) * 1024;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192" : "=&v"(f[r]), "=&v"(f[4 + r]) : "v"(ia) : "memory");
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const unsigned ib = base + OPB + (2 * wc + c) * 1024;
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:8192" : "=&v"(f[8 + c]), "=&v"(f[10 + c]) : "v"(ib) : "memory");
        }
        wait_frags(f);
        // main product
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[r][c] = mfma16(f[r], f[8 + c], acc[r][c]);
        if constexpr (V == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[r][c] = mfma16(f[r], f[10 + c], acc[r][c]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[r][c] = mfma16(f[4 + r], f[8 + c], acc[r][c]);
        } else if constexpr (V == 3) {
            // timing only (values meaningless): the WEIGHT operand's fp8 fragments come pre-packed from memory - here the raw bytes of
            // the step's two A planes stand in for them - so only the activation fragments are converted (16 instead of 48 per step)
            if constexpr (P == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a8[r][e] = (int)f[r][e]; a8[r][4 + e] = (int)f[4 + r][e]; }
                }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const I2 l = to_fp8(f[10 + c], a.cvt_lo), h = to_fp8(f[8 + c], a.cvt_hi);
                b8[c][4 * P] = l.x; b8[c][4 * P + 1] = l.y; b8[c][4 * P + 2] = h.x; b8[c][4 * P + 3] = h.y;
            }
            if constexpr (P == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[r], b8[c], acc[r][c], 0, 0, 0, 127 - 11, 0, 127);
            }
        } else if constexpr (V == 1) {
            // A' dwords [4P .. 4P+1] = a_h, [4P+2 .. 4P+3] = a_l x 2^11;  B' = w_l x 2^11, w_h: element by element the pairs of the two corrections
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const I2 h = to_fp8(f[r], a.cvt_hi), l = to_fp8(f[4 + r], a.cvt_lo);
                a8[r][4 * P] = h.x; a8[r][4 * P + 1] = h.y; a8[r][4 * P + 2] = l.x; a8[r][4 * P + 3] = l.y;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const I2 l = to_fp8(f[10 + c], a.cvt_lo), h = to_fp8(f[8 + c], a.cvt_hi);
                b8[c][4 * P] = l.x; b8[c][4 * P + 1] = l.y; b8[c][4 * P + 2] = h.x; b8[c][4 * P + 3] = h.y;
            }
            if constexpr (P == 1) {
                // both operands carry one factor 2^11 in every product: undone by the block scale of A (E8M0: 127 = 2^0)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[r], b8[c], acc[r][c], 0, 0, 0, 127 - 11, 0, 127);
            }
        }
    };
    for (int step = 0; step < a.nsteps; step += 2) {
        one_step(step, std::integral_constant<int, 0>{});
        one_step(step + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (blockIdx.x == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) a.out[(((size_t)(w * 4 + r) * 2 + c) * 64 + lane) * 16 + e] = acc[r][c][e];
    } else {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int e = 0; e < 16; ++e) v += acc[r][c][e];
        if (v == 123456.f) a.out[0] = v;                 // keeps the accumulators alive
    }
}

// what does the conversion's scale operand do?  out[0] / out[1] = the fp8 byte of 3.0 converted with scale 4 / scale 1
__global__ void cvt_semantics(int* out) {
    union { unsigned u; halfx2 h; } p;
    p.h = halfx2{(_Float16)3.0f, (_Float16)3.0f};
    shortx2 r = {0, 0}, q = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, p.h, 4.0f, false);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, p.h, 1.0f, false);
    out[0] = (unsigned short)r[0] & 0xFF;
    out[1] = (unsigned short)q[0] & 0xFF;
}

static double e4m3(int b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    const double v = e == 0 ? std::ldexp(m / 8.0, -6) : std::ldexp(1.0 + m / 8.0, e - 7);
    return s ? -v : v;
}
static unsigned short f2h(float x) { _Float16 h = (_Float16)x; unsigned short u; std::memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; std::memcpy(&h, &u, 2); return (float)h; }
// operand value at (row or column index, k): magnitudes spread over 2^-4 .. 2 so that the fp8 planes see a range
static float val(unsigned idx, unsigned k, unsigned salt) {
    unsigned x = idx * 0x9E3779B1u ^ (k + salt) * 0x85EBCA77u;
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    const float u = (float)(x & 0xFFFFFF) / 16777216.f * 2.f - 1.f;
    return u * std::ldexp(2.f, -(int)((x >> 24) & 3));
}

int main(int argc, char** argv) {
    const int ntm = argc > 1 ? std::atoi(argv[1]) : 16, ntn = argc > 2 ? std::atoi(argv[2]) : 32;
    int nsteps = argc > 3 ? std::atoi(argv[3]) : 416;
    nsteps += nsteps & 1;
    int* dsem;
    CHK(hipMalloc((void**)&dsem, 64));
    hipLaunchKernelGGL(cvt_semantics, dim3(1), dim3(64), 0, 0, dsem);
    int hsem[2] = {0, 0};
    CHK(hipMemcpy(hsem, dsem, 8, hipMemcpyDeviceToHost));
    const int de = ((hsem[0] >> 3) & 15) - ((hsem[1] >> 3) & 15);          // exponent field with scale 4 minus with scale 1
    const bool divides = de == -2;
    std::printf("v_cvt_scalef32_pk_fp8_f16(3.0): scale 1 -> byte 0x%02x (%.4f as OCP e4m3), scale 4 -> 0x%02x (%.4f): the scale %s the source\n", hsem[1],
                e4m3(hsem[1]), hsem[0], e4m3(hsem[0]), divides ? "DIVIDES" : (de == 2 ? "MULTIPLIES" : "does something unexpected to"));
    const float cvt_hi = 1.0f, cvt_lo = divides ? 1.0f / LO_SCALE : LO_SCALE;
    // operands in fragment order: value (row, k) with row = 32 rb + lane % 32, k = 16 step + 8 (lane / 32) + e
    const size_t szA = (size_t)ntm * nsteps * OPB, szB = (size_t)ntn * nsteps * OPB;
    std::vector<unsigned short> hA(szA / 2), hB(szB / 2);
    auto fill = [&](std::vector<unsigned short>& dst, int ntiles, unsigned salt) {
        for (int t = 0; t < ntiles; ++t)
            for (int s = 0; s < nsteps; ++s)
                for (int blk = 0; blk < 8; ++blk)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 8; ++e) {
                            const float x = val(t * 256 + blk * 32 + (l & 31), 16 * s + 8 * (l >> 5) + e, salt);
                            const unsigned short hi = f2h(x), lo = f2h(x - h2f(hi));
                            const size_t o = ((size_t)t * nsteps + s) * (OPB / 2) + ((size_t)blk * 64 + l) * 8 + e;
                            dst[o] = hi;
                            dst[o + 8 * 64 * 8] = lo;
                        }
    };
    fill(hA, ntm, 17u);
    fill(hB, ntn, 91u);
    char *A, *B; float* out;
    CHK(hipMalloc((void**)&A, szA)); CHK(hipMalloc((void**)&B, szB)); CHK(hipMalloc((void**)&out, (size_t)8 * 4 * 2 * 64 * 16 * 4));
    CHK(hipMemcpy(A, hA.data(), szA, hipMemcpyHostToDevice));
    CHK(hipMemcpy(B, hB.data(), szB, hipMemcpyHostToDevice));
    typedef void (*kern_t)(Args);
    kern_t kerns[4] = {kloop<0>, kloop<1>, kloop<2>, kloop<3>};
    for (int v = 0; v < 4; ++v) CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[v]), hipFuncAttributeMaxDynamicSharedMemorySize, RING * PBUF));
    Args a{A, B, out, ntm, ntn, nsteps, cvt_hi, cvt_lo};
    const int grid = ntm * ntn;
    std::printf("K loop of a 256 x 256 irrep-GEMM tile (8 waves of 128 x 64, K16 steps of 32 KiB by LDS DMA, ring of 3): %d x %d tiles, %d K16 steps (K = %d)\n",
                ntm, ntn, nsteps, 16 * nsteps);
    // ---- numerics: tile (0, 0) against fp64 on the host (x = hi + lo exactly as the planes hold it)
    std::vector<double> ref(256 * 256, 0.0);
    {
        std::vector<double> av((size_t)256 * 16 * nsteps), bv((size_t)256 * 16 * nsteps);
        for (int r = 0; r < 256; ++r)
            for (int k = 0; k < 16 * nsteps; ++k) {
                const float x = val(r, k, 17u), y = val(r, k, 91u);
                const unsigned short xh = f2h(x), yh = f2h(y);
                av[(size_t)r * 16 * nsteps + k] = (double)h2f(xh) + (double)h2f(f2h(x - h2f(xh)));
                bv[(size_t)r * 16 * nsteps + k] = (double)h2f(yh) + (double)h2f(f2h(y - h2f(yh)));
            }
        for (int i = 0; i < 256; ++i)
            for (int j = 0; j < 256; ++j) {
                double s = 0.0;
                const double* pa = &av[(size_t)i * 16 * nsteps];
                const double* pb = &bv[(size_t)j * 16 * nsteps];
                for (int k = 0; k < 16 * nsteps; ++k) s += pa[k] * pb[k];
                ref[i * 256 + j] = s;
            }
    }
    double rmax = 0.0;
    for (double v : ref) rmax = std::fmax(rmax, std::fabs(v));
    std::vector<float> hout((size_t)8 * 4 * 2 * 64 * 16);
    const char* names[4] = {"3 fp16 products (shipped arithmetic)", "fp16 main + fp8 e4m3 corrections", "main product only", "(timing only) weights pre-packed in fp8"};
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    double ms_of[4] = {0, 0, 0, 0};
    for (int round = 0; round < 2; ++round)
        for (int v = 0; v < 4; ++v) {
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kerns[v], dim3(grid), dim3(512), RING * PBUF, 0, a);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kerns[v], dim3(grid), dim3(512), RING * PBUF, 0, a);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 10;
            ms_of[v] = ms;
            CHK(hipMemcpy(hout.data(), out, hout.size() * 4, hipMemcpyDeviceToHost));
            double err = 0.0;
            for (int w = 0; w < 8; ++w)
                for (int r = 0; r < 4; ++r)
                    for (int c = 0; c < 2; ++c)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 16; ++e) {
                                const int row = 128 * (w >> 2) + 32 * r + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
                                const int col = 64 * (w & 3) + 32 * c + (l & 31);
                                // A is the M (row) operand: D[i][j], i from A's lane, j from B's lane
                                const double d = std::fabs((double)hout[(((size_t)(w * 4 + r) * 2 + c) * 64 + l) * 16 + e] - ref[row * 256 + col]);
                                err = std::fmax(err, d);
                            }
            const double flops3 = 3.0 * 2.0 * 256.0 * 256.0 * 16.0 * nsteps * grid;      // what the shipped arithmetic issues for this launch
            std::printf("  run %d  %-38s %8.3f ms per launch  (%.0f TFLOP/s in units of the shipped 3-product count)   tile (0,0): max |error| / max |C| = %.2e\n",
                        round, names[v], ms, flops3 / (ms * 1e-3) / 1e12, err / rmax);
        }
    std::printf("speed-up of the fp8-corrected loop over the shipped one: %.3f x (floor = main product only: %.3f x; with the weight operand pre-packed in fp8: %.3f x)\n", ms_of[0] / ms_of[1], ms_of[0] / ms_of[2], ms_of[0] / ms_of[3]);
    return 0;
}
