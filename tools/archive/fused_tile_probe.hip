// Timing probe for VERDICT r3 item 6: the "all-60-coefficient" fused tile of one PartI layer (512 -> 256 over 10000 keypoints).
//
// Today a layer is  fgemm3 (irrep GEMMs, 256 x 256 output tiles, fp32 coefficients to HBM)  +  gft16x (BN + ReLU between two
// 60 x 60 transforms, coefficients -> fp16x2 operand planes).  The transform needs all 60 Fourier coefficients of a (channel,
// keypoint), so fusing it into the GEMM epilogue means ONE workgroup must own all 60 coefficient tiles of its (cout, keypoint) block:
// 60 x 32 x 32 fp32 accumulators = 240 KB of registers = the 32 cout x 32 kp tile with 120-128 accumulator registers in each of 8
// waves (the register file of a CU holds nothing larger).  This file measures what that costs BEFORE any epilogue: the K loop alone
// - operand staging by LDS DMA, fragment reads, MFMAs - with the real operand volumes of the formulation:
//
//   per K16 step a tile needs, for every irrep (d = 1,3,3,4,5) and every m < d, the weight fragments W^(rho,i,m) (d of them) and the
//   activation fragments X^(rho,m,j) (d of them), hi and lo plane: sum_rho d * 2d * 2 KB = 240 KB (more than the 160 KB of LDS, so a
//   K16 step cannot be staged at once: here ten phases of <= 32 KB through a ring of four buffers, DMA three phases ahead, ten
//   barriers per step against fgemm3's one), and issues sum_rho d^3 * 3 = 732 MFMAs (32x32x16 f16).  A 256 x 256 fgemm3 tile issues
//   192 MFMAs per 32 KB staged: the fused tile moves 2.0x the bytes per MFMA through L2 -> LDS and, because a fragment is shared by
//   only d <= 5 tiles spread over the waves, ~1.3 LDS fragment reads per MFMA against 0.5.
//
// The probe is FAVOURABLE to the fused tile: every wave runs the same branch-free program with 7 of the 7.5 accumulator tiles a real
// split would give it (696 of 732 MFMAs per step), no residual, no epilogue (the two in-register transforms, BN + ReLU, the fp16x2
// pack and the plane stores - gft16x's 0.3 ms of work - would come on top), operands from dense buffers of the real total size.
// Variants: 0 = all, 1 = no LDS fragment reads (MFMAs on stale registers: DMA + MFMA + barriers only), 2 = no MFMAs, 4 = no DMA.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fused_tile_probe.hip -o tools/_fused_tile_probe && tools/_fused_tile_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int NPH = 10;                            // phases of one K16 step
constexpr int RING = 4;                            // LDS ring of 32 KiB buffers
constexpr int PBUF = 32768;
constexpr int STEP_BYTES = 120 * 1024;             // per operand (A or B) and K16 step: sum_rho d * d frags * 2 planes * 1 KiB
// phase p: bytes staged per operand, DMA pieces (1 KiB) per wave (A and B together), accumulator tiles touched per slice, slices
struct Phase { int bytes; int pieces; int tiles; int slices; int acc0; };
//                              d5 m0..m4 (10 KiB per operand and slice)                         d4 (8 KiB)              d3 (6 KiB), d1 (2 KiB)
constexpr Phase PH[NPH] = {{10240, 3, 3, 1, 0}, {10240, 3, 3, 1, 0}, {10240, 3, 3, 1, 0}, {10240, 3, 3, 1, 0}, {10240, 3, 3, 1, 0},
                           {16384, 4, 2, 2, 3}, {16384, 4, 2, 2, 3}, {12288, 3, 1, 2, 5}, {12288, 3, 1, 2, 5}, {14336, 4, 1, 2, 6}};
constexpr int phase_off(int p) { int o = 0; for (int i = 0; i < p; ++i) o += PH[i].bytes; return o; }
static_assert(phase_off(NPH) == STEP_BYTES, "phases must add up to one K16 step");

template <int I, int N, typename Fn>
__device__ __forceinline__ void sfor(Fn&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

__device__ __forceinline__ floatx16 mfma(uintx4 a, uintx4 b, floatx16 c) {
    union { uintx4 u; halfx8 h; } ca, cb;
    ca.u = a; cb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ca.h, cb.h, c, 0, 0, 0);
}

// the four fragments of one accumulator tile (A hi, A lo, B hi, B lo; lo planes OFF bytes behind the hi ones)
template <int OFF>
__device__ __forceinline__ void read4(unsigned ia, unsigned ib, uintx4& a0, uintx4& a1, uintx4& b0, uintx4& b1) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:%6\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:%6"
                 : "=&v"(a0), "=&v"(a1), "=&v"(b0), "=&v"(b1) : "v"(ia), "v"(ib), "n"(OFF) : "memory");
}
__device__ __forceinline__ void wait_frags(uintx4 (&f)[12]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                 "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]) :: "memory");
}

// vmcnt(n) expcnt(7) lgkmcnt(15) through the builtin (the compiler's scoreboard sees it): gfx9 encoding, vmcnt = n[3:0] | n[5:4] << 14
__device__ __forceinline__ constexpr int vm_wait(int n) { return (n & 15) | ((n >> 4) << 14) | 0x0F70; }

struct Args {
    const char* A;        // [cout block 8][K16 step][120 KiB]
    const char* B;        // [keypoint tile][K16 step][120 KiB]
    float* out;
    int nkt, nsteps, variant;
};

template <int V>
__global__ __launch_bounds__(512, 2) void fused_tile_kloop(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the 8 cout blocks of a keypoint tile run next to each other on one XCD (its activation operands are served from that L2)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int kt = (slot >> 3) * 8 + xcd, ob = slot & 7;
    if (kt >= a.nkt) return;
    const char* Ag = a.A + (size_t)ob * a.nsteps * STEP_BYTES + lane * 16;
    const char* Bg = a.B + (size_t)kt * a.nsteps * STEP_BYTES + lane * 16;
    constexpr bool rd = !(V & 1), mm = !(V & 2), dma = !(V & 4);      // compile-time: no branch inside the loop

    floatx16 acc[7];
#pragma unroll
    for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    uintx4 f[12];                                   // fragments of up to 3 tiles: A hi, A lo, B hi, B lo each
#pragma unroll
    for (int i = 0; i < 12; ++i) f[i] = uintx4{0x3c003c00u + (unsigned)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};

    // DMA of phase P of K16 step `step` into ring buffer `buf`: pieces w, w + 8, ... of the phase's A bytes followed by its B bytes
    auto stage = [&](auto pc, int step, int buf) {
        constexpr int P = decltype(pc)::value;
        constexpr int per_op = PH[P].bytes / 1024;                  // pieces per operand
#pragma unroll
        for (int i = 0; i < (dma ? PH[P].pieces : 0); ++i) {
            int piece = w + 8 * i;
            piece = piece < 2 * per_op ? piece : 2 * per_op - 1;      // a wave past the end repeats the last piece: uniform counts
            const bool isB = piece >= per_op;
            const char* src = (isB ? Bg : Ag) + (size_t)step * STEP_BYTES + phase_off(P) + (size_t)(isB ? piece - per_op : piece) * 1024;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * PBUF + piece * 1024), 16, 0, 0);
        }
    };
    const int total = a.nsteps * NPH;
    // prologue: phases 0, 1, 2 of step 0
    stage(std::integral_constant<int, 0>{}, 0, 0);
    stage(std::integral_constant<int, 1>{}, 0, 1);
    stage(std::integral_constant<int, 2>{}, 0, 2);
    int g = 0;                                       // global phase counter: ring buffer = g % RING
    for (int step = 0; step < a.nsteps; ++step) {
        sfor<0, NPH>([&](auto pc) {
            constexpr int P = decltype(pc)::value;
            constexpr int P1 = (P + 1) % NPH, P2 = (P + 2) % NPH, P3 = (P + 3) % NPH;
            // phase g must have landed; the pieces of g + 1 and g + 2 may stay in flight (completion is in issue order)
            __builtin_amdgcn_s_waitcnt(vm_wait(dma ? PH[P1].pieces + PH[P2].pieces : 0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            // refill the buffer read one phase ago with phase g + 3 (clamped at the end: re-stages the last phases, harmless)
            {
                const int s3 = step + (P + 3 >= NPH ? 1 : 0);
                stage(std::integral_constant<int, P3>{}, s3 < a.nsteps ? s3 : a.nsteps - 1, (g + 3) % RING);
            }
            const unsigned base = (unsigned)(size_t)(smem + (g % RING) * PBUF) + lane * 16;
            constexpr int per_op = PH[P].bytes / 1024;
            sfor<0, PH[P].slices>([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                constexpr int NT = PH[P].tiles;
                constexpr int fr = per_op / PH[P].slices / 2;         // fragments per plane, operand and slice (= d)
                if constexpr (rd) {
                    // fragment (tile n): A plane hi at row (w + n) % fr, lo behind the hi fragments; B likewise in the operand's second half
                    sfor<0, NT>([&](auto nc) {
                        constexpr int n = decltype(nc)::value;
                        const unsigned ia = base + ((S * 2 * fr) + (w + n) % fr) * 1024, ib = base + (per_op + (S * 2 * fr) + (w + 2 * n + 1) % fr) * 1024;
                        read4<fr * 1024>(ia, ib, f[4 * n], f[4 * n + 1], f[4 * n + 2], f[4 * n + 3]);
                    });
                    wait_frags(f);
                }
                if constexpr (mm) {
                    sfor<0, NT>([&](auto nc) { constexpr int n = decltype(nc)::value; acc[PH[P].acc0 + n] = mfma(f[4 * n + 1], f[4 * n + 2], acc[PH[P].acc0 + n]); });
                    sfor<0, NT>([&](auto nc) { constexpr int n = decltype(nc)::value; acc[PH[P].acc0 + n] = mfma(f[4 * n], f[4 * n + 3], acc[PH[P].acc0 + n]); });
                    sfor<0, NT>([&](auto nc) { constexpr int n = decltype(nc)::value; acc[PH[P].acc0 + n] = mfma(f[4 * n], f[4 * n + 2], acc[PH[P].acc0 + n]); });
                }
            });
            ++g;
        });
    }
    (void)total;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // keep the accumulators alive
    floatx16 s = acc[0];
#pragma unroll
    for (int t = 1; t < 7; ++t) s += acc[t];
    float v = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) v += s[e];
    if (v == 123456.f) a.out[blockIdx.x * 512 + threadIdx.x] = v;
}

int main(int argc, char** argv) {
    const int kp = argc > 1 ? std::atoi(argv[1]) : 10000, cin = argc > 2 ? std::atoi(argv[2]) : 512, cout = 256;
    const int nkt = (kp + 31) / 32, nsteps = cin / 16, nob = cout / 32;
    const size_t szA = (size_t)nob * nsteps * STEP_BYTES, szB = (size_t)nkt * nsteps * STEP_BYTES;
    char *A, *B; float* out;
    CHK(hipMalloc((void**)&A, szA)); CHK(hipMalloc((void**)&B, szB)); CHK(hipMalloc((void**)&out, (size_t)4 << 20));
    CHK(hipMemset(A, 0x3c, szA)); CHK(hipMemset(B, 0x3c, szB));         // fp16 1.0-ish everywhere: finite products
    typedef void (*kern_t)(Args);
    kern_t kerns[5] = {fused_tile_kloop<0>, fused_tile_kloop<1>, fused_tile_kloop<2>, nullptr, fused_tile_kloop<4>};
    for (int v : {0, 1, 2, 4}) CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kerns[v]), hipFuncAttributeMaxDynamicSharedMemorySize, RING * PBUF));
    const int grid = 8 * ((nkt + 7) / 8) * 8;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    std::printf("fused all-60-coefficient tile, K loop only: %d keypoints (%d tiles of 32), cin %d (%d K16 steps), cout %d (%d blocks of 32): %d workgroups of 8 waves,\n"
                "  operands: weights %.1f MB, activations %.1f MB; staged per workgroup %.2f MB, per launch %.2f GB (fgemm3 for the same layer: 4.0 GB)\n",
                kp, nkt, cin, nsteps, cout, nob, nkt * nob, szA / 1e6, szB / 1e6, 2.0 * nsteps * STEP_BYTES / 1e6, 2.0 * nsteps * STEP_BYTES * nkt * nob / 1e9);
    const double mfma_flops = 696.0 * 32768.0 * nsteps * nkt * nob;
    for (int variant : {0, 1, 2, 4, 0}) {
        Args a{A, B, out, nkt, nsteps, variant};
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kerns[variant], dim3(grid), dim3(512), RING * PBUF, 0, a);
        CHK(hipDeviceSynchronize());
        // eight launches back to back (sustained clock), the mean of them
        CHK(hipEventRecord(e0, 0));
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(kerns[variant], dim3(grid), dim3(512), RING * PBUF, 0, a);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 8;
        std::printf("variant %d (%s): %.3f ms per launch", variant,
                    variant == 0 ? "DMA + fragment reads + MFMAs" : variant == 1 ? "no LDS fragment reads" : variant == 2 ? "no MFMAs" : "no DMA", ms);
        if (!(variant & 2)) std::printf("  = %.0f TFLOP/s of issued fp16 MFMA (%.3f of 2500)", mfma_flops / (ms * 1e-3) / 1e12, mfma_flops / (ms * 1e-3) / 1e12 / 2500.0);
        std::printf("\n");
    }
    std::printf("compare: fgemm3 512 -> 256 (K loop + coefficient stores) + gft16x 256 ch of the same pass: see bench.py roofline_extra.launch_ms[2] + hbm.gft16 256ch\n");
    return 0;
}
