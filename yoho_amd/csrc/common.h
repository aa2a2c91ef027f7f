// Internal declarations shared by the translation units of libyoho_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <initializer_list>
#include "yoho_hip.h"

namespace yoho {

constexpr int G = 60;          // icosahedral rotation group order
constexpr int NTAP = 13;       // self + 12 neighbours (Nei_Index_in_SO3_ordered_13)
constexpr int F = 32;          // FCGF feature width
constexpr int TILE = 32;       // keypoints (or matches) per tile = MFMA N dimension
// One "slab" = the 8 channels of one c8-chunk for one group element and one tile:
// [h=2][kp=32][e=4] floats; channel c = c8*8 + h*4 + e.  A chunk = 60 slabs.
constexpr int SLAB_FLOATS = 2 * TILE * 4;          // 256 floats = 1 KiB
constexpr int CHUNK_FLOATS = G * SLAB_FLOATS;      // 15360 floats = 60 KiB

void set_error(const char* fmt, ...);
void clear_error();
// device tensors at the ABI are read with 16-byte vector loads and LDS DMA (include/yoho_hip.h: "contiguous, 16-byte aligned"); f64 /
// int64 arrays that are only ever read element-wise need their natural 8 bytes.  Null pointers pass (optional arguments).
inline bool yoho_misaligned(std::initializer_list<const void*> ps, unsigned mask) {
    for (const void* p : ps) if ((reinterpret_cast<uintptr_t>(p) & mask) != 0) return true;
    return false;
}
#define YOHO_NEED_ALIGNED(fn, mask, ...)                                                                                  \
    do {                                                                                                                  \
        if (::yoho::yoho_misaligned({__VA_ARGS__}, (mask))) {                                                             \
            ::yoho::set_error(fn ": a device pointer is not %d-byte aligned", (int)(mask) + 1);                           \
            return YOHO_EINVAL;                                                                                           \
        }                                                                                                                 \
    } while (0)
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define HIPCHK(expr)                                                         \
    do {                                                                     \
        hipError_t e_ = (expr);                                              \
        if (e_ != hipSuccess) return ::yoho::hip_fail(e_, #expr, __FILE__, __LINE__); \
    } while (0)

// One group-conv (or 1x1) layer as it lives on the device.
constexpr float HF_ASCALE = 4.f;      // Fourier-coefficient planes of the irrep GEMMs are stored as 4 * x (|x| < 16376)
constexpr int NIR_ORD = 5;            // irreps of the icosahedral group
constexpr float H2_ASCALE = 16.f;     // fp16x2 activations are stored as 16 * x (|x| < 4094)

// fp16 range guard.  The fp16x2 paths store activations / Fourier coefficients as fixed power-of-two multiples in fp16
// planes; a value beyond the fp16 range would become +-inf there and (through relu(NaN) = 0) could vanish silently.
// Every kernel that converts to fp16 tracks the largest magnitude it converted and raises the word `flag` points to
// (device memory owned by the context: one word per network) when it exceeds the largest finite fp16 number; the host
// reads it with yoho_range_status and repeats the pass in the bf16x3 format (fp32 exponent range).
constexpr float FP16_MAX = 65504.f;
__device__ __forceinline__ void note_range(int* flag, float amax) {
    if (flag && !(amax <= FP16_MAX)) atomicOr(flag, 1);
}
// the same for a running maximum kept as the bit pattern of |x| (unsigned order: finite < inf < NaN)
__device__ __forceinline__ void note_range_bits(int* flag, unsigned top, float limit = FP16_MAX) {
    if (flag && top > __float_as_uint(limit)) atomicOr(flag, 1);
}

// wave-uniform slot tables of the direct-conv kernels (gconv.hip / gconv16.hip): device memory owned by the context, handed to the
// kernels in their arguments and read there through the constant address space (scalar loads), so that contexts built on
// different group tables can coexist in one process (round 2 kept them in __constant__ objects of the code object)
struct SlotTables {
    int* slabtab = nullptr;   // [NCFG][13 * 60]  LDS byte offset of input slab N[g(slot), tap]
    int* outg = nullptr;      // [NCFG][60]       output group element of a slot, -1 = unused
    int* slab4 = nullptr;     // [3][7 * 32]      gconv16: four slab indices of a (tap pair, unit), one per byte
    int* unitg = nullptr;     // [3][32 * 2]      gconv16: output group elements (ga, gb) of a unit, -1 = unused
};

struct Layer {
    const SlotTables* tabs = nullptr;   // the owning context's slot tables (set by yoho_load_partI / yoho_load_partII for every layer they build)
    int cin = 0, cout = 0, cout_pad = 0, ntaps = 0;
    float* wp = nullptr;      // packed MFMA A-fragments [ob][c8][tap][lane64][4]
    void* wp16 = nullptr;     // bf16x3 planes [ob][c8][tap-pair 7][plane 3][lane64][8] (13-tap layers only)
    float wph_descale = 1.f;  // 1 / (power-of-two weight scale * H2_ASCALE) of the fp16x2 planes
    void* wph = nullptr;      // fp16x2 planes [ob][c8][tap-pair 7][plane 2][lane64][8] (13-tap layers only)
    float* wpf = nullptr;
    void* wpg = nullptr;      // irrep-GEMM A operand: fp16x2 planes of What (gemmf.hip), 13-tap layers with cout % 256 == 0
    void* wpg8 = nullptr;     // the same pack for fgemm3c (gconv_mode 7): hi plane as in wpg, the lo plane's 16-byte units replaced by the unit's fp8 e4m3
                              // operand [hi / 4 (8 values) | lo * 512 (8 values)]; layers with cin and cout >= 256 only
    float wpg_descale = 1.f;     // group-Fourier weights [ob][c8][frag 60][lane64][4] (13-tap layers only)
    void* wcg = nullptr;      // cone GEMM A operand (gemmf2.hip cgemm_kernel; PartII's 13-element cone layer only): the plain [cout][tap * cin + c]
    void* wcg8 = nullptr;     // matrix in fgemm's A pack - fp16 hi + lo planes (wcg) or hi + fp8 correction operand (wcg8)
    float wcg_descale = 1.f;
    float* bias = nullptr;    // [cout_pad]
    float* bn_s = nullptr;    // [cout_pad] scale of the BN that FOLLOWS this conv (applied with ReLU in the epilogue)
    float* bn_t = nullptr;    // [cout_pad] shift
};

// epilogue flags
enum { EPI_RES = 1, EPI_RAW = 2, EPI_ACT = 4, EPI_RAW32 = 8, EPI_ACT32 = 16 };

struct ConvArgs {
    const float* X;        // activated input, internal layout [tile][cin/8][60 slabs][256]
    const float* Wp;
    const float* bias;
    const float* bn_s;
    const float* bn_t;
    const float* res;      // raw residual, layout of the output (EPI_RES)
    float* out_raw;        // EPI_RAW
    float* out_act;        // EPI_ACT: relu(v*bn_s + bn_t)
    int nTiles, cin8, cout8, nOB, ntaps;
    const int* slabtab;    // SlotTables::slabtab / outg of the context
    const int* outg;
};

// slot-table configurations (constant memory, see gconv.hip)
enum { CFG_FULL = 0, CFG_C45 = 1, CFG_C13 = 2, CFG_C1 = 3, NCFG = 4 };
int upload_slot_tables(const int* slab_h /*[NCFG][13*60]*/, const int* outg_h /*[NCFG][60]*/, SlotTables& t);
bool mlp_head_supported(const Layer& A, const Layer& B, const Layer& C);      // PartII's 1x1 tail in one launch (gconv.hip)
int launch_mlp_head(const Layer& A, const Layer& B, const Layer& C, const float* X, int nTiles, int M, float* quat, hipStream_t s,
                    const float* part = nullptr, const Layer* P = nullptr, const float* res = nullptr);
int launch_gconv(const ConvArgs& a, int gpw, int flags, hipStream_t s);
int gconv_init();   // sets the dynamic-LDS attribute of every instantiation

int launch_pack_partI(const float* x, int B, int nTiles, float* out, hipStream_t s);
int launch_finalize_partI(const float* y, const float* x, int B, float* eqv, float* inv, float* inv_np, int layout16, hipStream_t s,
                          const float* x1 = nullptr, int B0 = 0);
// group-Fourier variant (fourier.hip)
struct FourierBasis {
    double rho[5][60][25];    // rho[r][g][a*d + b], real orthogonal irreps of dimension 1, 3, 3, 4, 5
    double F[60 * 60];        // F[(r,i,j)][g] = sqrt(d/60) rho_r(g)[i][j]  (orthogonal 60 x 60)
    int n0[13];               // N[0][k]
};
int build_fourier(const uint8_t* N, const uint8_t* P, FourierBasis& fb);
}  // namespace yoho
#include <vector>
#include <cstdlib>
namespace yoho {
void pack_fourier_weights(const FourierBasis& fb, const float* W, int cin, int cout, int cout_pad, std::vector<float>& out);
int gft_init();
int fgemm_init();
size_t fgemm_planes_bytes(int kppad, int cin);
void fgemm_qinfo(int* qi);
void fgemm_plane_offsets(int kppad, int cin, long long* off);
int pack_fgemm_weights(const FourierBasis& fb, const float* W, int cin, int cout, std::vector<unsigned short>& out, float* descale,
                       std::vector<unsigned short>* out8 = nullptr);
int pack_cgemm_weights(const float* W, int cin, int cout, int ntaps, std::vector<unsigned short>& out, float* descale, std::vector<unsigned short>& out8);
int cgemm_init();
int launch_cgemm(const Layer& L, const char* Bstages, int nslot, const unsigned char* slot, const unsigned char* outg, int nT32, int nTiles16,
                 char* out, hipStream_t s, int* rflag, const unsigned* amax);
int launch_gft16_invg(const float* in, float* res0, char* planesG, const int* slot_of, int nslot, const void* Ffrag, const float* bn_s,
                      const float* bn_t, int nTiles, int C8, int nCU, hipStream_t s, int* rflag, unsigned* amax);
int launch_fgemm(const Layer& L, const char* Bplanes, int kppad, int nT32, const float* res, float* out, int flags, hipStream_t s,
                 int* rflag = nullptr, int variant = 2, const unsigned* amax = nullptr);
void build_gft16_frags(const FourierBasis& fb, std::vector<unsigned short>& out);
int gconv_layer(yoho_ctx* c, const float* x, int B, int cin, int cout, const float* W, const float* bias, int transpose, float* y,
                hipStream_t s);
int bn_stats(const float* x, int B, int C, float* mean, float* var, hipStream_t s);
int bn_relu_apply(const float* x, int B, int C, const float* scale, const float* shift, float* y, hipStream_t s);
int bn_relu_backward(const float* x, const float* y, const float* dy, int B, int C, const float* gamma, const float* mean, const float* rstd,
                     int batch_stats, float* dx, float* dgamma, float* dbeta, hipStream_t s);
int gconv_wgrad(yoho_ctx* c, const float* x, const float* dy, int B, int cin, int cout, float* dW, float* db, hipStream_t s);
struct FcgfNet;
int fcgf_load(FcgfNet** out, const yoho_fcgf_config* cfg, const float* const* t, int ntensors, bool f32_kernels);
void fcgf_free(FcgfNet* n);
int fcgf_forward(yoho_ctx* ctx, const FcgfNet* net, const int* coords0, int n0, const int* off_host, int nb, float* out, hipStream_t s);
int fcgf_voxelize(yoho_ctx* ctx, const double* pts, int n, const double* R_host, double voxel, int64_t* sel, int* coords, float* pts_sel,
                  int* count_host, hipStream_t s);
int fcgf_rotate_select(const double* pts, const double* R_host, const int64_t* sel, int m, float* out, hipStream_t s);
int fcgf_voxelize_batch(yoho_ctx* ctx, const double* pts, int n, const double* R_host, int nb, double voxel, int64_t* sel, int* coords,
                        float* pts_sel, int* counts_host, hipStream_t s);
int launch_cone1(const Layer& L, const char* X, int nTiles32, int nTiles16, const float* res, float* out, const int* n0, hipStream_t s, float* part = nullptr);
int gft16_init();
int launch_gft16(const float* in, float* out32, char* planes, int kppad, const void* Ffrag, const float* bn_s, const float* bn_t, int nTiles,
                 int C8, int nCU, hipStream_t s, int B = 0, int* rflag = nullptr, int variant = 2, int* ctr = nullptr, unsigned* amax = nullptr);
int launch_gft16_invp(const float* in, float* res0, char* planes16, int nTiles16, const void* Ffrag, const float* bn_s, const float* bn_t,
                      int nTiles, int C8, int nCU, hipStream_t s, int* rflag = nullptr);
int launch_head2(const float* s0, const float* s1, const float* s2, const float* s3, const int64_t* pre_idx, const int* P, const float* bn_s,
                 const float* bn_t, int M, int nTiles, char* planes, int kppad, const void* Ffrag, hipStream_t s,
                 const int64_t* const* ridx = nullptr, int istride = 1, int* rflag = nullptr);
int launch_head16(const float* x, int B, int nTiles, char* planes, int kppad, const void* Ffrag, hipStream_t s, const float* x1 = nullptr,
                  int B0 = 0, int* rflag = nullptr);
int launch_gconvf(const Layer& L, const float* X, int nTiles, const float* res, float* out, int flags, hipStream_t s);
int launch_gft(int mode, const float* in, float* out, const float* Fpad, const float* bn_s, const float* bn_t, int nTiles, int C8, hipStream_t s);
// bf16x3 variant (gconv16.hip)
int upload_slot_tables16(const int* slab4_h, const int* unitg_h, SlotTables& t);
int gconv16_init();
int launch_gconv16(const Layer& L, const char* X, int nTiles, const float* res, float* out_raw, char* out_act, int flags, hipStream_t s,
                   int cfg = 0, float* out_raw32 = nullptr, float* out_act32 = nullptr, int npl = 3, int* rflag = nullptr);
int launch_pack16_partII(const float* s0, const float* s1, const float* s2, const float* s3, const int64_t* pre_idx, const int* P,
                         const float* bn_s, const float* bn_t, int M, int nTiles16, char* out, hipStream_t s, int npl = 3, int* rflag = nullptr);
int launch_pack16_partI(const float* x, int B, int nTiles, char* out, hipStream_t s, int npl = 3, int* rflag = nullptr);
int launch_group_mean_np(const float* eqv, int B, float* out, hipStream_t s);
int launch_pack_partII(const float* s0, const float* s1, const float* s2, const float* s3, const int64_t* pre_idx,
                       const int* P, const float* bn_s, const float* bn_t, int M, int nTiles, float* out, hipStream_t s);
int launch_quat_norm(const float* y, int M, float* quat, hipStream_t s);

size_t mutual_prefilter_ws_bytes(int Na, int Nb);
int launch_mutual_prefilter(const float* a, int Na, const float* b, int Nb, void* ws, unsigned long long** keysA, unsigned long long** keysB,
                            int nCU, hipStream_t s, int nn_splits = 0);
size_t grid_transfer_ws_bytes(int K, int nb, int mmax);
int launch_grid_transfer_batch(const double* pts, const int64_t* kidx, int K, const double* R_host, int nb, const float* const* ds,
                               const float* const* feat, const int* m, int g0, float* out, double cell, void* ws, int nCU, hipStream_t s);
struct Workspace;
}  // namespace yoho
struct yoho_ctx;
namespace yoho {
int ensure_ws(yoho_ctx* ctx, size_t bytes, hipStream_t s);
struct GnMat3;
size_t grid_nn_ws_bytes(int Ns, int Nt);
int launch_grid_nn(int mode, const void* src, int Ns, const GnMat3* R, const float* tgt, int Nt, double cell, void* ws, int64_t* idx, float* dist,
                   double* part_d, int* part_i, int nCU, hipStream_t s);

struct Workspace {
    void* p = nullptr;
    size_t bytes = 0;
};

// Timing-experiment switches (YOHO_PARTI_DEBUG, YOHO_FGEMM_DEBUG: serialised chunks, GEMMs without their stores, drained waits ...)
// exist only in a build made with -DYOHO_EXPERIMENTS (YOHO_EXPERIMENTS=1 python -m yoho_amd.build --force): the shipped library
// does not read them, so no environment variable can change what a production pass computes or how it is ordered.
inline const char* experiment_env(const char* name) {
#ifdef YOHO_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// Accumulating phase timer for the entries that are many launches long (the FCGF path: voxelisation, coordinate / kernel maps, the
// convolutions of each level, the feature transfer).  phase_mark(ctx, cat, s) records ONE event on the launch stream: it ends the
// span that is open and starts one in category `cat` (cat < 0: only ends); yoho_phase_read sums the spans per category.  Beside
// the time a category collects the fp16 MFMA flops its launches issue (counted on the host at launch).  Off by default
// (yoho_phase_profile): then a mark is one load and a branch.
struct PhaseProf {
    static constexpr int NCAT = 16;
    bool on = false;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    struct Span { int cat; hipEvent_t a, b; };
    std::vector<Span> spans;
    int open_cat = -1;
    hipEvent_t open_ev = nullptr;
    double flops[NCAT] = {0};
    double launches[NCAT] = {0};
};

}  // namespace yoho

// A/B and diagnostic switches of the environment.  They are read ONCE, in yoho_ctx_create, into the context (the list is in
// include/yoho_hip.h); no entry point reads the environment afterwards, so what a context does is fixed when it is created, two
// contexts of one process can differ, and a test can pin every switch.
struct yoho_env_switches {
    bool partII_tail_staged = false;   // YOHO_PARTII_TAIL=staged: PartII's 1x1 tail as three gconv launches + quat_norm instead of mlp_head
    bool transfer_staged = false;      // YOHO_TRANSFER=staged: the feature transfer of a pass copy by copy instead of the batched grid kernels
    bool xf_steal = true;              // YOHO_XF_STEAL=0: gft16x walks its chunks by static striding instead of tickets
    int nn_splits = 0;                 // YOHO_NN_SPLITS=<n>: column splits of the matcher's Gram passes (0 = two workgroups per CU)
    int spconv_debug = 0;              // YOHO_SPCONV_DEBUG=<bits>: ablations of the fine-level sparse conv (a -DYOHO_SPCONV_ABLATE build only)
    bool fcgf_f32 = false;             // YOHO_FCGF=f32: the backbone's weights are packed for the fp32-MFMA kernels at yoho_load_fcgf
    bool fcgf_full_maps = false;       // YOHO_FCGF_MAPS=full: every kernel map by its own probes (no mirrored / inverted maps)
    bool fcgf_norm_staged = false;     // YOHO_FCGF_NORM=staged: row normalisation as its own kernel behind the last convolution
    bool fcgf_heads_staged = false;    // YOHO_FCGF_HEADS=staged: the decoder's two 1 x 1 heads as two launches (heads_fused_kernel off); identical bits
    int partII_l1_variant = 2;         // YOHO_PARTII_L1=3: PartII's first (Fourier) layer on fgemm3 (256 x 256 tiles, eight waves) instead of fgemm2
    long long ws_limit_mb = 0;         // YOHO_WS_LIMIT_MB=<n>: a workspace request above n MiB fails as an exhausted device would (0 = no limit):
                                       // lets a test walk the YOHO_ENOMEM recoveries of the backbone (hash-table attempt, table voxelisation)
};

struct yoho_ctx {
    int device = 0;
    yoho_env_switches env;
    // tables
    float* dR32 = nullptr;       // (60,9)
    double* dR64 = nullptr;      // (60,9) widened from f32 exactly as the reference's f64 @ f32 promotes
    int* dN = nullptr;           // (60,13) int32
    int* dP = nullptr;           // (60,60) int32
    unsigned* dPq = nullptr;     // P packed for des2r_kernel: word [q][a] (q < 15, a < 64) = P[a][4q .. 4q+3] as bytes (a >= 60: row 59)
    uint8_t hN[60 * 13];
    uint8_t hP[60 * 60];
    float hR[60 * 9];
    // weights
    bool has_partI = false, has_partII = false;
    yoho::Layer p1[4];           // conv_in, res_in, res_out, conv_out
    yoho::Layer p2[6];           // init, res_in, res_out, fc0, fc1, fc2
    float *p2_init_bn_s = nullptr, *p2_init_bn_t = nullptr;  // BN(128) applied by the PartII pack kernel
    int partII_mode = 2;         // cone layers: 0 fp32 MFMA, 1 bf16x3 split MFMA, 2 fp16x2 split MFMA (direct cone kernels), 3 fp16x2 with the 13-element cone
                                 // layer as an implicit GEMM (cgemm_kernel), 4 = 3 with fp8 correction products
    int cone_nslot = 0;          // group elements the middle cone layer reads (45 for the icosahedral tables), their slot (or -1) per element,
    int cone_slot_of[60];        // the slot of N[n_j][k] per (output element j, tap k), and the output elements n_j = N[0][j]
    unsigned char cone_slot[13 * 13];
    unsigned char cone_outg[13];
    int gconv_mode = 4;          // 0 direct fp32 MFMA, 1 direct bf16x3 split, 2 group-Fourier fp32 MFMA, 3 direct fp16x2 split,
                                 // 4 group-Fourier irrep GEMMs on the fp16x2 split MFMA (default); 5 / 6 its other blockings; 7 = 4 with the
                                 // correction products of the two large layers on the fp8 matrix pipe (fgemm3c, opt-in)
    yoho::FourierBasis* fb = nullptr;
    float* dFpad = nullptr;      // F padded to 64 x 64 (device)
    void* dF16 = nullptr;        // fp16x2 MFMA fragments of F^T and F (gft16.hip)
    yoho::FcgfNet* fcgf = nullptr;   // FCGF backbone weights (sparse.hip)
    int tap_inv[13] = {0};       // inv[k]: the tap whose group element is the inverse of tap k's (train.hip)
    int* d_tap_inv = nullptr;
    int nCU = 256;
    yoho::SlotTables tabs;       // direct-conv slot tables (device)
    int* d_rflag = nullptr;      // fp16 range words (note_range): [0] PartI, [1] PartII; read and cleared by yoho_range_status
    unsigned* d_amax = nullptr;  // gconv_mode 7: largest |plane value| the transforms wrote, per stream slot and layer ([2][4], float bit patterns; zeroed at the head of a pass)
    int* d_xfctr = nullptr;      // chunk tickets of the persistent transform kernel (gft16x work stealing): [stream slot 2][launch 4][2], zero between launches
    int fcgf_cell_sort = 1;      // FCGF backbone: level-0 rows grouped by 8^3-voxel cell inside the pass (gather locality): 0 never, 1 passes of >= 2^18 rows, 2 always
    int fcgf_hash_coords = 0;    // 1: coordinate maps through hash tables even when the clouds fit rank-ordered bitmaps (YOHO_FCGF_COORDS=hash, yoho_set_fcgf_sort cell_sort | 4)
    int fcgf_parity_sort = 1;    // transposed convolutions of the FCGF backbone walk parity-sorted rows (sparse.hip); 0: YOHO_FCGF_SORT=0
    int nn_prefilter = 1;        // mutual NN of large sets: MFMA pre-filter + exact candidates (matchf.hip); 0 = brute force (YOHO_NN=brute)
    double nn_cell = 0.0;        // > 0: 3-D nearest-neighbour searches go through a hash grid of this cell size (gridnn.hip)
    // workspace (grown on demand)
    yoho::Workspace ws;
    void* pair_ws = nullptr; size_t pair_ws_bytes = 0;        // yoho_register_pair: device scratch that outlives the staged calls' use of ws
    void* pair_host = nullptr; size_t pair_host_bytes = 0;    // ... and its page-locked host side (match count, vote order, result)
    // depth-first PartI schedule (default mode): the pass is cut into chunks of partI_chunk keypoints (a multiple of 256; 0 = one
    // breadth-first pass), each chunk running head -> 4 GEMMs + 3 transforms -> tail on its own slice of the workspace so that
    // the intermediates of a chunk stay in the 256 MB Infinity Cache; with partI_streams == 2 the chunks alternate between the
    // caller's stream and an internal one (forked / joined with events), so one chunk's transforms overlap the other's GEMMs
    int partI_chunk = 0;
    int partI_streams = 1;
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // profiling: EV_PER_PASS events per chunk of the last profiled PartI pass
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    int ev_chunks = 0;           // chunks of the last profiled pass
    hipEvent_t ev_pass[2] = {nullptr, nullptr};   // around the whole pass on the caller's stream
    bool ev_created = false;
    float kernel_ms[8];
    yoho::PhaseProf phase;
};
namespace yoho {
inline void phase_mark(yoho_ctx* c, int cat, hipStream_t s) {
    PhaseProf& p = c->phase;
    if (!p.on) return;
    if (p.open_cat < 0 && cat < 0) return;
    if (p.used == p.pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        p.pool.push_back(e);
    }
    hipEvent_t e = p.pool[p.used++];
    (void)hipEventRecord(e, s);
    if (p.open_cat >= 0) p.spans.push_back({p.open_cat, p.open_ev, e});
    p.open_cat = cat < PhaseProf::NCAT ? cat : -1;
    p.open_ev = e;
}
inline void phase_work(yoho_ctx* c, int cat, double flops) {
    if (c->phase.on && cat >= 0 && cat < PhaseProf::NCAT) { c->phase.flops[cat] += flops; c->phase.launches[cat] += 1.0; }
}
}  // namespace yoho
namespace yoho { constexpr int EV_PER_PASS = 11; }
