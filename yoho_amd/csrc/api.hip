// C-ABI entry points: context, checkpoint loading (BN folding + MFMA weight packing) and the
// PartI / PartII forward passes.  See include/yoho_hip.h for the contract.
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>

namespace yoho {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void clear_error() { g_err[0] = 0; }

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    return YOHO_EHIP;
}

static int upload(const void* h, size_t bytes, void** d) {
    HIPCHK(hipMalloc(d, bytes));
    HIPCHK(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return 0;
}

static void free_layer(Layer& L) {
    if (L.wp) (void)hipFree(L.wp);
    if (L.wp16) (void)hipFree(L.wp16);
    if (L.wph) (void)hipFree(L.wph);
    if (L.wpg) (void)hipFree(L.wpg);
    if (L.wpg8) (void)hipFree(L.wpg8);
    if (L.wcg) (void)hipFree(L.wcg);
    if (L.wcg8) (void)hipFree(L.wcg8);
    if (L.wpf) (void)hipFree(L.wpf);
    if (L.bias) (void)hipFree(L.bias);
    if (L.bn_s) (void)hipFree(L.bn_s);
    if (L.bn_t) (void)hipFree(L.bn_t);
    L = Layer();
}

// BatchNorm2d(eval, eps = 1e-5) as y = x*s + t
static void bn_affine(const yoho_bn_w& bn, int c, int cpad, std::vector<float>& s, std::vector<float>& t) {
    s.assign(cpad, 0.f);
    t.assign(cpad, 0.f);
    for (int i = 0; i < c; ++i) {
        const float sc = bn.gamma[i] / std::sqrt(bn.var[i] + 1e-5f);
        s[i] = sc;
        t[i] = bn.beta[i] - bn.mean[i] * sc;
    }
}

static inline unsigned bf16_rne_bits(float x) {
    unsigned u;
    std::memcpy(&u, &x, 4);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
static inline float bf16_to_float(unsigned h) {
    unsigned u = h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// 13-tap conv weight -> bf16x3 planes in MFMA 32x32x16 A-fragment order
//   [ob][c8][tp][plane][lane = 32h + i][e]:  W[ob*32+i][c8*8+e][2*tp+h], split x = hi + mid + lo (RNE each)
static int build_wp16(Layer& L, const yoho_conv_w& cw) {
    const int nob = (L.cout_pad + 63) / 64 * 2;       // the 2-block kernel variant walks pairs of o-blocks
    const int c8n = L.cin / 8;
    std::vector<unsigned short> wp((size_t)nob * c8n * 7 * 3 * 64 * 8, 0);
    for (int ob = 0; ob < nob; ++ob)
        for (int c8 = 0; c8 < c8n; ++c8)
            for (int tp = 0; tp < 7; ++tp) {
                unsigned short* dst = &wp[(((size_t)ob * c8n + c8) * 7 + tp) * 3 * 512];
                for (int lane = 0; lane < 64; ++lane) {
                    const int h = lane >> 5, o = ob * 32 + (lane & 31), tap = 2 * tp + h;
                    if (o >= L.cout || tap >= L.ntaps) continue;
                    for (int e = 0; e < 8; ++e) {
                        const float x = cw.weight[((size_t)o * L.cin + c8 * 8 + e) * L.ntaps + tap];
                        const unsigned hi = bf16_rne_bits(x);
                        const float r1 = x - bf16_to_float(hi);
                        const unsigned mi = bf16_rne_bits(r1);
                        const float r2 = r1 - bf16_to_float(mi);
                        const unsigned lo = bf16_rne_bits(r2);
                        dst[0 * 512 + lane * 8 + e] = (unsigned short)hi;
                        dst[1 * 512 + lane * 8 + e] = (unsigned short)mi;
                        dst[2 * 512 + lane * 8 + e] = (unsigned short)lo;
                    }
                }
            }
    return upload(wp.data(), wp.size() * sizeof(unsigned short), &L.wp16);
}

static inline unsigned short half_bits(float x) {
    const _Float16 h = (_Float16)x;       // round to nearest even
    unsigned short u;
    std::memcpy(&u, &h, 2);
    return u;
}

// 13-tap conv weight -> fp16x2 planes, same fragment order with 2 planes: x * 2^k = hi + lo.  2^k puts max|w| into
// [2^9, 2^10) so that the lo plane of every weight that matters stays a normal fp16 number.
static int build_wph(Layer& L, const yoho_conv_w& cw) {
    const int nob = (L.cout_pad + 63) / 64 * 2;
    const int c8n = L.cin / 8;
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)L.cout * L.cin * L.ntaps; ++i) wmax = std::fmax(wmax, std::fabs(cw.weight[i]));
    int ex = 0;
    if (wmax > 0.f && std::isfinite(wmax)) { (void)std::frexp(wmax, &ex); }      // wmax = f * 2^ex, f in [0.5, 1)
    const float wscale = std::ldexp(1.f, 10 - ex);
    L.wph_descale = 1.f / (wscale * H2_ASCALE);
    std::vector<unsigned short> wp((size_t)nob * c8n * 7 * 2 * 64 * 8, 0);
    for (int ob = 0; ob < nob; ++ob)
        for (int c8 = 0; c8 < c8n; ++c8)
            for (int tp = 0; tp < 7; ++tp) {
                unsigned short* dst = &wp[(((size_t)ob * c8n + c8) * 7 + tp) * 2 * 512];
                for (int lane = 0; lane < 64; ++lane) {
                    const int h = lane >> 5, o = ob * 32 + (lane & 31), tap = 2 * tp + h;
                    if (o >= L.cout || tap >= L.ntaps) continue;
                    for (int e = 0; e < 8; ++e) {
                        const float x = cw.weight[((size_t)o * L.cin + c8 * 8 + e) * L.ntaps + tap] * wscale;
                        const _Float16 hi = (_Float16)x;
                        dst[0 * 512 + lane * 8 + e] = half_bits(x);
                        dst[1 * 512 + lane * 8 + e] = half_bits(x - (float)hi);
                    }
                }
            }
    return upload(wp.data(), wp.size() * sizeof(unsigned short), &L.wph);
}

// conv weight (cout,cin,1,ntaps) -> A-fragment order [ob][c8][tap][lane = h*32+i][s]:
//   value = W[ob*32 + i][c8*8 + 4h + s][tap]
static int build_layer(Layer& L, const yoho_conv_w& cw, int cin, int cout, int ntaps, const yoho_bn_w* bn_after, const FourierBasis* fb = nullptr) {
    free_layer(L);
    if (!cw.weight || !cw.bias) { set_error("null conv weight/bias pointer"); return YOHO_EINVAL; }
    L.cin = cin; L.cout = cout; L.ntaps = ntaps;
    L.cout_pad = (cout + 31) / 32 * 32;
    const int nob = L.cout_pad / 32, c8n = cin / 8;
    std::vector<float> wp((size_t)nob * c8n * ntaps * 256, 0.f);
    for (int ob = 0; ob < nob; ++ob)
        for (int c8 = 0; c8 < c8n; ++c8)
            for (int tap = 0; tap < ntaps; ++tap) {
                float* dst = &wp[(((size_t)ob * c8n + c8) * ntaps + tap) * 256];
                for (int lane = 0; lane < 64; ++lane) {
                    const int h = lane >> 5, o = ob * 32 + (lane & 31);
                    if (o >= cout) continue;
                    for (int s = 0; s < 4; ++s) {
                        const int c = c8 * 8 + 4 * h + s;
                        dst[lane * 4 + s] = cw.weight[((size_t)o * cin + c) * ntaps + tap];
                    }
                }
            }
    const int cpad64 = (L.cout_pad + 63) / 64 * 64;      // the bf16x3 kernel touches o-blocks in pairs
    std::vector<float> bias(cpad64, 0.f);
    std::memcpy(bias.data(), cw.bias, sizeof(float) * cout);
    int rc;
    if ((rc = upload(wp.data(), wp.size() * sizeof(float), (void**)&L.wp))) return rc;
    if (ntaps == NTAP && (rc = build_wp16(L, cw))) return rc;
    if (ntaps == NTAP && (rc = build_wph(L, cw))) return rc;
    if (ntaps == NTAP && fb) {
        std::vector<float> wf;
        pack_fourier_weights(*fb, cw.weight, cin, cout, L.cout_pad, wf);
        if ((rc = upload(wf.data(), wf.size() * sizeof(float), (void**)&L.wpf))) return rc;
        if ((cout % 256 == 0 || cout == 32) && cin % 32 == 0) {
            std::vector<unsigned short> wg, wg8;
            const bool big = cin >= 256 && cout >= 256;                  // the layers fgemm3c (gconv_mode 7) takes
            if (pack_fgemm_weights(*fb, cw.weight, cin, cout, wg, &L.wpg_descale, big ? &wg8 : nullptr)) { set_error("irrep-GEMM weight packing failed"); return YOHO_EINVAL; }
            if ((rc = upload(wg.data(), wg.size() * sizeof(unsigned short), &L.wpg))) return rc;
            if (big && (rc = upload(wg8.data(), wg8.size() * sizeof(unsigned short), &L.wpg8))) return rc;
        }
    }
    if ((rc = upload(bias.data(), bias.size() * sizeof(float), (void**)&L.bias))) return rc;
    if (bn_after) {
        if (!bn_after->gamma || !bn_after->beta || !bn_after->mean || !bn_after->var) {
            set_error("null batch-norm pointer"); return YOHO_EINVAL;
        }
        std::vector<float> s, t;
        bn_affine(*bn_after, cout, cpad64, s, t);
        if ((rc = upload(s.data(), s.size() * sizeof(float), (void**)&L.bn_s))) return rc;
        if ((rc = upload(t.data(), t.size() * sizeof(float), (void**)&L.bn_t))) return rc;
    }
    return 0;
}

int ensure_ws(yoho_ctx* ctx, size_t bytes, hipStream_t s) {
    if (ctx->ws.bytes >= bytes) return 0;
    if (ctx->ws.p) {
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipFree(ctx->ws.p));
        ctx->ws.p = nullptr; ctx->ws.bytes = 0;
    }
    const size_t want = bytes + bytes / 8;
    hipError_t e = hipSuccess;
    if (ctx->env.ws_limit_mb > 0 && want > (size_t)ctx->env.ws_limit_mb << 20) e = hipErrorOutOfMemory;      // YOHO_WS_LIMIT_MB (tests)
    else e = hipMalloc(&ctx->ws.p, want);
    if (e != hipSuccess) {
        // the runtime keeps a failed call as the thread's last error: callers that RECOVER from YOHO_ENOMEM (the backbone's attempts)
        // check hipGetLastError() behind their next launches and must not find this one there
        (void)hipGetLastError();
        ctx->ws.p = nullptr;
        set_error("workspace allocation of %zu bytes failed: %s", want, hipGetErrorString(e));
        return YOHO_ENOMEM;
    }
    ctx->ws.bytes = want;
    return 0;
}

static ConvArgs conv_args(const Layer& L, const float* X, int nTiles, const float* res, float* out_raw, float* out_act, bool osplit) {
    ConvArgs a;
    a.X = X; a.Wp = L.wp; a.bias = L.bias; a.bn_s = L.bn_s; a.bn_t = L.bn_t;
    a.res = res; a.out_raw = out_raw; a.out_act = out_act;
    a.nTiles = nTiles; a.cin8 = L.cin / 8; a.cout8 = L.cout_pad / 8;
    a.nOB = osplit ? L.cout_pad / 128 : L.cout_pad / 32;
    a.ntaps = L.ntaps;
    a.slabtab = L.tabs ? L.tabs->slabtab : nullptr; a.outg = L.tabs ? L.tabs->outg : nullptr;
    return a;
}

}  // namespace yoho

using namespace yoho;

namespace yoho {
__global__ void clock_probe_kernel(long long* out, long long wall_ticks, long long wall_khz) {
    if (threadIdx.x != 0) return;
    const long long w0 = wall_clock64(), c0 = clock64();
    long long w1 = w0;
    while (w1 - w0 < wall_ticks) { __builtin_amdgcn_s_sleep(16); w1 = wall_clock64(); }
    const long long c1 = clock64();
    out[0] = c1 - c0; out[1] = w1 - w0; out[2] = wall_khz;
}
}  // namespace yoho

extern "C" {

int yoho_clock_probe(yoho_ctx* c, int microseconds, long long* out3, void* stream) {
    if (!c || !out3 || microseconds < 1 || microseconds > 1000000) { set_error("yoho_clock_probe: bad argument"); return YOHO_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int khz = 0;
    HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device));
    if (khz <= 0) { set_error("yoho_clock_probe: the device reports no wall-clock rate"); return YOHO_EHIP; }
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out3, (long long)microseconds * khz / 1000, (long long)khz);
    HIPCHK(hipGetLastError());
    return 0;
}

const char* yoho_last_error(void) { return g_err; }
const char* yoho_version(void) { return "yoho_hip 0.1.0 (gfx950)"; }

int yoho_ctx_create(int device, const float* R, const uint8_t* N, const uint8_t* P, yoho_ctx** out) {
    if (!R || !N || !P || !out) { set_error("yoho_ctx_create: null argument"); return YOHO_EINVAL; }
    for (int i = 0; i < G * NTAP; ++i) if (N[i] >= G) { set_error("Nei table entry out of range"); return YOHO_EINVAL; }
    for (int i = 0; i < G * G; ++i) if (P[i] >= G) { set_error("60_60 table entry out of range"); return YOHO_EINVAL; }
    HIPCHK(hipSetDevice(device));
    yoho_ctx* c = new yoho_ctx();
    c->device = device;
    std::memcpy(c->hN, N, sizeof(c->hN));
    std::memcpy(c->hP, P, sizeof(c->hP));
    std::memcpy(c->hR, R, sizeof(c->hR));
    std::vector<int> n32(G * NTAP), p32(G * G);
    std::vector<double> r64(G * 9);
    for (int i = 0; i < G * NTAP; ++i) n32[i] = N[i];
    for (int i = 0; i < G * G; ++i) p32[i] = P[i];
    for (int i = 0; i < G * 9; ++i) r64[i] = (double)R[i];
    int rc;
    if ((rc = upload(R, sizeof(float) * G * 9, (void**)&c->dR32)) || (rc = upload(r64.data(), sizeof(double) * G * 9, (void**)&c->dR64)) ||
        (rc = upload(n32.data(), sizeof(int) * G * NTAP, (void**)&c->dN)) || (rc = upload(p32.data(), sizeof(int) * G * G, (void**)&c->dP))) {
        delete c; return rc;
    }
    {
        std::vector<unsigned> pq(15 * 64);
        for (int q = 0; q < 15; ++q)
            for (int a = 0; a < 64; ++a) {
                const int* pr = p32.data() + (a < G ? a : G - 1) * G + 4 * q;
                pq[q * 64 + a] = (unsigned)pr[0] | ((unsigned)pr[1] << 8) | ((unsigned)pr[2] << 16) | ((unsigned)pr[3] << 24);
            }
        if ((rc = upload(pq.data(), sizeof(unsigned) * pq.size(), (void**)&c->dPq))) { delete c; return rc; }
    }
    // tap inversion table for the data gradient of the group conv (train.hip): n_k = N[e][k] with e the identity,
    // inv[k] = k2 with n_k2 * n_k = e, i.e. N[n_k][k2] == e.  The tap set must be closed under inversion.
    {
        int e = -1;
        for (int g = 0; g < G && e < 0; ++g) {
            float d = 0.f;
            for (int i = 0; i < 9; ++i) d += std::fabs(R[g * 9 + i] - ((i % 4 == 0) ? 1.f : 0.f));
            if (d < 1e-4f) e = g;
        }
        bool ok = e >= 0;
        for (int k = 0; k < NTAP && ok; ++k) {
            const int nk = N[e * NTAP + k];
            int k2 = -1;
            for (int j = 0; j < NTAP; ++j) if (N[nk * NTAP + j] == e) k2 = j;
            if (k2 < 0) ok = false; else c->tap_inv[k] = k2;
        }
        if (!ok) for (int k = 0; k < NTAP; ++k) c->tap_inv[k] = -1;          // yoho_gconv_layer(transpose) then refuses
        if ((rc = upload(c->tap_inv, sizeof(int) * NTAP, (void**)&c->d_tap_inv))) { delete c; return rc; }
    }
    // slot tables: which output group elements each configuration computes
    std::vector<int> slab(NCFG * NTAP * G, 0), outg(NCFG * G, -1);
    auto fill = [&](int cfg, const std::vector<int>& glist, int gpw, bool replicate) {
        const int nslots = 4 * gpw;
        for (int sl = 0; sl < nslots; ++sl) {
            int g = -1;
            if (replicate) g = glist[0];
            else if (sl < (int)glist.size()) g = glist[sl];
            outg[cfg * G + sl] = g;
        }
        // kernel indexes c_slabtab[cfg][(tap*4 + ws)*GPW + j] = [tap][slot] with row length 4*GPW
        for (int tap = 0; tap < NTAP; ++tap)
            for (int sl = 0; sl < nslots; ++sl) {
                const int g = outg[cfg * G + sl];
                slab[cfg * NTAP * G + tap * nslots + sl] = (g < 0 ? 0 : (int)N[g * NTAP + tap]) * 1024;
            }
    };
    std::vector<int> all(G); for (int i = 0; i < G; ++i) all[i] = i;
    std::vector<int> one(NTAP); for (int k = 0; k < NTAP; ++k) one[k] = N[k];          // N[0][k]
    std::vector<bool> in2(G, false);
    for (int k = 0; k < NTAP; ++k) for (int k2 = 0; k2 < NTAP; ++k2) in2[N[one[k] * NTAP + k2]] = true;
    std::vector<int> two; for (int g = 0; g < G; ++g) if (in2[g]) two.push_back(g);
    if (two.size() > 48) { set_error("2-hop cone of group element 0 has %zu > 48 elements", two.size()); delete c; return YOHO_EINVAL; }
    fill(CFG_FULL, all, 15, false);
    fill(CFG_C45, two, 12, false);
    fill(CFG_C13, one, 4, false);
    fill(CFG_C1, std::vector<int>{0}, 1, true);
    if ((rc = upload_slot_tables(slab.data(), outg.data(), c->tabs)) || (rc = gconv_init())) { delete c; return rc; }
    // bf16x3 variant: unit u = output group elements (2u, 2u+1); tap pair tp = taps (2tp, 2tp+1), tap 13 = zero weights
    {
        std::vector<int> slab4(3 * 7 * 32, 0), unitg(3 * 32 * 2, -1);
        const std::vector<int>* lists[3] = {&all, &two, &one};       // cfg 0: 60 outputs, 1: 45-cone, 2: 13-cone
        for (int cfg = 0; cfg < 3; ++cfg) {
            const std::vector<int>& gl = *lists[cfg];
            int* ug = &unitg[cfg * 64];
            for (size_t k = 0; k < gl.size(); ++k) ug[k] = gl[k];      // unit u = (gl[2u], gl[2u+1]); a trailing -1 = unused half
            for (int tp = 0; tp < 7; ++tp)
                for (int u = 0; u < 32; ++u)
                    for (int gs = 0; gs < 2; ++gs)
                        for (int h = 0; h < 2; ++h) {
                            int g = ug[2 * u + gs];
                            if (g < 0) g = ug[2 * u];                   // unused half: read the partner's (valid, finite) slabs
                            int tap = 2 * tp + h;
                            if (tap >= NTAP) tap = 2 * tp;              // weights are zero there; any finite slab will do
                            slab4[(cfg * 7 + tp) * 32 + u] |= (g < 0 ? 0 : (int)N[g * NTAP + tap]) << (8 * (2 * gs + h));
                        }
        }
        if ((rc = upload_slot_tables16(slab4.data(), unitg.data(), c->tabs)) || (rc = gconv16_init())) { delete c; return rc; }
    }
    c->gconv_mode = 4;      // default: group-Fourier irrep GEMMs on the fp16x2 split MFMA; YOHO_GCONV=f32 | bf16x3 | fourier | fp16x2 | fgemm
    c->partII_mode = 2;     // default: fp16x2 cone layers; YOHO_PARTII=f32 | bf16x3 | fp16x2
    // ---- the environment is read HERE and nowhere else (include/yoho_hip.h lists the variables)
    {
        auto is = [](const char* name, const char* val) { const char* e = std::getenv(name); return e && std::strcmp(e, val) == 0; };
        auto num = [](const char* name) { const char* e = std::getenv(name); return e ? std::atoi(e) : 0; };
        c->env.partII_tail_staged = is("YOHO_PARTII_TAIL", "staged");
        c->env.transfer_staged = is("YOHO_TRANSFER", "staged");
        c->env.xf_steal = !is("YOHO_XF_STEAL", "0");
        c->env.nn_splits = num("YOHO_NN_SPLITS");
        c->env.spconv_debug = num("YOHO_SPCONV_DEBUG");
        c->env.fcgf_f32 = is("YOHO_FCGF", "f32");
        c->env.fcgf_full_maps = is("YOHO_FCGF_MAPS", "full");
        c->env.fcgf_norm_staged = is("YOHO_FCGF_NORM", "staged");
        c->env.fcgf_heads_staged = is("YOHO_FCGF_HEADS", "staged");
        if (const char* e = std::getenv("YOHO_WS_LIMIT_MB")) c->env.ws_limit_mb = std::atoll(e);
        if (is("YOHO_PARTII_L1", "3")) c->env.partII_l1_variant = 3;
    }
    if (const char* m = std::getenv("YOHO_FCGF_CELLS")) c->fcgf_cell_sort = std::atoi(m);
    if (const char* m = std::getenv("YOHO_FCGF_SORT")) c->fcgf_parity_sort = std::strcmp(m, "0") == 0 ? 0 : 1;
    if (const char* m = std::getenv("YOHO_FCGF_COORDS")) c->fcgf_hash_coords = std::strcmp(m, "hash") == 0 ? 1 : 0;
    if (const char* m = std::getenv("YOHO_NN")) c->nn_prefilter = std::strcmp(m, "brute") == 0 ? 0 : 1;
    if (const char* m = std::getenv("YOHO_PARTII")) c->partII_mode = std::strcmp(m, "f32") == 0 ? 0 : (std::strcmp(m, "fp16x2") == 0 ? 2 : (std::strcmp(m, "cgemm") == 0 ? 3 : (std::strcmp(m, "cgemm8") == 0 ? 4 : 1)));
    if (const char* m = std::getenv("YOHO_PARTI_CHUNK")) {       // "1024" or "1024x2" (chunk keypoints x streams), see yoho_set_partI_schedule
        int ck = 0, ns = 1;
        if (std::sscanf(m, "%dx%d", &ck, &ns) >= 1 && ck >= 0 && (ns == 1 || ns == 2)) { c->partI_chunk = (ck + 255) / 256 * 256; c->partI_streams = ns; }
    }
    if (const char* m = std::getenv("YOHO_GCONV")) c->gconv_mode = std::strcmp(m, "f32") == 0 ? 0 : (std::strcmp(m, "bf16x3") == 0 ? 1 : (std::strcmp(m, "fp16x2") == 0 ? 3 : (std::strcmp(m, "fgemm") == 0 ? 4 : (std::strcmp(m, "fgemm256") == 0 ? 5 : (std::strcmp(m, "fgemm128") == 0 ? 6 : (std::strcmp(m, "fgemm8") == 0 ? 7 : 2))))));
    // group-Fourier basis (irreps of the table's group)
    c->fb = new FourierBasis();
    if ((rc = build_fourier(N, P, *c->fb))) { delete c->fb; delete c; return rc; }
    {
        std::vector<float> fpad(2 * 64 * 64, 0.f);             // F padded to 64x64, then its transpose
        for (int s = 0; s < G; ++s)
            for (int g = 0; g < G; ++g) {
                fpad[s * 64 + g] = (float)c->fb->F[s * G + g];
                fpad[64 * 64 + g * 64 + s] = (float)c->fb->F[s * G + g];
            }
        if ((rc = upload(fpad.data(), fpad.size() * sizeof(float), (void**)&c->dFpad)) || (rc = gft_init()) || (rc = fgemm_init()) || (rc = gft16_init())) { delete c->fb; delete c; return rc; }
        std::vector<unsigned short> f16;
        build_gft16_frags(*c->fb, f16);
        if ((rc = upload(f16.data(), f16.size() * sizeof(unsigned short), &c->dF16))) { delete c->fb; delete c; return rc; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->nCU = prop.multiProcessorCount;
    }
    HIPCHK(hipMalloc((void**)&c->d_rflag, 4 * sizeof(int)));
    HIPCHK(hipMemset(c->d_rflag, 0, 4 * sizeof(int)));
    HIPCHK(hipMalloc((void**)&c->d_amax, 12 * sizeof(unsigned)));      // [2][4] PartI, [8] PartII's cone GEMM
    HIPCHK(hipMemset(c->d_amax, 0, 12 * sizeof(unsigned)));
    {
        // receptive cone of group element 0 through two 13-tap layers: outputs of the middle layer at n_j = N[0][j], inputs at N[n_j][k]
        for (int g = 0; g < G; ++g) c->cone_slot_of[g] = -1;
        for (int j = 0; j < NTAP; ++j)
            for (int k = 0; k < NTAP; ++k) c->cone_slot_of[c->hN[c->hN[j] * NTAP + k]] = 0;
        int ns = 0;
        for (int g = 0; g < G; ++g) if (c->cone_slot_of[g] == 0) c->cone_slot_of[g] = ns++;
        c->cone_nslot = ns;
        for (int j = 0; j < NTAP; ++j) {
            c->cone_outg[j] = c->hN[j];
            for (int k = 0; k < NTAP; ++k) c->cone_slot[j * 13 + k] = (unsigned char)c->cone_slot_of[c->hN[c->hN[j] * NTAP + k]];
        }
    }
    HIPCHK(hipMalloc((void**)&c->d_xfctr, 16 * sizeof(int)));
    HIPCHK(hipMemset(c->d_xfctr, 0, 16 * sizeof(int)));
    *out = c;
    return 0;
}

int yoho_ctx_destroy(yoho_ctx* c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& L : c->p1) free_layer(L);
    for (auto& L : c->p2) free_layer(L);
    if (c->p2_init_bn_s) (void)hipFree(c->p2_init_bn_s);
    if (c->p2_init_bn_t) (void)hipFree(c->p2_init_bn_t);
    if (c->ws.p) (void)hipFree(c->ws.p);
    if (c->dR32) (void)hipFree(c->dR32);
    if (c->dR64) (void)hipFree(c->dR64);
    if (c->dN) (void)hipFree(c->dN);
    if (c->dP) (void)hipFree(c->dP);
    if (c->dPq) (void)hipFree(c->dPq);
    if (c->pair_ws) (void)hipFree(c->pair_ws);
    if (c->pair_host) (void)hipHostFree(c->pair_host);
    if (c->dFpad) (void)hipFree(c->dFpad);
    if (c->dF16) (void)hipFree(c->dF16);
    if (c->fcgf) fcgf_free(c->fcgf);
    if (c->d_tap_inv) (void)hipFree(c->d_tap_inv);
    if (c->d_rflag) (void)hipFree(c->d_rflag);
    if (c->d_xfctr) (void)hipFree(c->d_xfctr);
    if (c->d_amax) (void)hipFree(c->d_amax);
    for (int* t : {c->tabs.slabtab, c->tabs.outg, c->tabs.slab4, c->tabs.unitg}) if (t) (void)hipFree(t);
    delete c->fb;
    for (auto& e : c->ev) (void)hipEventDestroy(e);
    for (auto& e : c->ev_pass) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->phase.pool) (void)hipEventDestroy(e);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    delete c;
    return 0;
}

int yoho_load_partI(yoho_ctx* c, const yoho_partI_weights* w) {
    if (!c || !w) { set_error("yoho_load_partI: null argument"); return YOHO_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    c->has_partI = false;
    int rc;
    // each layer carries the BN+ReLU that precedes the NEXT conv as its epilogue
    if ((rc = build_layer(c->p1[0], w->conv_in, 32, 256, NTAP, &w->res_in_bn, c->fb))) return rc;
    if ((rc = build_layer(c->p1[1], w->res_in, 256, 512, NTAP, &w->res_out_bn, c->fb))) return rc;
    if ((rc = build_layer(c->p1[2], w->res_out, 512, 256, NTAP, &w->out_bn, c->fb))) return rc;
    if ((rc = build_layer(c->p1[3], w->conv_out, 256, 32, NTAP, nullptr, c->fb))) return rc;
    for (auto& L : c->p1) L.tabs = &c->tabs;
    c->has_partI = true;
    return 0;
}

int yoho_load_partII(yoho_ctx* c, const yoho_partII_weights* w) {
    if (!c || !w) { set_error("yoho_load_partII: null argument"); return YOHO_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    c->has_partII = false;
    int rc;
    std::vector<float> s, t;
    if (!w->init_bn.gamma || !w->init_bn.beta || !w->init_bn.mean || !w->init_bn.var) { set_error("null batch-norm pointer"); return YOHO_EINVAL; }
    bn_affine(w->init_bn, 128, 128, s, t);
    if (c->p2_init_bn_s) { (void)hipFree(c->p2_init_bn_s); c->p2_init_bn_s = nullptr; }
    if (c->p2_init_bn_t) { (void)hipFree(c->p2_init_bn_t); c->p2_init_bn_t = nullptr; }
    if ((rc = upload(s.data(), 128 * sizeof(float), (void**)&c->p2_init_bn_s))) return rc;
    if ((rc = upload(t.data(), 128 * sizeof(float), (void**)&c->p2_init_bn_t))) return rc;
    if ((rc = build_layer(c->p2[0], w->init, 128, 256, NTAP, &w->res_in_bn, c->fb))) return rc;      // + irrep-GEMM weights
    if ((rc = build_layer(c->p2[1], w->res_in, 256, 512, NTAP, &w->res_out_bn))) return rc;
    {
        // the cone layer's weights once more as the A operand of the cone GEMM (PartII modes 3 / 4)
        std::vector<unsigned short> wc, wc8;
        if (pack_cgemm_weights(w->res_in.weight, 256, 512, NTAP, wc, &c->p2[1].wcg_descale, wc8)) { set_error("cone-GEMM weight packing failed"); return YOHO_EINVAL; }
        if ((rc = upload(wc.data(), wc.size() * sizeof(unsigned short), &c->p2[1].wcg))) return rc;
        if ((rc = upload(wc8.data(), wc8.size() * sizeof(unsigned short), &c->p2[1].wcg8))) return rc;
    }
    if ((rc = build_layer(c->p2[2], w->res_out, 512, 256, NTAP, nullptr))) return rc;
    if ((rc = build_layer(c->p2[3], w->fc0, 256, 512, 1, &w->fc0_bn))) return rc;
    if ((rc = build_layer(c->p2[4], w->fc1, 512, 128, 1, &w->fc1_bn))) return rc;
    if ((rc = build_layer(c->p2[5], w->fc2, 128, 4, 1, nullptr))) return rc;
    for (auto& L : c->p2) L.tabs = &c->tabs;
    c->has_partII = true;
    return 0;
}

int yoho_set_gconv_mode(yoho_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 7) { set_error("yoho_set_gconv_mode: mode must be 0 (fp32 MFMA), 1 (bf16x3 MFMA), 2 (group-Fourier fp32 MFMA), 3 (fp16x2 MFMA), 4 (group-Fourier irrep GEMMs, fp16x2 MFMA), 5 or 6 (4 with the other two GEMM blockings), 7 (4 with fp8 correction products)"); return YOHO_EINVAL; }
    c->gconv_mode = mode;
    return 0;
}

int yoho_set_partI_schedule(yoho_ctx* c, int chunk_kp, int streams) {
    if (!c || chunk_kp < 0 || streams < 1 || streams > 2) { set_error("yoho_set_partI_schedule: chunk_kp >= 0 and streams 1 or 2"); return YOHO_EINVAL; }
    c->partI_chunk = (chunk_kp + 255) / 256 * 256;
    c->partI_streams = streams;
    return 0;
}

int yoho_set_partII_mode(yoho_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 4) { set_error("yoho_set_partII_mode: mode must be 0 (fp32 MFMA), 1 (bf16x3), 2 (fp16x2 MFMA, direct cone kernels), 3 (fp16x2, 13-element cone layer as an implicit GEMM) or 4 (3 with fp8 correction products)"); return YOHO_EINVAL; }
    c->partII_mode = mode;
    return 0;
}

int yoho_set_nn_prefilter(yoho_ctx* c, int enable) {
    if (!c) { set_error("yoho_set_nn_prefilter: null ctx"); return YOHO_EINVAL; }
    c->nn_prefilter = enable ? 1 : 0;
    return 0;
}

int yoho_set_fcgf_sort(yoho_ctx* c, int parity_sort, int cell_sort) {
    if (!c) { set_error("yoho_set_fcgf_sort: null ctx"); return YOHO_EINVAL; }
    c->fcgf_parity_sort = parity_sort ? 1 : 0;
    c->fcgf_hash_coords = (cell_sort >= 0 && (cell_sort & 4)) ? 1 : 0;
    cell_sort = cell_sort < 0 ? 0 : (cell_sort & 3);
    c->fcgf_cell_sort = cell_sort > 2 ? 2 : cell_sort;
    return 0;
}

int yoho_set_nn_grid(yoho_ctx* c, double cell) {
    if (!c || !(cell >= 0.0) || !std::isfinite(cell)) { set_error("yoho_set_nn_grid: cell must be a finite number >= 0 (0 = brute force)"); return YOHO_EINVAL; }
    c->nn_cell = cell;
    return 0;
}

int yoho_range_status(yoho_ctx* c, int* partI_overflow, int* partII_overflow, void* stream) {
    if (!c) { set_error("yoho_range_status: null ctx"); return YOHO_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    int h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, c->d_rflag, sizeof(h), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    // only a word the caller asked for is reported and cleared: a flag nobody has looked at yet stays pending
    if (!partI_overflow) h[0] = 0;
    if (!partII_overflow) h[1] = 0;
    if (h[0]) HIPCHK(hipMemsetAsync(c->d_rflag, 0, sizeof(int), s));
    if (h[1]) HIPCHK(hipMemsetAsync(c->d_rflag + 1, 0, sizeof(int), s));
    if (partI_overflow) *partI_overflow = h[0];
    if (partII_overflow) *partII_overflow = h[1];
    if (h[0] || h[1]) {
        set_error("fp16 range exceeded in the fp16x2 arithmetic of %s%s%s since the last check: repeat the pass in mode 1 (bf16x3) or 0 (f32)",
                  h[0] ? "PartI" : "", (h[0] && h[1]) ? " and " : "", h[1] ? "PartII" : "");
        return YOHO_ERANGE;
    }
    return 0;
}

// events of a profiled pass: EV_PER_PASS per chunk, grown on demand
static int ensure_events(yoho_ctx* c, int nchunks) {
    while ((int)c->ev.size() < nchunks * EV_PER_PASS) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        c->ev.push_back(e);
    }
    for (auto& e : c->ev_pass) if (!e) HIPCHK(hipEventCreate(&e));
    c->ev_created = true;
    return 0;
}

int yoho_set_profiling(yoho_ctx* c, int enable) {
    if (!c) { set_error("null ctx"); return YOHO_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (enable) { int rc = ensure_events(c, 1); if (rc) return rc; }
    c->profiling = enable != 0;
    c->ev_chunks = 0;
    for (auto& m : c->kernel_ms) m = -1.f;
    return 0;
}

// Phase profile of the multi-launch entries (PhaseProf in common.h; categories in include/yoho_hip.h).
int yoho_phase_profile(yoho_ctx* c, int enable) {
    if (!c) { set_error("null ctx"); return YOHO_EINVAL; }
    PhaseProf& p = c->phase;
    p.on = enable != 0;
    p.used = 0; p.spans.clear(); p.open_cat = -1;
    for (int i = 0; i < PhaseProf::NCAT; ++i) { p.flops[i] = 0; p.launches[i] = 0; }
    return 0;
}

int yoho_phase_read(yoho_ctx* c, double* ms, double* flops, double* launches, void* stream) {
    if (!c || !ms) { set_error("yoho_phase_read: bad argument"); return YOHO_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    PhaseProf& p = c->phase;
    phase_mark(c, -1, (hipStream_t)stream);                   // ends a span left open
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    for (int i = 0; i < PhaseProf::NCAT; ++i) ms[i] = 0.0;
    for (const auto& sp : p.spans) {
        float d = 0.f;
        HIPCHK(hipEventSynchronize(sp.b));
        HIPCHK(hipEventElapsedTime(&d, sp.a, sp.b));
        ms[sp.cat] += d;
    }
    for (int i = 0; i < PhaseProf::NCAT; ++i) {
        if (flops) flops[i] = p.flops[i];
        if (launches) launches[i] = p.launches[i];
        p.flops[i] = 0; p.launches[i] = 0;
    }
    p.used = 0; p.spans.clear(); p.open_cat = -1;
    return 0;
}

// which: 0..3 = PartI group-conv launches (conv_in, res_in, res_out, conv_out) of the last profiled
// yoho_partI_forward pass; 4 = head (pack [+ forward transform]), 5 = tail ([inverse transform +] finalize),
// 6 = the inter-layer transform kernels together (group-Fourier modes), 7 / 8 / 9 = each of them, 10 = the inverse
// transform of the tail, 11 = finalize alone, 12 = the whole pass (first launch to last, on the caller's stream).
// A chunked pass (yoho_set_partI_schedule) reports the sums over its chunks.  Synchronises on the recorded events.
int yoho_get_kernel_ms(yoho_ctx* c, int which, float* ms) {
    if (!c || !ms || which < 0 || which > 12) { set_error("yoho_get_kernel_ms: bad argument"); return YOHO_EINVAL; }
    if (!c->ev_created || c->ev_chunks < 1) { set_error("no profiled PartI pass (yoho_set_profiling, then yoho_partI_forward)"); return YOHO_EINVAL; }
    // event order: e0 head e1 conv0 e2 xf e3 conv1 e4 xf e5 conv2 e6 xf e7 conv3 e8 inverse transform e10 finalize e9
    static const int first[6] = {1, 3, 5, 7, 0, 8};
    HIPCHK(hipEventSynchronize(c->ev_pass[1]));
    if (which == 12) { HIPCHK(hipEventElapsedTime(ms, c->ev_pass[0], c->ev_pass[1])); return 0; }
    float tot = 0.f;
    for (int k = 0; k < c->ev_chunks; ++k) {
        hipEvent_t* ev = c->ev.data() + (size_t)k * EV_PER_PASS;
        auto el = [&](int a, int b) -> int { float d = 0.f; HIPCHK(hipEventElapsedTime(&d, ev[a], ev[b])); tot += d; return 0; };
        int rc = 0;
        if (which >= 7 && which <= 9) rc = el(2 * (which - 6), 2 * (which - 6) + 1);
        else if (which == 10) rc = el(8, 10);
        else if (which == 11) rc = el(10, 9);
        else if (which == 5) rc = el(8, 9);
        else if (which == 6) { for (int i = 2; i <= 6 && !rc; i += 2) rc = el(i, i + 1); }      // the three inter-layer transform kernels (group-Fourier modes; ~0 otherwise)
        else rc = el(first[which], first[which] + 1);
        if (rc) return rc;
    }
    *ms = tot;
    return 0;
}

// split-precision variants (npl = 3: bf16x3, npl = 2: fp16x2): 16-keypoint tiles, 16-bit plane activations, fp32 raw
// residual / output
static int partI_pass16(yoho_ctx* c, const float* x, int B, float* eqv, float* inv, float* inv_np, hipStream_t s, int npl) {
    const int nT = (B + 15) / 16;
    const size_t chunk = (size_t)npl * 15360, rawslabs = (size_t)G * 128 * sizeof(float);      // bytes per (tile, c8)
    const size_t szX = (size_t)nT * 4 * chunk, szA = (size_t)nT * 32 * chunk, szA1 = (size_t)nT * 64 * chunk;
    const size_t szH0 = (size_t)nT * 32 * rawslabs, szY = (size_t)nT * 8 * rawslabs;   // Y padded to 64 channels
    int rc;
    if ((rc = ensure_ws(c, szX + szA + szA1 + szH0 + szY, s))) return rc;
    char* bX = (char*)c->ws.p;
    char* bA = bX + szX;
    char* bA1 = bA + szA;
    float* bH0 = (float*)(bA1 + szA1);
    float* bY = (float*)((char*)bH0 + szH0);
    const bool prof = c->profiling && c->ev_created;
    auto mark = [&](int i) { if (prof) (void)hipEventRecord(c->ev[i], s); };
    if (prof) { c->ev_chunks = 1; (void)hipEventRecord(c->ev_pass[0], s); }
    mark(0);
    int* rf = c->d_rflag;
    if ((rc = launch_pack16_partI(x, B, nT, bX, s, npl, rf))) return rc;
    mark(1);
    if ((rc = launch_gconv16(c->p1[0], bX, nT, nullptr, bH0, bA, EPI_RAW | EPI_ACT, s, 0, nullptr, nullptr, npl, rf))) return rc;
    mark(2); mark(3);
    if ((rc = launch_gconv16(c->p1[1], bA, nT, nullptr, nullptr, bA1, EPI_ACT, s, 0, nullptr, nullptr, npl, rf))) return rc;
    mark(4); mark(5);
    if ((rc = launch_gconv16(c->p1[2], bA1, nT, bH0, nullptr, bA, EPI_RES | EPI_ACT, s, 0, nullptr, nullptr, npl, rf))) return rc;
    mark(6); mark(7);
    if ((rc = launch_gconv16(c->p1[3], bA, nT, nullptr, bY, nullptr, EPI_RAW, s, 0, nullptr, nullptr, npl, rf))) return rc;
    mark(8); mark(10);
    if ((rc = launch_finalize_partI(bY, x, B, eqv, inv, inv_np, 1, s))) return rc;
    mark(9);
    if (prof) (void)hipEventRecord(c->ev_pass[1], s);
    return 0;
}

// group-Fourier variant: 244 instead of 780 slab products per chunk; BN+ReLU between layers in the group domain
static int partI_passF(yoho_ctx* c, const float* x, int B, float* eqv, float* inv, float* inv_np, hipStream_t s) {
    const int nT = (B + TILE - 1) / TILE;
    const size_t nX = (size_t)nT * 4, n256 = (size_t)nT * 32, n512 = (size_t)nT * 64;
    int rc;
    if ((rc = ensure_ws(c, (nX * 4 + n256 * 2 + n512) * CHUNK_FLOATS * sizeof(float), s))) return rc;
    float* bS = (float*)c->ws.p;                  // packed input, group domain
    float* bX = bS + nX * CHUNK_FLOATS;           // its Fourier coefficients
    float* bH0 = bX + nX * CHUNK_FLOATS;          // raw h0 (Fourier), kept for the residual
    float* bA = bH0 + n256 * CHUNK_FLOATS;        // act(h0), later h2 / act(h2)
    float* bM = bA + n256 * CHUNK_FLOATS;         // mid 512 (in place raw -> act)
    float* bY = bM + n512 * CHUNK_FLOATS;         // conv_out raw (Fourier)
    float* bYs = bY + nX * CHUNK_FLOATS;          // conv_out raw (group domain)
    const bool prof = c->profiling && c->ev_created;
    auto mark = [&](int i) { if (prof) (void)hipEventRecord(c->ev[i], s); };
    const Layer* L = c->p1;
    if (prof) { c->ev_chunks = 1; (void)hipEventRecord(c->ev_pass[0], s); }
    mark(0);
    if ((rc = launch_pack_partI(x, B, nT, bS, s))) return rc;
    if ((rc = launch_gft(0, bS, bX, c->dFpad, nullptr, nullptr, nT, 4, s))) return rc;
    mark(1);
    if ((rc = launch_gconvf(L[0], bX, nT, nullptr, bH0, 0, s))) return rc;
    mark(2);
    if ((rc = launch_gft(2, bH0, bA, c->dFpad, L[0].bn_s, L[0].bn_t, nT, 32, s))) return rc;
    mark(3);
    if ((rc = launch_gconvf(L[1], bA, nT, nullptr, bM, 0, s))) return rc;
    mark(4);
    if ((rc = launch_gft(2, bM, bM, c->dFpad, L[1].bn_s, L[1].bn_t, nT, 64, s))) return rc;
    mark(5);
    if ((rc = launch_gconvf(L[2], bM, nT, bH0, bA, EPI_RES, s))) return rc;
    mark(6);
    if ((rc = launch_gft(2, bA, bA, c->dFpad, L[2].bn_s, L[2].bn_t, nT, 32, s))) return rc;
    mark(7);
    if ((rc = launch_gconvf(L[3], bA, nT, nullptr, bY, 0, s))) return rc;
    mark(8);
    if ((rc = launch_gft(1, bY, bYs, c->dFpad, nullptr, nullptr, nT, 4, s))) return rc;
    mark(10);
    if ((rc = launch_finalize_partI(bYs, x, B, eqv, inv, inv_np, 0, s))) return rc;
    mark(9);
    if (prof) (void)hipEventRecord(c->ev_pass[1], s);
    return 0;
}

// group-Fourier variant: all four layers as irrep GEMMs on the fp16x2 split MFMA, fp16x2 transform kernels between them
static size_t partI_G_ws_bytes(int B) {
    const int nT = (B + TILE - 1) / TILE;
    const int kppad = (B + 255) / 256 * 256;
    const size_t nX = (size_t)nT * 4, n256 = (size_t)nT * 32, n512 = (size_t)nT * 64;
    const size_t sz = (nX * 2 + n256 * 2 + n512) * CHUNK_FLOATS * sizeof(float) + fgemm_planes_bytes(kppad, 32) + fgemm_planes_bytes(kppad, 256) +
                      fgemm_planes_bytes(kppad, 512);
    return (sz + 4095) / 4096 * 4096;
}

// one chunk of the pass: B keypoints through head -> 4 GEMMs + 3 transforms -> tail on the workspace slice at `ws`; rows >= B0 of
// the chunk come from x1 (when set).  evbase: first of this chunk's EV_PER_PASS profiling events.
static int partI_passG_chunk(yoho_ctx* c, char* ws, int evbase, const float* x, int B, float* eqv, float* inv, float* inv_np, hipStream_t s,
                             const float* x1, int B0, int slot) {
    // GEMM blocking: mode 4 = 256 x 256 tile, eight waves (two per SIMD) sharing the A stage | mode 5 = 256 x 256, four waves (one per SIMD) |
    // mode 6 = 256 x 128 tiles, two four-wave workgroups per CU.  The transform kernel follows: two waves per SIMD except in mode 5.
    const int gv = c->gconv_mode == 5 ? 1 : (c->gconv_mode == 6 ? 2 : 3);
    const int nT = (B + TILE - 1) / TILE;
    const int kppad = (B + 255) / 256 * 256;
    const size_t nX = (size_t)nT * 4, n256 = (size_t)nT * 32, n512 = (size_t)nT * 64;
    const size_t szP32 = fgemm_planes_bytes(kppad, 32), szP256 = fgemm_planes_bytes(kppad, 256);
    int rc;
    float* bH0 = (float*)ws;                      // raw h0 (Fourier), kept for the residual
    float* bA = bH0 + n256 * CHUNK_FLOATS;        // raw h2
    float* bM = bA + n256 * CHUNK_FLOATS;         // raw mid 512
    float* bY = bM + n512 * CHUNK_FLOATS;         // conv_out raw (Fourier)
    float* bYs = bY + nX * CHUNK_FLOATS;          // conv_out raw (group domain), (B,32,60)
    char* bP32 = (char*)(bYs + nX * CHUNK_FLOATS);    // input coefficients as GEMM operand planes
    char* bP256 = bP32 + szP32;                       // act(h0), later act(h2)
    char* bP512 = bP256 + szP256;                     // act(mid)
    const bool prof = c->profiling && c->ev_created;
    auto mark = [&](int i) { if (prof) (void)hipEventRecord(c->ev[evbase + i], s); };
    const Layer* L = c->p1;
    mark(0);
    int* rf = c->d_rflag;
    int* xc = c->env.xf_steal ? c->d_xfctr + (size_t)slot * 8 : nullptr;     // ticket counters of this stream slot's three transform launches (a launch leaves them at zero)
    // gconv_mode 7: the two large GEMMs take their correction products on the fp8 pipe and scale their activation operand by the largest
    // magnitude the transform in front of them wrote (am[0]: planes of act(h0) for 256 -> 512, am[1]: planes of act(mid) for 512 -> 256)
    unsigned* am = c->gconv_mode == 7 ? c->d_amax + (size_t)slot * 4 : nullptr;
    if (am) HIPCHK(hipMemsetAsync(am, 0, 4 * sizeof(unsigned), s));
    if ((rc = launch_head16(x, B, nT, bP32, kppad, c->dF16, s, x1, B0, rf))) return rc;
    mark(1);
    if ((rc = launch_fgemm(L[0], bP32, kppad, nT, nullptr, bH0, 0, s, rf, gv))) return rc;
    mark(2);
    if ((rc = launch_gft16(bH0, nullptr, bP256, kppad, c->dF16, L[0].bn_s, L[0].bn_t, nT, 32, c->nCU, s, 0, rf, gv, xc, am))) return rc;
    mark(3);
    if ((rc = launch_fgemm(L[1], bP256, kppad, nT, nullptr, bM, 0, s, rf, gv, am))) return rc;
    mark(4);
    if ((rc = launch_gft16(bM, nullptr, bP512, kppad, c->dF16, L[1].bn_s, L[1].bn_t, nT, 64, c->nCU, s, 0, rf, gv, xc ? xc + 2 : nullptr, am ? am + 1 : nullptr))) return rc;
    mark(5);
    if ((rc = launch_fgemm(L[2], bP512, kppad, nT, bH0, bA, EPI_RES, s, rf, gv, am ? am + 1 : nullptr))) return rc;
    mark(6);
    if ((rc = launch_gft16(bA, nullptr, bP256, kppad, c->dF16, L[2].bn_s, L[2].bn_t, nT, 32, c->nCU, s, 0, rf, gv, xc ? xc + 4 : nullptr))) return rc;
    mark(7);
    if ((rc = launch_fgemm(L[3], bP256, kppad, nT, nullptr, bY, 0, s, rf, gv))) return rc;
    mark(8);
    if ((rc = launch_gft16(bY, bYs, nullptr, kppad, c->dF16, nullptr, nullptr, nT, 4, c->nCU, s, B, rf))) return rc;
    mark(10);
    if ((rc = launch_finalize_partI(bYs, x, B, eqv, inv, inv_np, 2, s, x1, B0))) return rc;
    mark(9);
    return 0;
}

// The pass over B keypoints (rows >= B0 from x1 when set).  Breadth-first (one chunk) unless yoho_set_partI_schedule asked for the
// depth-first schedule: chunks of partI_chunk keypoints, each with its own workspace slice per stream.  In every mode but 7 the results are
// bit-identical (a keypoint's arithmetic does not depend on which other keypoints share its launch).  Mode 7 ('fgemm8') is NOT
// schedule-invariant: the fp8 scale of a layer's correction planes is ONE amax word per launch, so a keypoint's bits depend on which
// other keypoints share the launch (pair pass vs single pass, chunked vs breadth-first); the differences stay inside the mode's 1e-5.
static int partI_passG(yoho_ctx* c, const float* x, int B, float* eqv, float* inv, float* inv_np, hipStream_t s, const float* x1 = nullptr,
                       int B0 = 0) {
    for (int i = 0; i < 4; ++i) if (!c->p1[i].wpg) { set_error("irrep-GEMM weights missing"); return YOHO_ENOWEIGHTS; }
    const int chunk = c->partI_chunk > 0 ? c->partI_chunk : B;
    const int nch = (B + chunk - 1) / chunk;
    const int nstr = (nch > 1 && c->partI_streams == 2) ? 2 : 1;
    const size_t slice = partI_G_ws_bytes(nch > 1 ? chunk : B);
    int rc;
    static const char* dbg = experiment_env("YOHO_PARTI_DEBUG");   // experiments (-DYOHO_EXPERIMENTS builds only): "serial" = wait after every chunk, "sideonly" = all chunks on the side stream
    // the workspace is sized BEFORE the side stream is forked (growing it frees the old allocation behind a device synchronisation)
    if ((rc = ensure_ws(c, slice * (size_t)((dbg && std::strstr(dbg, "ownslice")) ? nch : nstr), s))) return rc;
    const bool prof = c->profiling && c->ev_created;
    if (prof) {
        if ((rc = ensure_events(c, nch))) return rc;
        c->ev_chunks = nch;
        (void)hipEventRecord(c->ev_pass[0], s);
    }
    if (nstr == 2) {
        if (!c->side_stream) {
            HIPCHK(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
        }
        HIPCHK(hipEventRecord(c->ev_fork, s));                      // the side stream starts behind everything queued on the caller's
        HIPCHK(hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
    }
    const bool dbg_serial = dbg && std::strstr(dbg, "serial"), dbg_side = dbg && std::strstr(dbg, "sideonly");
    const bool dbg_own = dbg && std::strstr(dbg, "ownslice");      // every chunk on its own workspace slice (no address is reused inside a pass)
    for (int k = 0; k < nch; ++k) {
        const int off = k * chunk, n = B - off < chunk ? B - off : chunk;
        const float *xc, *x1c = nullptr;
        int B0c = 0;
        if (x1 && off >= B0) xc = x1 + (size_t)(off - B0) * F * G;
        else {
            xc = x + (size_t)off * F * G;
            if (x1 && off + n > B0) { x1c = x1; B0c = B0 - off; }
        }
        const bool on_side = nstr == 2 && ((k & 1) || dbg_side);
        hipStream_t sk = on_side ? c->side_stream : s;
        char* ws = (char*)c->ws.p + (dbg_own ? slice * (size_t)k : (on_side ? slice : 0));
        if ((rc = partI_passG_chunk(c, ws, k * EV_PER_PASS, xc, n, eqv + (size_t)off * F * G, inv ? inv + (size_t)off * F : nullptr,
                                    inv_np ? inv_np + (size_t)off * F : nullptr, sk, x1c, B0c, on_side ? 1 : 0))) break;
        if (dbg_serial && hipStreamSynchronize(sk) != hipSuccess) { set_error("hipStreamSynchronize failed"); rc = YOHO_EHIP; break; }
    }
    if (nstr == 2) {
        // also on an error inside the loop: the caller's stream continues behind whatever the side stream was given, so the fork is
        // always joined and the call keeps its contract (ordered on the caller's stream) when it reports a failure
        HIPCHK(hipEventRecord(c->ev_join, c->side_stream));
        HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
    }
    if (rc) return rc;
    if (prof) (void)hipEventRecord(c->ev_pass[1], s);
    return 0;
}

static int partI_pass(yoho_ctx* c, const float* x, int B, float* eqv, float* inv, float* inv_np, hipStream_t s) {
    if (c->gconv_mode >= 4) return partI_passG(c, x, B, eqv, inv, inv_np, s);
    if (c->gconv_mode == 1 || c->gconv_mode == 3) return partI_pass16(c, x, B, eqv, inv, inv_np, s, c->gconv_mode == 1 ? 3 : 2);
    if (c->gconv_mode == 2) return partI_passF(c, x, B, eqv, inv, inv_np, s);
    const int nT = (B + TILE - 1) / TILE;
    const size_t ch = (size_t)CHUNK_FLOATS * sizeof(float);
    const size_t nX = (size_t)nT * 4, n256 = (size_t)nT * 32, n512 = (size_t)nT * 64;
    int rc;
    if ((rc = ensure_ws(c, (nX * 2 + n256 * 2 + n512) * ch, s))) return rc;
    float* bX = (float*)c->ws.p;
    float* bH0 = bX + nX * CHUNK_FLOATS;
    float* bA = bH0 + n256 * CHUNK_FLOATS;        // a0, later a2
    float* bA1 = bA + n256 * CHUNK_FLOATS;
    float* bY = bA1 + n512 * CHUNK_FLOATS;
    const bool prof = c->profiling && c->ev_created;
    auto mark = [&](int i) { if (prof) (void)hipEventRecord(c->ev[i], s); };
    if (prof) { c->ev_chunks = 1; (void)hipEventRecord(c->ev_pass[0], s); }
    mark(0);
    if ((rc = launch_pack_partI(x, B, nT, bX, s))) return rc;
    mark(1);
    if ((rc = launch_gconv(conv_args(c->p1[0], bX, nT, nullptr, bH0, bA, false), 15, EPI_RAW | EPI_ACT, s))) return rc;
    mark(2); mark(3);
    if ((rc = launch_gconv(conv_args(c->p1[1], bA, nT, nullptr, nullptr, bA1, false), 15, EPI_ACT, s))) return rc;
    mark(4); mark(5);
    if ((rc = launch_gconv(conv_args(c->p1[2], bA1, nT, bH0, nullptr, bA, false), 15, EPI_RES | EPI_ACT, s))) return rc;
    mark(6); mark(7);
    if ((rc = launch_gconv(conv_args(c->p1[3], bA, nT, nullptr, bY, nullptr, false), 15, EPI_RAW, s))) return rc;
    mark(8); mark(10);
    if ((rc = launch_finalize_partI(bY, x, B, eqv, inv, inv_np, 0, s))) return rc;
    mark(9);
    if (prof) (void)hipEventRecord(c->ev_pass[1], s);
    return 0;
}

int yoho_partI_forward(yoho_ctx* c, const float* x, int B, float* eqv, float* inv, float* inv_np, void* stream) {
    if (!c || !x || !eqv || B < 1) { set_error("yoho_partI_forward: bad argument (B=%d)", B); return YOHO_EINVAL; }
    if (!c->has_partI) { set_error("yoho_partI_forward: PartI weights not loaded"); return YOHO_ENOWEIGHTS; }
    YOHO_NEED_ALIGNED("yoho_partI_forward", 15, x, eqv, inv, inv_np);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const int MAXB = 16384;                      // bounds the workspace to ~4.5 GB
    for (int b0 = 0; b0 < B; b0 += MAXB) {
        const int nb = B - b0 < MAXB ? B - b0 : MAXB;
        int rc = partI_pass(c, x + (size_t)b0 * F * G, nb, eqv + (size_t)b0 * F * G, inv ? inv + (size_t)b0 * F : nullptr,
                            inv_np ? inv_np + (size_t)b0 * F : nullptr, s);
        if (rc) return rc;
    }
    return 0;
}

int yoho_group_mean_np(yoho_ctx* c, const float* eqv, int B, float* out, void* stream) {
    if (!c || !eqv || !out || B < 0) { set_error("yoho_group_mean_np: bad argument"); return YOHO_EINVAL; }
    if (B == 0) return 0;
    YOHO_NEED_ALIGNED("yoho_group_mean_np", 15, eqv, out);
    HIPCHK(hipSetDevice(c->device));
    return launch_group_mean_np(eqv, B, out, (hipStream_t)stream);
}

// PartII with the two large cone layers (128->256 @45 g, 256->512 @13 g) on the bf16x3 split MFMA; the g = 0 tail
// (512->256 conv + the 1x1 MLP) stays on the fp32 kernels, fed through fp32 32-tile hand-over buffers.
static bool partII_fourier_head(const yoho_ctx* c) {
    return c->partII_mode >= 2 && c->p2[0].wpg;
}

static int partII_pass16(yoho_ctx* c, const float* s0, const float* s1, const float* s2, const float* s3, const int64_t* idx,
                         int M, float* quat, hipStream_t s, int npl, const int64_t* const* ridx = nullptr, int istride = 1) {
    const int nT16 = (M + 15) / 16, nT = (M + TILE - 1) / TILE;
    const size_t ch = (size_t)CHUNK_FLOATS * sizeof(float), ch16 = (size_t)npl * 15360;
    const size_t n128 = (size_t)nT * 16, n256 = (size_t)nT * 32, n512 = (size_t)nT * 64, n32 = (size_t)nT * 4;
    // (PartII modes 3 / 4 keep the cone GEMM's stage blocks where the direct kernels keep their 256-channel planes: whole 256-match column tiles)
    const size_t szG = c->partII_mode >= 3 ? (size_t)((nT + 7) / 8) * c->cone_nslot * 8 * 32768 : 0;
    const size_t szX = (size_t)nT16 * 16 * ch16, szA0 = std::max((size_t)nT16 * 32 * ch16, szG);
    const size_t szA1p = npl == 2 ? (size_t)nT16 * 64 * ch16 : 0;      // fp16x2: 13-cone activation planes for cone1_kernel
    int rc;
    int* rf = c->d_rflag + 1;                           // PartII's range word
    if ((rc = ensure_ws(c, szX + szA0 + szA1p + (n256 + n512 + n256 + n512 + n128 + n32) * ch, s))) return rc;
    char* bX = (char*)c->ws.p;                          // 128 ch planes
    char* bA0 = bX + szX;                               // 256 ch planes (45 slabs valid)
    char* bA1p = bA0 + szA0;                            // 512 ch planes (13 slabs valid)
    float* bH0 = (float*)(bA1p + szA1p);                // 256 raw fp32, 32-tile layout
    float* bA1 = bH0 + n256 * CHUNK_FLOATS;             // 512 act fp32 (13 slabs valid)
    float* bF = bA1 + n512 * CHUNK_FLOATS;              // 256 raw (g = 0)
    float* bF0 = bF + n256 * CHUNK_FLOATS;              // 512 act
    float* bF1 = bF0 + n512 * CHUNK_FLOATS;             // 128 act
    float* bQ = bF1 + n128 * CHUNK_FLOATS;              // 32 raw (4 used)
    if (ridx && !partII_fourier_head(c)) { set_error("indexed PartII needs the default (fp16x2, Fourier first layer) mode"); return YOHO_EINVAL; }
    if (npl == 2 && partII_fourier_head(c)) {
        // first layer (128 -> 256) in the group-Fourier domain: all 60 outputs cost 244/780 of a full direct layer, i.e.
        // less than half of the 45-element cone the direct kernel computes
        const int kppad = (M + 255) / 256 * 256;
        const size_t szP = fgemm_planes_bytes(kppad, 128);
        if ((rc = ensure_ws(c, szX + szA0 + szA1p + (n256 + n512 + n256 + n512 + n128 + n32) * ch + szP + n256 * ch, s))) return rc;
        // ensure_ws may have moved the workspace
        bX = (char*)c->ws.p; bA0 = bX + szX; bA1p = bA0 + szA0; bH0 = (float*)(bA1p + szA1p); bA1 = bH0 + n256 * CHUNK_FLOATS;
        bF = bA1 + n512 * CHUNK_FLOATS; bF0 = bF + n256 * CHUNK_FLOATS; bF1 = bF0 + n512 * CHUNK_FLOATS; bQ = bF1 + n128 * CHUNK_FLOATS;
        char* bP = (char*)(bQ + n32 * CHUNK_FLOATS);
        float* bC = (float*)(bP + szP);                 // raw Fourier coefficients of the first layer
        if ((rc = launch_head2(s0, s1, s2, s3, idx, c->dP, c->p2_init_bn_s, c->p2_init_bn_t, M, nT, bP, kppad, c->dF16, s, ridx, istride, rf))) return rc;
        if ((rc = launch_fgemm(c->p2[0], bP, kppad, nT, nullptr, bC, 0, s, rf, c->env.partII_l1_variant))) return rc;
        if (c->partII_mode >= 3 && c->p2[1].wcg && !c->env.partII_tail_staged && mlp_head_supported(c->p2[3], c->p2[4], c->p2[5])) {
            // modes 3 / 4: the 13-element cone layer as ONE implicit GEMM (cgemm_kernel, gemmf2.hip).  The inverse transform leaves the 45
            // cone elements of the first layer's output as B-operand stage blocks [column tile][slot][32-channel block][32 KiB] (in the
            // region the direct kernels' planes bA0 occupy: 45 of 60 slabs, the same bytes per match), the GEMM writes cone1's planes
            const int nslot = c->cone_nslot;
            unsigned* amax = nullptr;
            if (c->partII_mode == 4) {
                amax = c->d_amax + 8;                        // slot of its own behind PartI's [2][4] words
                HIPCHK(hipMemsetAsync(amax, 0, sizeof(unsigned), s));
            }
            if ((rc = launch_gft16_invg(bC, bH0, bA0, c->cone_slot_of, nslot, c->dF16, c->p2[0].bn_s, c->p2[0].bn_t, nT, 32, c->nCU, s, rf, amax))) return rc;
            if ((rc = launch_cgemm(c->p2[1], bA0, nslot, c->cone_slot, c->cone_outg, nT, nT16, bA1p, s, rf, amax))) return rc;
            int n0[NTAP];
            for (int k = 0; k < NTAP; ++k) n0[k] = c->hN[k];
            if ((rc = launch_cone1(c->p2[2], bA1p, nT, nT16, nullptr, nullptr, n0, s, bF0))) return rc;
            return launch_mlp_head(c->p2[3], c->p2[4], c->p2[5], nullptr, nT, M, quat, s, bF0, &c->p2[2], bH0);
        }
        if ((rc = launch_gft16_invp(bC, bH0, bA0, nT16, c->dF16, c->p2[0].bn_s, c->p2[0].bn_t, nT, 32, c->nCU, s, rf))) return rc;
    } else {
        if ((rc = launch_pack16_partII(s0, s1, s2, s3, idx, c->dP, c->p2_init_bn_s, c->p2_init_bn_t, M, nT16, bX, s, npl, rf))) return rc;
        if ((rc = launch_gconv16(c->p2[0], bX, nT16, nullptr, nullptr, bA0, EPI_RAW32 | EPI_ACT, s, 1, bH0, nullptr, npl, rf))) return rc;
    }
    if (npl == 2) {
        int n0[NTAP];
        for (int k = 0; k < NTAP; ++k) n0[k] = c->hN[k];
        if ((rc = launch_gconv16(c->p2[1], bA0, nT16, nullptr, nullptr, bA1p, EPI_ACT, s, 2, nullptr, nullptr, npl, rf))) return rc;
        // with the one-launch tail behind it the layer runs its K over two workgroups per tile and the tail adds the halves (bF0,
        // the staged tail's 512-channel buffer, holds them)
        if (!c->env.partII_tail_staged && mlp_head_supported(c->p2[3], c->p2[4], c->p2[5])) {
            if ((rc = launch_cone1(c->p2[2], bA1p, nT, nT16, nullptr, nullptr, n0, s, bF0))) return rc;
            return launch_mlp_head(c->p2[3], c->p2[4], c->p2[5], nullptr, nT, M, quat, s, bF0, &c->p2[2], bH0);
        }
        if ((rc = launch_cone1(c->p2[2], bA1p, nT, nT16, bH0, bF, n0, s))) return rc;
    } else {
        if ((rc = launch_gconv16(c->p2[1], bA0, nT16, nullptr, nullptr, nullptr, EPI_ACT32, s, 2, nullptr, bA1, npl))) return rc;
        if ((rc = launch_gconv(conv_args(c->p2[2], bA1, nT, bH0, bF, nullptr, true), -1, EPI_RES | EPI_RAW, s))) return rc;
    }
    if (!c->env.partII_tail_staged && mlp_head_supported(c->p2[3], c->p2[4], c->p2[5])) return launch_mlp_head(c->p2[3], c->p2[4], c->p2[5], bF, nT, M, quat, s);
    if ((rc = launch_gconv(conv_args(c->p2[3], bF, nT, nullptr, nullptr, bF0, true), -1, EPI_ACT, s))) return rc;
    if ((rc = launch_gconv(conv_args(c->p2[4], bF0, nT, nullptr, nullptr, bF1, true), -1, EPI_ACT, s))) return rc;
    if ((rc = launch_gconv(conv_args(c->p2[5], bF1, nT, nullptr, bQ, nullptr, false), 1, EPI_RAW, s))) return rc;
    return launch_quat_norm(bQ, M, quat, s);
}

static int partII_pass(yoho_ctx* c, const float* s0, const float* s1, const float* s2, const float* s3, const int64_t* idx,
                       int M, float* quat, hipStream_t s) {
    if (c->partII_mode != 0) return partII_pass16(c, s0, s1, s2, s3, idx, M, quat, s, c->partII_mode == 1 ? 3 : 2);      // modes 2, 3, 4: two planes
    const int nT = (M + TILE - 1) / TILE;
    const size_t ch = (size_t)CHUNK_FLOATS * sizeof(float);
    const size_t n128 = (size_t)nT * 16, n256 = (size_t)nT * 32, n512 = (size_t)nT * 64, n32 = (size_t)nT * 4;
    int rc;
    if ((rc = ensure_ws(c, (n128 + 3 * n256 + 2 * n512 + n128 + n32) * ch, s))) return rc;
    float* bX = (float*)c->ws.p;                       // 128 ch
    float* bH0 = bX + n128 * CHUNK_FLOATS;             // 256 raw
    float* bA0 = bH0 + n256 * CHUNK_FLOATS;            // 256 act
    float* bA1 = bA0 + n256 * CHUNK_FLOATS;            // 512 act
    float* bF = bA1 + n512 * CHUNK_FLOATS;             // 256 raw (g = 0)
    float* bF0 = bF + n256 * CHUNK_FLOATS;             // 512 act
    float* bF1 = bF0 + n512 * CHUNK_FLOATS;            // 128 act
    float* bQ = bF1 + n128 * CHUNK_FLOATS;             // 32 raw (4 used)
    if ((rc = launch_pack_partII(s0, s1, s2, s3, idx, c->dP, c->p2_init_bn_s, c->p2_init_bn_t, M, nT, bX, s))) return rc;
    // Only group element 0 of the last feature map is consumed (utils/network.py:273-276), so the
    // convs run on its receptive cone: 45 -> 13 -> 1 group elements.
    if ((rc = launch_gconv(conv_args(c->p2[0], bX, nT, nullptr, bH0, bA0, false), 12, EPI_RAW | EPI_ACT, s))) return rc;
    if ((rc = launch_gconv(conv_args(c->p2[1], bA0, nT, nullptr, nullptr, bA1, false), 4, EPI_ACT, s))) return rc;
    if ((rc = launch_gconv(conv_args(c->p2[2], bA1, nT, bH0, bF, nullptr, true), -1, EPI_RES | EPI_RAW, s))) return rc;
    if (!c->env.partII_tail_staged && mlp_head_supported(c->p2[3], c->p2[4], c->p2[5])) return launch_mlp_head(c->p2[3], c->p2[4], c->p2[5], bF, nT, M, quat, s);
    if ((rc = launch_gconv(conv_args(c->p2[3], bF, nT, nullptr, nullptr, bF0, true), -1, EPI_ACT, s))) return rc;
    if ((rc = launch_gconv(conv_args(c->p2[4], bF0, nT, nullptr, nullptr, bF1, true), -1, EPI_ACT, s))) return rc;
    if ((rc = launch_gconv(conv_args(c->p2[5], bF1, nT, nullptr, bQ, nullptr, false), 1, EPI_RAW, s))) return rc;
    return launch_quat_norm(bQ, M, quat, s);
}

int yoho_partII_forward(yoho_ctx* c, const float* before_eqv0, const float* before_eqv1, const float* after_eqv0,
                        const float* after_eqv1, const int64_t* pre_idx, int M, float* quat, void* stream) {
    if (!c || M < 0) { set_error("yoho_partII_forward: bad argument"); return YOHO_EINVAL; }
    if (!c->has_partII) { set_error("yoho_partII_forward: PartII weights not loaded"); return YOHO_ENOWEIGHTS; }
    if (M == 0) return 0;                                  // empty match set: nothing to do (pointers may be null)
    if (!before_eqv0 || !before_eqv1 || !after_eqv0 || !after_eqv1 || !pre_idx || !quat) {
        set_error("yoho_partII_forward: bad argument"); return YOHO_EINVAL;
    }
    YOHO_NEED_ALIGNED("yoho_partII_forward", 15, before_eqv0, before_eqv1, after_eqv0, after_eqv1, quat);
    YOHO_NEED_ALIGNED("yoho_partII_forward", 7, pre_idx);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const int MAXM = 8192;
    for (int m0 = 0; m0 < M; m0 += MAXM) {
        const int nm = M - m0 < MAXM ? M - m0 : MAXM;
        const size_t o = (size_t)m0 * F * G;
        int rc = partII_pass(c, before_eqv0 + o, before_eqv1 + o, after_eqv0 + o, after_eqv1 + o, pre_idx + m0, nm,
                             quat + (size_t)m0 * 4, s);
        if (rc) return rc;
    }
    return 0;
}

int yoho_partII_forward_indexed(yoho_ctx* c, const float* s0, const int64_t* i0, const float* s1, const int64_t* i1, const float* s2,
                                const int64_t* i2, const float* s3, const int64_t* i3, int istride, const int64_t* pre_idx, int M,
                                float* quat, void* stream) {
    if (!c || M < 0 || istride < 1) { set_error("yoho_partII_forward_indexed: bad argument"); return YOHO_EINVAL; }
    if (!c->has_partII) { set_error("yoho_partII_forward_indexed: PartII weights not loaded"); return YOHO_ENOWEIGHTS; }
    if (M == 0) return 0;
    if (!s0 || !s1 || !s2 || !s3 || !i0 || !i1 || !i2 || !i3 || !pre_idx || !quat) {
        set_error("yoho_partII_forward_indexed: bad argument"); return YOHO_EINVAL;
    }
    if (!partII_fourier_head(c)) { set_error("yoho_partII_forward_indexed: needs the default PartII mode (fp16x2, Fourier first layer)"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_partII_forward_indexed", 15, s0, s1, s2, s3, quat);
    YOHO_NEED_ALIGNED("yoho_partII_forward_indexed", 7, i0, i1, i2, i3, pre_idx);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const int MAXM = 8192;
    for (int m0 = 0; m0 < M; m0 += MAXM) {
        const int nm = M - m0 < MAXM ? M - m0 : MAXM;
        const size_t io = (size_t)m0 * istride;
        const int64_t* ridx[4] = {i0 + io, i1 + io, i2 + io, i3 + io};
        int rc = partII_pass16(c, s0, s1, s2, s3, pre_idx + m0, nm, quat + (size_t)m0 * 4, s, 2, ridx, istride);
        if (rc) return rc;
    }
    return 0;
}

int yoho_load_fcgf(yoho_ctx* c, const yoho_fcgf_config* cfg, const float* const* tensors, int ntensors) {
    if (!c || !cfg || !tensors) { set_error("yoho_load_fcgf: null argument"); return YOHO_EINVAL; }
    for (int i = 0; i < ntensors; ++i) if (!tensors[i]) { set_error("yoho_load_fcgf: null tensor %d", i); return YOHO_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    FcgfNet* n = nullptr;
    int rc = fcgf_load(&n, cfg, tensors, ntensors, c->env.fcgf_f32);
    if (rc) return rc;
    if (c->fcgf) fcgf_free(c->fcgf);
    c->fcgf = n;
    return 0;
}

int yoho_fcgf_voxelize(yoho_ctx* c, const double* pts, int n, double voxel_size, int64_t* sel, int32_t* coords, int* count, void* stream) {
    if (!c || !count || n < 0 || !(voxel_size > 0)) { set_error("yoho_fcgf_voxelize: bad argument"); return YOHO_EINVAL; }
    if (n == 0) { *count = 0; return 0; }
    if (!pts || !sel || !coords) { set_error("yoho_fcgf_voxelize: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_fcgf_voxelize", 7, pts, sel);
    YOHO_NEED_ALIGNED("yoho_fcgf_voxelize", 3, coords);
    HIPCHK(hipSetDevice(c->device));
    return fcgf_voxelize(c, pts, n, nullptr, voxel_size, sel, coords, nullptr, count, (hipStream_t)stream);
}

int yoho_fcgf_voxelize_rotated(yoho_ctx* c, const double* pts, int n, const double* R_host, double voxel_size, int64_t* sel, int32_t* coords,
                               float* pts_sel, int* count, void* stream) {
    if (!c || !count || !R_host || n < 0 || !(voxel_size > 0)) { set_error("yoho_fcgf_voxelize_rotated: bad argument"); return YOHO_EINVAL; }
    if (n == 0) { *count = 0; return 0; }
    if (!pts || !sel || !coords) { set_error("yoho_fcgf_voxelize_rotated: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_fcgf_voxelize_rotated", 7, pts, sel);
    YOHO_NEED_ALIGNED("yoho_fcgf_voxelize_rotated", 3, coords, pts_sel);
    HIPCHK(hipSetDevice(c->device));
    return fcgf_voxelize(c, pts, n, R_host, voxel_size, sel, coords, pts_sel, count, (hipStream_t)stream);
}

int yoho_fcgf_voxelize_rotated_batch(yoho_ctx* c, const double* pts, int n, const double* R_host, int nb, double voxel_size, int64_t* sel,
                                     int32_t* coords, float* pts_sel, int* counts, void* stream) {
    if (!c || !counts || !R_host || n < 0 || nb < 1 || !(voxel_size > 0)) { set_error("yoho_fcgf_voxelize_rotated_batch: bad argument"); return YOHO_EINVAL; }
    if (n > 0 && (!pts || !sel || !coords)) { set_error("yoho_fcgf_voxelize_rotated_batch: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_fcgf_voxelize_rotated_batch", 7, pts, sel);
    YOHO_NEED_ALIGNED("yoho_fcgf_voxelize_rotated_batch", 3, coords, pts_sel);
    HIPCHK(hipSetDevice(c->device));
    return fcgf_voxelize_batch(c, pts, n, R_host, nb, voxel_size, sel, coords, pts_sel, counts, (hipStream_t)stream);
}

int yoho_rotate_select(yoho_ctx* c, const double* pts, const double* R_host, const int64_t* sel, int m, float* out, void* stream) {
    if (!c || m < 0) { set_error("yoho_rotate_select: bad argument"); return YOHO_EINVAL; }
    if (m == 0) return 0;
    if (!pts || !sel || !out) { set_error("yoho_rotate_select: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_rotate_select", 7, pts, sel);
    YOHO_NEED_ALIGNED("yoho_rotate_select", 3, out);
    HIPCHK(hipSetDevice(c->device));
    return fcgf_rotate_select(pts, R_host, sel, m, out, (hipStream_t)stream);
}

int yoho_group_transfer_batch(yoho_ctx* c, const double* pts, const int64_t* kidx, int K, const double* R_host, int nb,
                              const float* const* ds, const float* const* feat, const int* m, int g0, float* out,
                              float* q_scratch, int64_t* idx_scratch, void* stream) {
    if (!c || K < 0 || nb < 1 || nb > 64 || g0 < 0 || g0 + nb > G) { set_error("yoho_group_transfer_batch: bad argument (1..64 copies, g0 + nb <= 60)"); return YOHO_EINVAL; }
    if (K == 0) return 0;
    if (!pts || !kidx || !R_host || !ds || !feat || !m || !out || !q_scratch || !idx_scratch) { set_error("yoho_group_transfer_batch: null argument"); return YOHO_EINVAL; }
    for (int b = 0; b < nb; ++b)
        if (!ds[b] || !feat[b] || m[b] < 1) { set_error("yoho_group_transfer_batch: copy %d has no down-sampled points", b); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_group_transfer_batch", 7, pts, kidx, idx_scratch);
    YOHO_NEED_ALIGNED("yoho_group_transfer_batch", 15, out);
    int rc;
    phase_mark(c, 15, (hipStream_t)stream);
    if (c->nn_cell > 0.0 && !c->env.transfer_staged) {                 // (YOHO_TRANSFER=staged: A/B)
        // with a cell-size hint: all copies of the pass through the hash grid in four launches (gridnn.hip), the keypoints rotated and
        // the feature rows written by the query kernel itself; q_scratch / idx_scratch stay unused
        HIPCHK(hipSetDevice(c->device));
        int mmax = 1;
        for (int b = 0; b < nb; ++b) mmax = m[b] > mmax ? m[b] : mmax;
        if ((rc = ensure_ws(c, grid_transfer_ws_bytes(K, nb, mmax), (hipStream_t)stream))) return rc;
        rc = launch_grid_transfer_batch(pts, kidx, K, R_host, nb, ds, feat, m, g0, out, c->nn_cell, c->ws.p, c->nCU, (hipStream_t)stream);
        phase_mark(c, -1, (hipStream_t)stream);
        return rc;
    }
    for (int b = 0; b < nb; ++b) {
        if ((rc = yoho_rotate_select(c, pts, R_host + 9 * (size_t)b, kidx, K, q_scratch, stream))) return rc;
        if ((rc = yoho_nn_search(c, q_scratch, K, ds[b], m[b], 3, YOHO_DIST_SQUARE_L2, idx_scratch, nullptr, stream))) return rc;
        if ((rc = yoho_group_scatter(c, feat[b], m[b], idx_scratch, K, g0 + b, out, stream))) return rc;
    }
    phase_mark(c, -1, (hipStream_t)stream);
    return 0;
}

int yoho_fcgf_forward(yoho_ctx* c, const int32_t* coords, int n, float* out, void* stream) {
    if (!c || n < 0) { set_error("yoho_fcgf_forward: bad argument"); return YOHO_EINVAL; }
    if (!c->fcgf) { set_error("yoho_fcgf_forward: backbone weights not loaded"); return YOHO_ENOWEIGHTS; }
    if (n == 0) return 0;
    if (!coords || !out) { set_error("yoho_fcgf_forward: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_fcgf_forward", 15, out);
    YOHO_NEED_ALIGNED("yoho_fcgf_forward", 3, coords);
    HIPCHK(hipSetDevice(c->device));
    return fcgf_forward(c, c->fcgf, coords, n, nullptr, 1, out, (hipStream_t)stream);
}

int yoho_fcgf_forward_batch(yoho_ctx* c, const int32_t* coords, const int32_t* offsets, int nb, float* out, void* stream) {
    if (!c || !offsets || nb < 1 || nb > 64) { set_error("yoho_fcgf_forward_batch: bad argument (1..64 clouds)"); return YOHO_EINVAL; }
    if (!c->fcgf) { set_error("yoho_fcgf_forward_batch: backbone weights not loaded"); return YOHO_ENOWEIGHTS; }
    if (offsets[0] != 0) { set_error("yoho_fcgf_forward_batch: offsets[0] must be 0"); return YOHO_EINVAL; }
    for (int b = 0; b < nb; ++b) if (offsets[b + 1] < offsets[b]) { set_error("yoho_fcgf_forward_batch: offsets must be non-decreasing"); return YOHO_EINVAL; }
    const int n = offsets[nb];
    if (n == 0) return 0;
    if (!coords || !out) { set_error("yoho_fcgf_forward_batch: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_fcgf_forward_batch", 15, out);
    YOHO_NEED_ALIGNED("yoho_fcgf_forward_batch", 3, coords);
    HIPCHK(hipSetDevice(c->device));
    return fcgf_forward(c, c->fcgf, coords, n, offsets, nb, out, (hipStream_t)stream);
}

int yoho_partI_forward_pair(yoho_ctx* c, const float* x0, int B0, const float* x1, int B1, float* eqv, float* inv, float* inv_np,
                            void* stream) {
    if (!c || !x0 || !x1 || !eqv || B0 < 1 || B1 < 1) { set_error("yoho_partI_forward_pair: bad argument"); return YOHO_EINVAL; }
    if (!c->has_partI) { set_error("yoho_partI_forward_pair: PartI weights not loaded"); return YOHO_ENOWEIGHTS; }
    if (c->gconv_mode < 4 || B0 + B1 > 16384) { set_error("yoho_partI_forward_pair: default arithmetic mode and at most 16384 keypoints"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_partI_forward_pair", 15, x0, x1, eqv, inv, inv_np);
    HIPCHK(hipSetDevice(c->device));
    return partI_passG(c, x0, B0 + B1, eqv, inv, inv_np, (hipStream_t)stream, x1, B0);
}

int yoho_gconv_layer(yoho_ctx* c, const float* x, int B, int cin, int cout, const float* weight, const float* bias, int transpose,
                     float* y, void* stream) {
    if (!c || B < 0 || cin < 1 || cout < 1) { set_error("yoho_gconv_layer: bad argument"); return YOHO_EINVAL; }
    if (B == 0) return 0;
    if (!x || !weight || !y) { set_error("yoho_gconv_layer: bad argument"); return YOHO_EINVAL; }
    if (transpose && c->tap_inv[0] < 0) { set_error("yoho_gconv_layer: the neighbour table is not closed under inversion"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_gconv_layer", 15, x, weight, y);
    HIPCHK(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const int xc = transpose ? cout : cin, yc = transpose ? cin : cout;
    const int MAXB = 4096;
    for (int b0 = 0; b0 < B; b0 += MAXB) {
        const int nb = B - b0 < MAXB ? B - b0 : MAXB;
        int rc = gconv_layer(c, x + (size_t)b0 * xc * G, nb, cin, cout, weight, bias, transpose, y + (size_t)b0 * yc * G, s);
        if (rc) return rc;
    }
    return 0;
}

int yoho_bn_stats(yoho_ctx* c, const float* x, int B, int C, float* mean, float* var, void* stream) {
    if (!c || !x || !mean || !var || B < 1 || C < 1) { set_error("yoho_bn_stats: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_bn_stats", 15, x);
    HIPCHK(hipSetDevice(c->device));
    return bn_stats(x, B, C, mean, var, (hipStream_t)stream);
}

int yoho_bn_relu_apply(yoho_ctx* c, const float* x, int B, int C, const float* scale, const float* shift, float* y, void* stream) {
    if (!c || !x || !scale || !shift || !y || B < 1 || C < 1) { set_error("yoho_bn_relu_apply: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_bn_relu_apply", 15, x, y);
    HIPCHK(hipSetDevice(c->device));
    return bn_relu_apply(x, B, C, scale, shift, y, (hipStream_t)stream);
}

int yoho_bn_relu_backward(yoho_ctx* c, const float* x, const float* y, const float* dy, int B, int C, const float* gamma, const float* mean,
                          const float* rstd, int batch_stats, float* dx, float* dgamma, float* dbeta, void* stream) {
    if (!c || !x || !y || !dy || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || B < 1 || C < 1) {
        set_error("yoho_bn_relu_backward: bad argument"); return YOHO_EINVAL;
    }
    YOHO_NEED_ALIGNED("yoho_bn_relu_backward", 15, x, y, dy, dx);
    HIPCHK(hipSetDevice(c->device));
    return bn_relu_backward(x, y, dy, B, C, gamma, mean, rstd, batch_stats, dx, dgamma, dbeta, (hipStream_t)stream);
}

int yoho_gconv_wgrad(yoho_ctx* c, const float* x, const float* dy, int B, int cin, int cout, float* dW, float* db, void* stream) {
    if (!c || B < 1 || cin < 1 || cout < 1 || !x || !dy || !dW) { set_error("yoho_gconv_wgrad: bad argument"); return YOHO_EINVAL; }
    YOHO_NEED_ALIGNED("yoho_gconv_wgrad", 15, x, dy, dW);
    HIPCHK(hipSetDevice(c->device));
    return gconv_wgrad(c, x, dy, B, cin, cout, dW, db, (hipStream_t)stream);
}

}  // extern "C"
