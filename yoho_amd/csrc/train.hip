// Training-layout group-conv layer on device-resident parameters (forward and data gradient of the
// (1,13) convolution over the icosahedral neighbourhood, utils/network.py:46-52 + nn.Conv2d(Cin, Cout, (1,13))).
//
//   forward    y[b,o,g]  = bias[o] + sum_k sum_c W[o,c,k] * x[b,c,N[g,k]]
//   data grad  dx[b,c,g] = sum_k sum_o W[o,c,k] * dy[b,o,N'_k[g]],   N'_k = inverse of the permutation g -> N[g,k]
//
// For every tap g -> N[g,k] = n_k * g is a permutation of the group and the tap set {n_k} is closed under inversion,
// so N'_k = N[., inv(k)]: the data gradient is the SAME convolution with the weights transposed in (o,c) and the taps
// relabelled by inv - both directions run on gconv_kernel (fp32 MFMA, gconv.hip).  Here: (B,C,60) <-> tile layout and the
// on-device weight re-packing (the parameters change every optimiser step, so they never visit the host).
#include <hip/hip_runtime.h>

#include "common.h"

namespace yoho {

// x (B,C,60) -> [tile][c8][g][h][kp][4]; one workgroup per (tile, c8)
__global__ __launch_bounds__(256) void pack_bcg_kernel(const float* __restrict__ x, int B, int C8, float* __restrict__ out) {
    __shared__ float lds[CHUNK_FLOATS];
    const int tile = blockIdx.x / C8, c8 = blockIdx.x - tile * C8;
    const int C = C8 * 8;
    for (int i = threadIdx.x; i < TILE * 8 * G; i += 256) {
        const int kp = i / (8 * G);
        const int r = i - kp * (8 * G);
        const int cl = r / G, g = r - cl * G;
        const int b = tile * TILE + kp;
        const float v = b < B ? x[((size_t)b * C + (c8 * 8 + cl)) * G + g] : 0.f;
        lds[(g * 2 + (cl >> 2)) * (TILE * 4) + kp * 4 + (cl & 3)] = v;
    }
    __syncthreads();
    float4* o = reinterpret_cast<float4*>(out + ((size_t)tile * C8 + c8) * CHUNK_FLOATS);
    const float4* l = reinterpret_cast<const float4*>(lds);
    for (int i = threadIdx.x; i < CHUNK_FLOATS / 4; i += 256) o[i] = l[i];
}

// tile layout with C8p chunks per tile -> y (B,C,60), C <= 8 * C8p
__global__ __launch_bounds__(256) void unpack_bcg_kernel(const float* __restrict__ t, int B, int C8p, int C, float* __restrict__ y) {
    __shared__ float lds[CHUNK_FLOATS];
    const int tile = blockIdx.x / C8p, c8 = blockIdx.x - tile * C8p;
    const float4* src = reinterpret_cast<const float4*>(t + ((size_t)tile * C8p + c8) * CHUNK_FLOATS);
    float4* l = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < CHUNK_FLOATS / 4; i += 256) l[i] = src[i];
    __syncthreads();
    for (int i = threadIdx.x; i < TILE * 8 * G; i += 256) {
        const int kp = i / (8 * G);
        const int r = i - kp * (8 * G);
        const int cl = r / G, g = r - cl * G;
        const int b = tile * TILE + kp, c = c8 * 8 + cl;
        if (b < B && c < C) y[((size_t)b * C + c) * G + g] = lds[(g * 2 + (cl >> 2)) * (TILE * 4) + kp * 4 + (cl & 3)];
    }
}

// W (cout, cin, 1, 13) on the device -> MFMA A-fragment order of the effective layer
//   wp[ob][c8][tap][lane = 32h + i][s] = Weff[ob*32 + i][c8*8 + 4h + s][tap]
//   forward: Weff[o][c][k] = W[o][c][k];   data gradient: Weff[c][o][k] = W[o][c][inv[k]]  (effective cin = cout, cout = cin)
__global__ void pack_w_dev_kernel(const float* __restrict__ W, int cin, int cout, int transpose, const int* __restrict__ inv,
                                  int ecin, int ecout, int nob, float* __restrict__ wp) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)nob * (ecin / 8) * NTAP * 256;
    if (idx >= total) return;
    const int s = idx & 3, lane = (idx >> 2) & 63;
    size_t r = idx >> 8;
    const int tap = r % NTAP; r /= NTAP;
    const int c8 = r % (ecin / 8);
    const int ob = r / (ecin / 8);
    const int eo = ob * 32 + (lane & 31), ec = c8 * 8 + 4 * (lane >> 5) + s;
    float v = 0.f;
    if (eo < ecout) v = transpose ? W[((size_t)ec * cin + eo) * NTAP + inv[tap]] : W[((size_t)eo * cin + ec) * NTAP + tap];
    wp[idx] = v;
}

__global__ void pad_bias_kernel(const float* __restrict__ b, int n, int npad, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < npad) out[i] = (b && i < n) ? b[i] : 0.f;
}

// Weight gradient of the layer (the third product of its backward pass):
//   dW[o,c,0,k] = sum_b sum_g dy[b,o,g] * x[b,c,N[g,k]]          db[o] = sum_b sum_g dy[b,o,g]
// i.e. per tap a (cout x cin) product contracted over (batch, group element), the activations read through the neighbour
// table instead of a materialised (B,C,60,13) gather (utils/network.py:46-52 under autograd).  One workgroup per
// (32 output channels, 32 input channels): thread (oi = tid >> 3, cj = tid & 7) owns dW[oi][4 cj .. 4 cj + 3][0..12] - 52 fp32
// accumulators - and walks the batch in order (deterministic sums); per sample the two 32 x 60 blocks are staged in LDS
// ([g][channel], so the four channels of a thread are one 16-byte read and dy is a broadcast).  Training batches are a few
// dozen keypoints (train/trainer.py), so this kernel is judged on parity, not on the roofline.
__global__ __launch_bounds__(256) void wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int B, int cin, int cout,
                                                    const int* __restrict__ nei, float* __restrict__ dW, float* __restrict__ db) {
    __shared__ float xs[G][32];
    __shared__ float ds[G][32];
    __shared__ int nb[G * NTAP];
    const int ob = blockIdx.x, cb = blockIdx.y;
    const int tid = threadIdx.x, oi = tid >> 3, cj = tid & 7;
    for (int i = tid; i < G * NTAP; i += 256) nb[i] = nei[i];
    float acc[4][NTAP];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < NTAP; ++k) acc[e][k] = 0.f;
    float bsum = 0.f;
    for (int b = 0; b < B; ++b) {
        __syncthreads();
        const float* xp = x + ((size_t)b * cin + cb * 32) * G;          // 32 channels x 60 group elements, contiguous
        const float* dp = dy + ((size_t)b * cout + ob * 32) * G;
        for (int i = tid; i < 32 * G; i += 256) {
            const int ch = i / G, g = i - ch * G;
            xs[g][ch] = xp[i];
            ds[g][ch] = dp[i];
        }
        __syncthreads();
        for (int g = 0; g < G; ++g) {
            const float d = ds[g][oi];
            bsum += d;
#pragma unroll
            for (int k = 0; k < NTAP; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(&xs[nb[g * NTAP + k]][4 * cj]);
                acc[0][k] = fmaf(d, v.x, acc[0][k]);
                acc[1][k] = fmaf(d, v.y, acc[1][k]);
                acc[2][k] = fmaf(d, v.z, acc[2][k]);
                acc[3][k] = fmaf(d, v.w, acc[3][k]);
            }
        }
    }
    float* o = dW + ((size_t)(ob * 32 + oi) * cin + cb * 32 + 4 * cj) * NTAP;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < NTAP; ++k) o[e * NTAP + k] = acc[e][k];
    if (db && cb == 0 && cj == 0) db[ob * 32 + oi] = bsum;
}

// ---- BatchNorm (batch statistics) + ReLU of a (B,C,60) tensor, fused (utils/network.py:16-17: BatchNorm2d + ReLU in front of every
// conv; on the un-gathered tensor, see train/network.py) ---------------------------------------------------------------------
// stats: one workgroup per channel, f64 sums over the B x 60 values -> mean, biased variance
__device__ __forceinline__ double wg_sum(double v, double* red) {
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int B, int C, float* __restrict__ mean, float* __restrict__ var) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < B * G; i += 256) {
        const int b = i / G, g = i - b * G;
        const double v = x[((size_t)b * C + c) * G + g];
        s += v; q += v * v;
    }
    s = wg_sum(s, red);
    q = wg_sum(q, red);
    if (threadIdx.x == 0) {
        const double n = (double)B * G, m = s / n;
        mean[c] = (float)m;
        var[c] = (float)fmax(q / n - m * m, 0.0);
    }
}

// y = relu(x * scale[c] + shift[c])
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float* __restrict__ x, size_t n, int C, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ y) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)((i / G) % C);
    y[i] = fmaxf(fmaf(x[i], scale[c], shift[c]), 0.f);
}

// backward, reductions: dz = dy where y > 0; sum_dz[c], sum_dz_xhat[c] with xhat = (x - mean) * rstd
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                                 int B, int C, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 float* __restrict__ sum_dz, float* __restrict__ sum_dzx) {
    __shared__ double red[4];
    const int c = blockIdx.x;
    const double m = mean[c], r = rstd[c];
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < B * G; i += 256) {
        const int b = i / G, g = i - b * G;
        const size_t o = ((size_t)b * C + c) * G + g;
        const double dz = y[o] > 0.f ? (double)dy[o] : 0.0;
        s += dz; q += dz * (((double)x[o] - m) * r);
    }
    s = wg_sum(s, red);
    q = wg_sum(q, red);
    if (threadIdx.x == 0) { sum_dz[c] = (float)s; sum_dzx[c] = (float)q; }
}

// dx = gamma * rstd * (dz - sum_dz / N - xhat * sum_dzx / N)    (batch statistics);    dx = dz * gamma * rstd    (running statistics)
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ dy,
                                                                size_t n, int C, float invN, const float* __restrict__ gamma,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ sum_dz, const float* __restrict__ sum_dzx, int batch_stats,
                                                                float* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)((i / G) % C);
    const float dz = y[i] > 0.f ? dy[i] : 0.f;
    const float gr = gamma[c] * rstd[c];
    if (!batch_stats) { dx[i] = dz * gr; return; }
    const float xh = (x[i] - mean[c]) * rstd[c];
    dx[i] = gr * (dz - sum_dz[c] * invN - xh * sum_dzx[c] * invN);
}

int bn_stats(const float* x, int B, int C, float* mean, float* var, hipStream_t s) {
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C), dim3(256), 0, s, x, B, C, mean, var);
    HIPCHK(hipGetLastError());
    return 0;
}
int bn_relu_apply(const float* x, int B, int C, const float* scale, const float* shift, float* y, hipStream_t s) {
    const size_t n = (size_t)B * C * G;
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, C, scale, shift, y);
    HIPCHK(hipGetLastError());
    return 0;
}
int bn_relu_backward(const float* x, const float* y, const float* dy, int B, int C, const float* gamma, const float* mean, const float* rstd,
                     int batch_stats, float* dx, float* dgamma, float* dbeta, hipStream_t s) {
    // dbeta = sum dz, dgamma = sum dz * xhat: exactly the two reductions the input gradient needs
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel, dim3(C), dim3(256), 0, s, x, y, dy, B, C, mean, rstd, dbeta, dgamma);
    const size_t n = (size_t)B * C * G;
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, dy, n, C, 1.f / ((float)B * G), gamma, mean,
                       rstd, dbeta, dgamma, batch_stats, dx);
    HIPCHK(hipGetLastError());
    return 0;
}

int gconv_wgrad(yoho_ctx* c, const float* x, const float* dy, int B, int cin, int cout, float* dW, float* db, hipStream_t s) {
    if (cin % 32 || cout % 32) { set_error("yoho_gconv_wgrad: channel counts must be multiples of 32 (got %d -> %d)", cin, cout); return YOHO_EINVAL; }
    hipLaunchKernelGGL(wgrad_kernel, dim3(cout / 32, cin / 32), dim3(256), 0, s, x, dy, B, cin, cout, c->dN, dW, db);
    HIPCHK(hipGetLastError());
    return 0;
}

int gconv_layer(yoho_ctx* c, const float* x, int B, int cin, int cout, const float* W, const float* bias, int transpose, float* y,
                hipStream_t s) {
    const int ecin = transpose ? cout : cin, ecout = transpose ? cin : cout;
    if (ecin % 8) { set_error("yoho_gconv_layer: input channels must be a multiple of 8"); return YOHO_EINVAL; }
    const int cpad = (ecout + 31) / 32 * 32, cpad64 = (cpad + 63) / 64 * 64;
    const int nob = cpad / 32, nT = (B + TILE - 1) / TILE;
    const size_t szX = (size_t)nT * (ecin / 8) * CHUNK_FLOATS, szY = (size_t)nT * (cpad / 8) * CHUNK_FLOATS;
    const size_t szW = (size_t)nob * (ecin / 8) * NTAP * 256;
    int rc;
    if ((rc = ensure_ws(c, (szX + szY + szW + cpad64 + 64) * sizeof(float), s))) return rc;
    float* bX = (float*)c->ws.p;
    float* bY = bX + szX;
    float* wp = bY + szY;
    float* bb = wp + szW;
    hipLaunchKernelGGL(pack_bcg_kernel, dim3(nT * (ecin / 8)), dim3(256), 0, s, x, B, ecin / 8, bX);
    hipLaunchKernelGGL(pack_w_dev_kernel, dim3((unsigned)((szW + 255) / 256)), dim3(256), 0, s, W, cin, cout, transpose, c->d_tap_inv, ecin, ecout,
                       nob, wp);
    hipLaunchKernelGGL(pad_bias_kernel, dim3((cpad64 + 255) / 256), dim3(256), 0, s, transpose ? nullptr : bias, ecout, cpad64, bb);
    HIPCHK(hipGetLastError());
    Layer L;
    L.cin = ecin; L.cout = ecout; L.cout_pad = cpad; L.ntaps = NTAP; L.wp = wp; L.bias = bb;
    ConvArgs a;
    a.X = bX; a.Wp = L.wp; a.bias = L.bias; a.bn_s = nullptr; a.bn_t = nullptr; a.res = nullptr; a.out_raw = bY; a.out_act = nullptr;
    a.nTiles = nT; a.cin8 = ecin / 8; a.cout8 = cpad / 8; a.nOB = nob; a.ntaps = NTAP;
    a.slabtab = c->tabs.slabtab; a.outg = c->tabs.outg;
    if ((rc = launch_gconv(a, 15, EPI_RAW, s))) return rc;
    hipLaunchKernelGGL(unpack_bcg_kernel, dim3(nT * (cpad / 8)), dim3(256), 0, s, bY, B, cpad / 8, ecout, y);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
