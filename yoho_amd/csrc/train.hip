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
    if ((rc = launch_gconv(a, 15, EPI_RAW, s))) return rc;
    hipLaunchKernelGGL(unpack_bcg_kernel, dim3(nT * (cpad / 8)), dim3(256), 0, s, bY, B, cpad / 8, ecout, y);
    HIPCHK(hipGetLastError());
    return 0;
}

}  // namespace yoho
