// FCGF backbone: sparse 3-D ResUNet forward pass on a voxelised cloud, fp32.
// Reference: fcgf_model/resunet.py:10-190 (ResUNet2 family), residual_block.py:9-52, simple_yoho/fcgf_feat.py:33-49, written
// against MinkowskiEngine 0.5.x; the sparse-tensor semantics implemented here are those listed in oracle/fcgf_oracle.py.
//
//   coordinate maps   open-addressing hash tables (64-bit packed voxel key -> row); a coarser map is built by inserting the
//                     quantised coordinates with atomicMin of the source row, so its rows come out in first-occurrence order
//                     (the CPU coordinate manager's order), compacted by a block-count / scan / scatter pass;
//   kernel maps       map[k][n] = input row at coord(n) + offset(k) (or coord(n) - offset(k) for a transposed conv), -1 if the
//                     voxel is empty: output-stationary, so a convolution needs no atomics and sums in kernel-index order;
//   convolution       one wave = 32 output rows x all output channels on v_mfma_f32_32x32x2_f32: A = gathered input rows
//                     (a lane reads 16 consecutive channels of its row straight from global memory, no LDS, no barriers),
//                     B = W[k] read through L1/L2, accumulators = Cout/32 x 16 registers; BN / residual / ReLU / channel
//                     concatenation (write at a column offset of a wider buffer) in the epilogue.
#include <hip/hip_runtime.h>
#include <vector>
#include <algorithm>
#include <cstring>
#include <cmath>
#include <cstdlib>

#include "common.h"

namespace yoho {

typedef unsigned long long u64;
typedef float floatx16s __attribute__((ext_vector_type(16)));
constexpr u64 HEMPTY = ~0ull;

// 19 bits per axis (|voxel index| < 2^18) + 7 bits of cloud (batch) index
__device__ __forceinline__ u64 pack_key(int x, int y, int z, int b) {
    return ((u64)(unsigned)b << 57) | ((u64)(unsigned)((x + (1 << 18)) & 0x7FFFF) << 38) | ((u64)(unsigned)((y + (1 << 18)) & 0x7FFFF) << 19) |
           (u64)(unsigned)((z + (1 << 18)) & 0x7FFFF);
}
__device__ __forceinline__ unsigned hslot(u64 key, unsigned mask) { return (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 33) & mask; }
__device__ __forceinline__ int floor_to(int c, int ts) {            // floor(c / ts) * ts  (src/coordinate_map.hpp:58-76)
    if (ts <= 1) return c;
    int q = c / ts;
    if ((c % ts) != 0 && c < 0) --q;
    return q * ts;
}

__global__ void hash_clear_kernel(u64* keys, int* vals, unsigned cap) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    if (i < cap) { keys[i] = HEMPTY; vals[i] = 0x7FFFFFFF; }
}

// voxel of point i: from integer coordinates (quantised to `ts`) or from f64 points (floor(p / voxel), fcgf_feat.py:34)
struct CoordSrc {
    const int* coords;       // (n,4) rows (x, y, z, cloud) or null
    const double* pts;       // (n,3) or null
    double voxel;
    int ts;
    int rot;                 // pts are rotated on the fly: p' = R p  (the 60 rotated copies of a fragment, YOHO_testset.py:143)
    double R[9];
    int* oor;                // optional device flag: raised when a point's voxel index does not fit the 19-bit key fields
    int* dup;                // optional device flag: raised by the table insert when a voxel arrives a second time
};
constexpr int VOX_LIM = (1 << 18) - 16;      // |voxel index| bound of pack_key, minus the reach of the coarsest kernel offsets
// floor(p / voxel) as an int with defined behaviour for huge or non-finite values (they land outside VOX_LIM)
__device__ __forceinline__ int voxel_index(double p, double voxel) {
    const double q = floor(p / voxel);
    return (q >= -1073741824.0 && q <= 1073741824.0) ? (int)q : 1073741824;
}
// one coordinate of R p in f64, fixed operation order
__device__ __forceinline__ double rot_coord(const double* R3, double p0, double p1, double p2) { return fma(p2, R3[2], fma(p1, R3[1], p0 * R3[0])); }
__device__ __forceinline__ void point_of(const CoordSrc& s, int i, double& p0, double& p1, double& p2) {
    const double q0 = s.pts[3 * (size_t)i], q1 = s.pts[3 * (size_t)i + 1], q2 = s.pts[3 * (size_t)i + 2];
    if (s.rot) { p0 = rot_coord(s.R, q0, q1, q2); p1 = rot_coord(s.R + 3, q0, q1, q2); p2 = rot_coord(s.R + 6, q0, q1, q2); }
    else { p0 = q0; p1 = q1; p2 = q2; }
}
__device__ __forceinline__ void voxel_of(const CoordSrc& s, int i, int& x, int& y, int& z, int& b) {
    if (s.pts) {
        double p0, p1, p2;
        point_of(s, i, p0, p1, p2);
        x = voxel_index(p0, s.voxel);
        y = voxel_index(p1, s.voxel);
        z = voxel_index(p2, s.voxel);
        b = 0;
        if (s.oor && (x < -VOX_LIM || x > VOX_LIM || y < -VOX_LIM || y > VOX_LIM || z < -VOX_LIM || z > VOX_LIM)) atomicOr(s.oor, 1);
    } else {
        const int4 c = reinterpret_cast<const int4*>(s.coords)[i];
        x = floor_to(c.x, s.ts); y = floor_to(c.y, s.ts); z = floor_to(c.z, s.ts); b = c.w;
    }
}

// slot value = smallest source row with that voxel
__global__ void hash_insert_min_kernel(CoordSrc src, int n, u64* keys, int* vals, unsigned mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int x, y, z, b;
    voxel_of(src, i, x, y, z, b);
    const u64 key = pack_key(x, y, z, b);
    unsigned s = hslot(key, mask);
    for (;;) {
        const u64 old = atomicCAS(&keys[s], HEMPTY, key);
        if (old == HEMPTY || old == key) {
            atomicMin(&vals[s], i);
            if (old == key && src.dup) atomicOr(src.dup, 1);
            return;
        }
        s = (s + 1) & mask;
    }
}

__device__ __forceinline__ int hash_find_slot(const u64* keys, unsigned mask, u64 key) {
    unsigned s = hslot(key, mask);
    for (;;) {
        const u64 k = keys[s];
        if (k == key) return (int)s;
        if (k == HEMPTY) return -1;
        s = (s + 1) & mask;
    }
}

// first occurrences in source order -> new rows (order = the CPU coordinate manager's).  Three phases: per-block counts,
// single-workgroup scan of the block counts, per-block ballot scan + scatter.
__device__ __forceinline__ bool is_first(const CoordSrc& src, int i, int n, const u64* keys, const int* vals, unsigned mask, int& x, int& y,
                                         int& z, int& b) {
    if (i >= n) return false;
    voxel_of(src, i, x, y, z, b);
    const int slot = hash_find_slot(keys, mask, pack_key(x, y, z, b));
    return vals[slot] == i;
}

__global__ __launch_bounds__(1024) void first_count_kernel(CoordSrc src, int n, const u64* keys, const int* vals, unsigned mask, int* bsum) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int x, y, z, b;
    const bool keep = is_first(src, blockIdx.x * 1024 + tid, n, keys, vals, mask, x, y, z, b);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    if (tid == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; bsum[blockIdx.x] = t; }
}

// exclusive scan of nb block counts in place, total -> *count
__global__ __launch_bounds__(1024) void block_scan_kernel(int* bsum, int nb, int* count) {
    __shared__ int sh[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        const int i = b0 + tid;
        const int v = i < nb ? bsum[i] : 0;
        sh[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int t = tid >= o ? sh[tid - o] : 0;
            __syncthreads();
            sh[tid] += t;
            __syncthreads();
        }
        if (i < nb) bsum[i] = carry + sh[tid] - v;
        __syncthreads();
        if (tid == 0) carry += sh[1023];
        __syncthreads();
    }
    if (tid == 0) *count = carry;
}

// out_coords rows: ocs = 3 (x, y, z: the caller's voxelisation output) or 4 (x, y, z, cloud: internal coordinate maps)
__global__ __launch_bounds__(1024) void first_scatter_kernel(CoordSrc src, int n, const u64* keys, const int* vals, unsigned mask,
                                                             const int* bsum, int* out_coords, int ocs, int64_t* sel) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = blockIdx.x * 1024 + tid;
    int x = 0, y = 0, z = 0, b = 0;
    const bool keep = is_first(src, i, n, keys, vals, mask, x, y, z, b);
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = bsum[blockIdx.x];
    for (int k = 0; k < wv; ++k) off += wsum[k];
    if (keep) {
        const int r = off + before;
        out_coords[ocs * (size_t)r] = x; out_coords[ocs * (size_t)r + 1] = y; out_coords[ocs * (size_t)r + 2] = z;
        if (ocs == 4) out_coords[4 * (size_t)r + 3] = b;
        if (sel) sel[r] = i;
    }
}

static int launch_first_compact(const CoordSrc& src, int n, const u64* keys, const int* vals, unsigned mask, int* bsum, int* out_coords,
                                int ocs, int64_t* sel, int* count, hipStream_t s) {
    const int nb = (n + 1023) / 1024;
    hipLaunchKernelGGL(first_count_kernel, dim3(nb), dim3(1024), 0, s, src, n, keys, vals, mask, bsum);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, s, bsum, nb, count);
    hipLaunchKernelGGL(first_scatter_kernel, dim3(nb), dim3(1024), 0, s, src, n, keys, vals, mask, bsum, out_coords, ocs, sel);
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- rank-ordered occupancy bitmaps: the coordinate maps of all four levels without a hash table ------------------------------
// When every cloud of a pass fits a dense bitmap (it does for anything the backbone is used on: bounding boxes of a few hundred voxels
// per axis), a level's coordinate map IS its bitmap plus a prefix popcount: row(voxel) = rank[word] + popcount(bits below it).  The
// bitmap of level l + 1 is the 2 x 2 x 2 OR-reduction of level l's (coordinates are floored to the coarser stride,
// src/coordinate_map.hpp:58-76, and the bitmap origin is a multiple of 16, so flooring is a shift of the cell index); sizes, rows and
// coordinates of all levels come out of bit operations and scans over a few MB instead of four hash tables of up to 48 MB built with
// two atomics per voxel (8 ms per fragment with the lookups that followed).  The INTERNAL row order of every level becomes rank order.
// Ranks run over 32 (x) x 8 x 8 bricks of words, so rows that are close in space are close in memory - the job the cell sort did
// for level 0.  The order is free: a row's sum is taken in kernel-offset order whatever its number, level 0 is handed back in the
// caller's order (operm), so every output bit is what the hash-table path produces (YOHO_FCGF_COORDS=hash, and the automatic fall-back
// for clouds too large for a bitmap or inputs with duplicate voxels).
struct RkDesc {
    long long base;          // first word of this cloud's bitmap at this level (row-major: x words fastest, then y, then z)
    int x0, y0, z0;          // voxel coordinate of cell (0,0,0): the same multiples of 16 at every level
    int wx, ny, nz;          // words per x row, rows, slices at this level
    long long rbase;         // first entry of this cloud's rank array at this level
    int nyb, nrank;          // y bricks; rank entries of this cloud = nzb * nyb * wx * 64
    int blk0;                // first 1024-entry scan block of this cloud
};
__device__ __forceinline__ long long rk_index(const RkDesc& d, int w, int Y, int Z) {
    return d.rbase + ((long long)((Z >> 3) * d.nyb + (Y >> 3)) * d.wx + w) * 64 + (Z & 7) * 8 + (Y & 7);
}
// row of the voxel at coordinate (qx, qy, qz) - a multiple of the level's stride 2^sh - or -1
__device__ __forceinline__ int rk_lookup(const RkDesc& d, const unsigned* __restrict__ bm, const int* __restrict__ rank, int qx, int qy, int qz, int sh) {
    const int X = (qx - d.x0) >> sh, Y = (qy - d.y0) >> sh, Z = (qz - d.z0) >> sh;
    if (X < 0 || X >= d.wx * 32 || Y < 0 || Y >= d.ny || Z < 0 || Z >= d.nz) return -1;
    const unsigned word = bm[d.base + ((long long)Z * d.ny + Y) * d.wx + (X >> 5)];
    const int bit = X & 31;
    if (!((word >> bit) & 1u)) return -1;
    return rank[rk_index(d, X >> 5, Y, Z)] + __popc(word & ((1u << bit) - 1u));
}

// level l -> l + 1: out cell (X, Y, Z) = OR of the in cells (2X .. 2X+1, 2Y .. 2Y+1, 2Z .. 2Z+1); one thread per output word
__global__ void rk_coarsen_kernel(const RkDesc* __restrict__ din, const RkDesc* __restrict__ dout, const unsigned* __restrict__ bin, unsigned* __restrict__ bout) {
    const RkDesc di = din[blockIdx.y], d = dout[blockIdx.y];
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)d.wx * d.ny * d.nz) return;
    const int w = (int)(i % d.wx), Y = (int)((i / d.wx) % d.ny), Z = (int)(i / ((long long)d.wx * d.ny));
    unsigned a = 0u, b = 0u;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int y = 2 * Y + dy, z = 2 * Z + dz;
            if (y < di.ny && z < di.nz) {
                const unsigned* row = bin + di.base + ((long long)z * di.ny + y) * di.wx;
                a |= row[2 * w];
                if (2 * w + 1 < di.wx) b |= row[2 * w + 1];
            }
        }
    auto squeeze = [](unsigned v) {                      // bit i of the result = bits 2i | 2i+1 of v
        v = (v | (v >> 1)) & 0x55555555u;
        v = (v | (v >> 1)) & 0x33333333u;
        v = (v | (v >> 2)) & 0x0F0F0F0Fu;
        v = (v | (v >> 4)) & 0x00FF00FFu;
        v = (v | (v >> 8)) & 0x0000FFFFu;
        return v;
    };
    bout[d.base + i] = squeeze(a) | (squeeze(b) << 16);
}

// the word of rank entry r (brick order) of cloud descriptor d, 0 for the padding of incomplete bricks
__device__ __forceinline__ unsigned rk_word_of(const RkDesc& d, const unsigned* __restrict__ bm, int r, int& w, int& Y, int& Z) {
    const int in = r & 63, br = r >> 6;
    w = br % d.wx;
    const int byz = br / d.wx;
    Y = (byz % d.nyb) * 8 + (in & 7);
    Z = (byz / d.nyb) * 8 + (in >> 3);
    return (Y < d.ny && Z < d.nz) ? bm[d.base + ((long long)Z * d.ny + Y) * d.wx + w] : 0u;
}

// popcounts of 1024 rank entries: exclusive prefix inside the block -> rank[], block total -> btot[]   (grid: blocks of the cloud, cloud)
__global__ __launch_bounds__(1024) void rk_count_kernel(const RkDesc* __restrict__ desc, const unsigned* __restrict__ bm, int* __restrict__ rank,
                                                         int* __restrict__ btot) {
    __shared__ int wsum[16];
    const RkDesc d = desc[blockIdx.y];
    if ((int)blockIdx.x * 1024 >= d.nrank) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r = blockIdx.x * 1024 + tid;
    int w, Y, Z;
    const int v = r < d.nrank ? __popc(rk_word_of(d, bm, r, w, Y, Z)) : 0;
    int sc = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(sc, o); if (lane >= o) sc += t; }
    if (lane == 63) wsum[wv] = sc;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < wv; ++k) off += wsum[k];
    if (r < d.nrank) rank[d.rbase + r] = off + sc - v;
    if (tid == 1023) btot[d.blk0 + blockIdx.x] = off + sc;
}

// finishes rank[] (adds the scanned block offsets) and writes the level's rows in rank order: coords[row] = (x, y, z, cloud)
__global__ __launch_bounds__(256) void rk_rows_kernel(const RkDesc* __restrict__ desc, const unsigned* __restrict__ bm, int* __restrict__ rank,
                                                      const int* __restrict__ bscan, int ts, int* __restrict__ coords) {
    const RkDesc d = desc[blockIdx.y];
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= d.nrank) return;
    int w, Y, Z;
    unsigned word = rk_word_of(d, bm, r, w, Y, Z);
    const int row0 = rank[d.rbase + r] + bscan[d.blk0 + (r >> 10)];
    rank[d.rbase + r] = row0;
    int k = 0;
    while (word) {
        const int bit = __ffs(word) - 1;
        word &= word - 1u;
        reinterpret_cast<int4*>(coords)[row0 + k] = make_int4(d.x0 + (w * 32 + bit) * ts, d.y0 + Y * ts, d.z0 + Z * ts, (int)blockIdx.y);
        ++k;
    }
}

// caller's level-0 row i -> internal row: operm[row] = i (the input voxels of a cloud are distinct, so every row has one writer)
__global__ void rk_operm_kernel(const int* __restrict__ c4, int n, const RkDesc* __restrict__ desc, const unsigned* __restrict__ bm,
                                const int* __restrict__ rank, int* __restrict__ operm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(c4)[i];
    const int r = rk_lookup(desc[c.w], bm, rank, c.x, c.y, c.z, 0);
    if (r >= 0) operm[r] = i;
}

__global__ void rk_fill_kernel(const int* __restrict__ c4, int n, const RkDesc* __restrict__ desc, unsigned* __restrict__ bm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(c4)[i];
    const RkDesc d = desc[c.w];
    const int bx = c.x - d.x0;
    atomicOr(bm + d.base + ((long long)(c.z - d.z0) * d.ny + (c.y - d.y0)) * d.wx + (bx >> 5), 1u << (bx & 31));
}

// ---- batched voxelisation: the rotated copies of ONE cloud, copy = blockIdx.y --------------------------------------------------
// Per copy the same stages as fcgf_voxelize (insert-min, count, scan, scatter), but one launch per stage for up to VOX_BATCH
// copies: 300 k points are ~1200 workgroups, far too few to cover the latency of the table atomics, and 7 launches per copy were
// 105 per backbone pass.  The rotations travel in the kernel arguments.  A copy's table / block sums / counters are slices of one
// allocation; the scatter also writes the rotated fp32 points of the selected rows (the reference's pcd[sel].float(): the very f64
// values the voxel index was taken from), so no second pass over `sel` is needed.
constexpr int VOX_BATCH = 16;
struct VoxBatch {
    const double* pts; int n; double voxel;
    double R[VOX_BATCH][9];
    u64* keys; int* vals; unsigned cap;          // copy b: keys + b * cap
    int* bsum; int nblk;                         // copy b: bsum + b * (nblk + 1)
    int* dcount;                                 // copy b: [2b] voxels, [2b + 1] out-of-range flag
    int* coords; int64_t* sel; float* pts_sel;   // copy b: + b * n rows (pts_sel may be null)
    // rank-ordered bitmaps instead of the tables (rk != null): copy b's bitmap descriptor rk[b0 + b], the first point of voxel row r in first[r]
    const RkDesc* rk; const unsigned* bm; const int* rank; int* first; int b0;
};
__device__ __forceinline__ void vox_point(const VoxBatch& a, int b, int i, double& p0, double& p1, double& p2) {
    const double q0 = a.pts[3 * (size_t)i], q1 = a.pts[3 * (size_t)i + 1], q2 = a.pts[3 * (size_t)i + 2];
    const double* R = a.R[b];
    p0 = rot_coord(R, q0, q1, q2); p1 = rot_coord(R + 3, q0, q1, q2); p2 = rot_coord(R + 6, q0, q1, q2);
}
__global__ void vox_clear_kernel(u64* keys, int* vals, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) { keys[i] = HEMPTY; vals[i] = 0x7FFFFFFF; }
}
__global__ void vox_insert_kernel(VoxBatch a) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= a.n) return;
    double p0, p1, p2;
    vox_point(a, b, i, p0, p1, p2);
    const int x = voxel_index(p0, a.voxel), y = voxel_index(p1, a.voxel), z = voxel_index(p2, a.voxel);
    if (x < -VOX_LIM || x > VOX_LIM || y < -VOX_LIM || y > VOX_LIM || z < -VOX_LIM || z > VOX_LIM) atomicOr(a.dcount + 2 * b + 1, 1);
    const u64 key = pack_key(x, y, z, 0);
    u64* keys = a.keys + (size_t)b * a.cap;
    int* vals = a.vals + (size_t)b * a.cap;
    const unsigned mask = a.cap - 1;
    unsigned s = hslot(key, mask);
    for (;;) {
        const u64 old = atomicCAS(&keys[s], HEMPTY, key);
        if (old == HEMPTY || old == key) { atomicMin(&vals[s], i); return; }
        s = (s + 1) & mask;
    }
}
__device__ __forceinline__ bool vox_is_first(const VoxBatch& a, int b, int i, int& x, int& y, int& z, double& p0, double& p1, double& p2) {
    if (i >= a.n) return false;
    vox_point(a, b, i, p0, p1, p2);
    x = voxel_index(p0, a.voxel); y = voxel_index(p1, a.voxel); z = voxel_index(p2, a.voxel);
    if (a.rk) {
        const int r = rk_lookup(a.rk[a.b0 + b], a.bm, a.rank, x, y, z, 0);
        return r >= 0 && a.first[r] == i;
    }
    const int slot = hash_find_slot(a.keys + (size_t)b * a.cap, a.cap - 1, pack_key(x, y, z, 0));
    return a.vals[(size_t)b * a.cap + slot] == i;
}
// rank mode, pass 1: the voxel of every (point, copy) sets its bit; a voxel outside its copy's bitmap raises a.dcount[2b + 1]
__global__ void vox_fill_kernel(VoxBatch a, unsigned* bm) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= a.n) return;
    double p0, p1, p2;
    vox_point(a, b, i, p0, p1, p2);
    const int x = voxel_index(p0, a.voxel), y = voxel_index(p1, a.voxel), z = voxel_index(p2, a.voxel);
    const RkDesc d = a.rk[a.b0 + b];
    const int X = x - d.x0, Y = y - d.y0, Z = z - d.z0;
    if (X < 0 || X >= d.wx * 32 || Y < 0 || Y >= d.ny || Z < 0 || Z >= d.nz || x < -VOX_LIM || x > VOX_LIM || y < -VOX_LIM || y > VOX_LIM ||
        z < -VOX_LIM || z > VOX_LIM) { atomicOr(a.dcount + 2 * b + 1, 1); return; }
    // (bound by the rate of device atomics, ~22 G/s: 0.2 ms for the 4.5 M points of a 15-copy pass.  Reading the word first and skipping
    // the atomic when the bit is there was measured: 3 x SLOWER - 0.60 ms - the read goes to memory and sees the bit too rarely)
    atomicOr(bm + d.base + ((long long)Z * d.ny + Y) * d.wx + (X >> 5), 1u << (X & 31));
}
// rank mode, pass 2 (ranks finished): first[row of the voxel] = smallest point index
__global__ void vox_first_kernel(VoxBatch a) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= a.n) return;
    double p0, p1, p2;
    vox_point(a, b, i, p0, p1, p2);
    const int r = rk_lookup(a.rk[a.b0 + b], a.bm, a.rank, voxel_index(p0, a.voxel), voxel_index(p1, a.voxel), voxel_index(p2, a.voxel), 0);
    if (r >= 0) atomicMin(a.first + r, i);                  // (a read-and-skip in front of it: 0.18 -> 0.55 ms, as in vox_fill_kernel)
}
// rank[] += scanned block offsets (rk_rows_kernel does this for the coordinate maps, where it also writes the rows)
__global__ void rk_finish_kernel(const RkDesc* __restrict__ desc, int* __restrict__ rank, const int* __restrict__ bscan) {
    const RkDesc d = desc[blockIdx.y];
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < d.nrank) rank[d.rbase + r] += bscan[d.blk0 + (r >> 10)];
}
// per-workgroup axis-aligned bounds of (n,3) f64 points -> part[block][6] = (min x, y, z, max x, y, z); the host combines the blocks
__global__ __launch_bounds__(256) void aabb_kernel(const double* __restrict__ pts, int n, double* __restrict__ part) {
    __shared__ double red[4][6];
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    bool bad = false;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const double v = pts[3 * (size_t)i + a];
            bad |= !(v > -1e300 && v < 1e300);                 // NaN / inf: reported as an unbounded box, the caller falls back to the tables
            lo[a] = fmin(lo[a], v); hi[a] = fmax(hi[a], v);
        }
    if (bad) { lo[0] = -1e308; hi[0] = 1e308; }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int o = 32; o >= 1; o >>= 1) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], o)); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { for (int a = 0; a < 3; ++a) { red[w][a] = lo[a]; red[w][3 + a] = hi[a]; } }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = red[0][threadIdx.x];
        for (int ww = 1; ww < 4; ++ww) v = threadIdx.x < 3 ? fmin(v, red[ww][threadIdx.x]) : fmax(v, red[ww][threadIdx.x]);
        part[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
__global__ __launch_bounds__(1024) void vox_count_kernel(VoxBatch a) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.y;
    int x, y, z; double p0, p1, p2;
    const bool keep = vox_is_first(a, b, blockIdx.x * 1024 + tid, x, y, z, p0, p1, p2);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    if (tid == 0) { int t = 0; for (int k = 0; k < 16; ++k) t += wsum[k]; a.bsum[(size_t)b * (a.nblk + 1) + blockIdx.x] = t; }
}
// one workgroup per copy: exclusive scan of its block counts in place, total -> dcount[2b]
__global__ __launch_bounds__(1024) void vox_scan_kernel(VoxBatch a) {
    __shared__ int sh[1024];
    __shared__ int carry;
    const int tid = threadIdx.x, b = blockIdx.x;
    int* bsum = a.bsum + (size_t)b * (a.nblk + 1);
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < a.nblk; b0 += 1024) {
        const int i = b0 + tid;
        const int v = i < a.nblk ? bsum[i] : 0;
        sh[tid] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int t = tid >= o ? sh[tid - o] : 0;
            __syncthreads();
            sh[tid] += t;
            __syncthreads();
        }
        if (i < a.nblk) bsum[i] = carry + sh[tid] - v;
        __syncthreads();
        if (tid == 0) carry += sh[1023];
        __syncthreads();
    }
    if (tid == 0) a.dcount[2 * b] = carry;
}
__global__ __launch_bounds__(1024) void vox_scatter_kernel(VoxBatch a) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.y;
    const int i = blockIdx.x * 1024 + tid;
    int x = 0, y = 0, z = 0; double p0 = 0, p1 = 0, p2 = 0;
    const bool keep = vox_is_first(a, b, i, x, y, z, p0, p1, p2);
    const unsigned long long m = __ballot(keep);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __popcll(m);
    __syncthreads();
    int off = a.bsum[(size_t)b * (a.nblk + 1) + blockIdx.x];
    for (int k = 0; k < wv; ++k) off += wsum[k];
    if (keep) {
        const size_t r = (size_t)b * a.n + off + before;
        a.coords[3 * r] = x; a.coords[3 * r + 1] = y; a.coords[3 * r + 2] = z;
        a.sel[r] = i;
        if (a.pts_sel) { a.pts_sel[3 * r] = (float)p0; a.pts_sel[3 * r + 1] = (float)p1; a.pts_sel[3 * r + 2] = (float)p2; }
    }
}

// table value := row of the compacted map
__global__ void hash_set_rows_kernel(const int* coords, int n, const u64* keys, int* vals, unsigned mask) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[r];
    const int slot = hash_find_slot(keys, mask, pack_key(c.x, c.y, c.z, c.w));
    vals[slot] = r;
}

// Dense occupancy bitmap of the level-0 voxels of every cloud of a pass (bounding box + K/2 margin, x fastest, 32 voxels per
// word): the first convolution tests its K^3 neighbours with one cached word read each instead of a hash probe, and the
// level-0 kernel maps use it as a presence filter in front of the hash table.
struct BmDesc {
    long long base;          // first word of this cloud's bitmap
    int x0, y0, z0;          // voxel coordinate of bit 0 (bounding-box minimum minus the margin)
    int wx, ny, nz;          // words per x row, rows per z slice, slices
};

// map[k][n] = row of (coord(n) + sign * offset(k) * ts) in the table, -1 if absent; kernel index with x fastest.
// (rk != null: the looked-up level is a rank-ordered bitmap, `sh` = log2 of its stride; else its hash table.)
// Two cheap rejections before the hash probe: a coordinate that is not a multiple of the table's tensor stride ts_in
// cannot be in it (7 of 8 candidates of a transposed map, whose offsets live on the finer stride), and for a level-0
// table the occupancy bitmap (bm != null) answers "absent" for the two thirds of a 3^3 region that are empty.
__global__ void build_map_kernel(const int* out_coords, int nout, const u64* keys, const int* vals, unsigned mask, int ksize, int ts,
                                 int sign, int ts_in, const BmDesc* __restrict__ desc, const unsigned* __restrict__ bm, int* map,
                                 const RkDesc* __restrict__ rk = nullptr, const int* __restrict__ rkrank = nullptr, int sh = 0) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (n >= nout) return;
    const int h = ksize / 2;
    const int ox = (k % ksize - h) * ts * sign, oy = ((k / ksize) % ksize - h) * ts * sign, oz = (k / (ksize * ksize) - h) * ts * sign;
    const int4 c = reinterpret_cast<const int4*>(out_coords)[n];
    const int qx = c.x + ox, qy = c.y + oy, qz = c.z + oz;
    int row = -1;
    bool probe = ((qx | qy | qz) & (ts_in - 1)) == 0;                 // tensor strides are powers of two
    if (rk) {
        map[(size_t)k * nout + n] = probe ? rk_lookup(rk[c.w], bm, rkrank, qx, qy, qz, sh) : -1;
        return;
    }
    if (probe && bm) {
        const BmDesc d = desc[c.w];
        const int bx = qx - d.x0, by = qy - d.y0, bz = qz - d.z0;
        if (bx >= 0 && bx < d.wx * 32 && by >= 0 && by < d.ny && bz >= 0 && bz < d.nz)
            probe = (bm[d.base + ((long long)bz * d.ny + by) * d.wx + (bx >> 5)] >> (bx & 31)) & 1u;
    }
    if (probe) {
        const int slot = hash_find_slot(keys, mask, pack_key(qx, qy, qz, c.w));
        row = slot < 0 ? -1 : vals[slot];
    }
    map[(size_t)k * nout + n] = row;
}

// The 3^3 stride-1 map of a level onto itself is symmetric under the point reflection of the kernel: (k, in = i, out = o) is a
// pair iff (26 - k, in = o, out = i) is.  So only the offsets k < 13 are looked up; a hit also fills map[26 - k][i] = o (each
// (k', row) entry has one possible writer: the row at coord(row) + offset(k'), so the stores do not race), k = 13 is the identity,
// and the upper half is preset to -1 by the caller.  Halves the hash probes of the largest maps.
__global__ void build_map_sym_kernel(const int* out_coords, int nout, const u64* keys, const int* vals, unsigned mask, int ts,
                                     const BmDesc* __restrict__ desc, const unsigned* __restrict__ bm, int* map,
                                     const RkDesc* __restrict__ rk = nullptr, const int* __restrict__ rkrank = nullptr, int sh = 0) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;                                  // 0..13
    if (n >= nout) return;
    if (k == 13) { map[(size_t)13 * nout + n] = n; return; }
    const int ox = (k % 3 - 1) * ts, oy = ((k / 3) % 3 - 1) * ts, oz = (k / 9 - 1) * ts;
    const int4 c = reinterpret_cast<const int4*>(out_coords)[n];
    const int qx = c.x + ox, qy = c.y + oy, qz = c.z + oz;
    int row = -1;
    bool probe = true;
    if (rk) {
        row = rk_lookup(rk[c.w], bm, rkrank, qx, qy, qz, sh);
        map[(size_t)k * nout + n] = row;
        if (row >= 0) map[(size_t)(26 - k) * nout + row] = n;
        return;
    }
    if (bm) {
        const BmDesc d = desc[c.w];
        const int bx = qx - d.x0, by = qy - d.y0, bz = qz - d.z0;
        if (bx >= 0 && bx < d.wx * 32 && by >= 0 && by < d.ny && bz >= 0 && bz < d.nz)
            probe = (bm[d.base + ((long long)bz * d.ny + by) * d.wx + (bx >> 5)] >> (bx & 31)) & 1u;
    }
    if (probe) {
        const int slot = hash_find_slot(keys, mask, pack_key(qx, qy, qz, c.w));
        row = slot < 0 ? -1 : vals[slot];
    }
    map[(size_t)k * nout + n] = row;
    if (row >= 0) map[(size_t)(26 - k) * nout + row] = n;
}

// A transposed convolution's kernel map is the strided convolution's with input and output exchanged (MinkowskiEngine asks its
// manager for the same map with is_transpose, src/convolution_transpose_cpu.cpp:75-107): up[k][f] = c  iff  down[k][c] = f.
// `up` is preset to -1; every (k, f) has at most one coarse row c, so the stores do not race.
__global__ void invert_map_kernel(const int* __restrict__ down, int ncoarse, int nfine, int* __restrict__ up) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int k = blockIdx.y;
    if (c >= ncoarse) return;
    const int f = down[(size_t)k * ncoarse + c];
    if (f >= 0) up[(size_t)k * nfine + f] = c;
}

// row offsets of the concatenated clouds of a pass (row ranges off[0..nb]), in the kernel arguments: bbox_kernel<true> turns the (n,3)
// voxel rows into (n,4) rows with the cloud index
struct CloudOff { int off[65]; };

// Rows of a level sorted by the parity class of their coordinates on the next coarser stride (8 classes, each padded with
// -1 to a multiple of 128 slots = one workgroup of the fine-level kernel).  A transposed convolution reaches a fine row
// from 1, 2, 4 or 8 of the 27 offsets - per axis: offset 0 if the coordinate is even on the coarse stride, +-1 if odd -
// and the set is the same for the whole class, so class-pure tiles skip the other offsets (sp_next_offset).  The order
// inside a class follows the atomics and does not matter: every row's sum is taken in kernel-offset order.
constexpr int PAR_PAD = 128;
__device__ __forceinline__ int parity_class(int4 c, int sh) { return ((c.x >> sh) & 1) | (((c.y >> sh) & 1) << 1) | (((c.z >> sh) & 1) << 2); }

// A workgroup takes PAR_ROWS rows (eight per thread) and adds its class counts to the global counters once: with one row per
// thread the 8 atomics per workgroup on ONE cache line - 41 k of them on a 1.3 M-row level, serialised in the L2 - were most of
// the 60 us either kernel took (the same finding as bbox_kernel's).
constexpr int PAR_ROWS = 2048;
__global__ __launch_bounds__(256) void parity_count_kernel(const int* __restrict__ coords, int n, int sh, int* __restrict__ cnt) {
    __shared__ int lc[8];
    if (threadIdx.x < 8) lc[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PAR_ROWS / 256; ++u) {
        const int i = blockIdx.x * PAR_ROWS + u * 256 + threadIdx.x;
        if (i < n) atomicAdd(&lc[parity_class(reinterpret_cast<const int4*>(coords)[i], sh)], 1);
    }
    __syncthreads();
    if (threadIdx.x < 8 && lc[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], lc[threadIdx.x]);
}

// cnt[0..7]: class sizes, cnt[8..15]: cursors (zeroed); perm: n + 8 * PAR_PAD slots preset to -1
__global__ __launch_bounds__(256) void parity_scatter_kernel(const int* __restrict__ coords, int n, int sh, int* __restrict__ cnt,
                                                             int* __restrict__ perm) {
    __shared__ int lc[8], lbase[8];
    if (threadIdx.x < 8) lc[threadIdx.x] = 0;
    __syncthreads();
    int cls[PAR_ROWS / 256], pos[PAR_ROWS / 256];
#pragma unroll
    for (int u = 0; u < PAR_ROWS / 256; ++u) {
        const int i = blockIdx.x * PAR_ROWS + u * 256 + threadIdx.x;
        cls[u] = 0; pos[u] = 0;
        if (i < n) {
            cls[u] = parity_class(reinterpret_cast<const int4*>(coords)[i], sh);
            pos[u] = atomicAdd(&lc[cls[u]], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        int base = 0;
        for (int c = 0; c < (int)threadIdx.x; ++c) base += (cnt[c] + PAR_PAD - 1) / PAR_PAD * PAR_PAD;
        lbase[threadIdx.x] = base + (lc[threadIdx.x] ? atomicAdd(&cnt[8 + threadIdx.x], lc[threadIdx.x]) : 0);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PAR_ROWS / 256; ++u) {
        const int i = blockIdx.x * PAR_ROWS + u * 256 + threadIdx.x;
        if (i < n) perm[lbase[cls[u]] + pos[u]] = i;
    }
}

// Level-0 rows grouped by the 8^3-voxel cell they lie in (cells in Morton order inside a cloud, 16 cells per axis with
// wrap-around): the rows a workgroup's 128 output rows gather are then mostly shared (a surface patch and its one-voxel
// halo) and hit in the L2 instead of each coming from the MALL / HBM, and the coarser levels - compacted in first-occurrence
// order - inherit the grouping.  Counting sort: cell histogram, scan (in-block + block totals), scatter; the order inside a
// cell follows the atomics and does not matter: a row's result does not depend on where the row sits, and the final kernel
// writes through the permutation, so the caller's row order is kept.
constexpr int CELL_SH = 3, CELL_PER_CLOUD = 4096;
constexpr int CELL_SORT_MIN_ROWS = 1 << 18;      // measured: -0.2 ms on a 1.3 M-row pass (sort included), +0.06 ms on an 88 k-row pass
__device__ __forceinline__ int cell_of(int4 c) {
    auto spread = [](unsigned v) { v &= 15u; v = (v | (v << 4)) & 0x0C3u; v = (v | (v << 2)) & 0x249u; return v; };     // abcd -> a00b00c00d
    return c.w * CELL_PER_CLOUD + (int)(spread(c.x >> CELL_SH) | (spread(c.y >> CELL_SH) << 1) | (spread(c.z >> CELL_SH) << 2));
}

__global__ void cell_count_kernel(const int* __restrict__ coords, int n, int* __restrict__ cnt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&cnt[cell_of(reinterpret_cast<const int4*>(coords)[i])], 1);
}

// cnt[1024 b .. 1024 b + 1023] -> exclusive prefix inside the block, block total -> btot[b]
__global__ __launch_bounds__(1024) void cell_scan_kernel(int* __restrict__ cnt, int* __restrict__ btot) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int v = cnt[blockIdx.x * 1024 + tid];
    int s = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(s, o); if (lane >= o) s += t; }
    if (lane == 63) wsum[wv] = s;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < wv; ++k) off += wsum[k];
    cnt[blockIdx.x * 1024 + tid] = off + s - v;
    if (tid == 1023) btot[blockIdx.x] = off + s;
}

__global__ void cell_scatter_kernel(const int* __restrict__ coords, int n, const int* __restrict__ pre, const int* __restrict__ btot,
                                    int* __restrict__ cursor, int* __restrict__ perm, int* __restrict__ sorted) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const int cell = cell_of(c);
    const int r = btot[cell >> 10] + pre[cell] + atomicAdd(&cursor[cell], 1);
    perm[r] = i;
    reinterpret_cast<int4*>(sorted)[r] = c;
}

struct SpConvArgs {
    const float* in; int ldin, cin;
    const int* map;          // [K][nout] or null (K = 1, identity)
    int K, nout;
    const float* W;          // (K, cin, cout)
    const void* Wh;          // fp16x2 planes of W * 2^s in MFMA B-fragment order (null: fp32 MFMA path), see pack_w16
    float descale;           // 1 / (2^s * SP_ASCALE)
    int cout;
    float* out; int ldout, ocoff;
    const float* aff_s;      // per output channel affine (BN folded) or null
    const float* aff_t;      // shift / bias or null
    const float* res; int ldres, rcoff;    // residual added after the affine, or null
    int relu;
    const int* rowperm;      // fp16x2 kernels: tile slot -> output row (-1 = padding), or null (slot = row)
    int nslots;              // tile slots (= nout without a permutation)
    int debug;               // timing experiments only (YOHO_SPCONV_DEBUG): 1 = no (offset, chunk) loop, 2 = no epilogue
    int norm;                // spconv16w_kernel<1>, cout == 32 only: rows /= |row| this many times in the epilogue (the feature head)
    const int* operm;        // with norm: output row -> caller's row (level-0 rows are kept in an internal order), or null
};

// Offsets that no row of a tile reaches are skipped (their rows of the A operand are all zero: the skipped MFMAs would add
// exact zeros, so the sums are bit-identical).  The mask has one bit per kernel offset; iteration is in ascending order.
__device__ __forceinline__ int sp_next_offset(unsigned& mask) {
    const int k = __builtin_ctz(mask);
    mask &= mask - 1;
    return k;
}

constexpr int SP_MAXK = 27;

// NCB = 32-channel output blocks per wave.  SPLIT = false: every wave of the workgroup owns its own 32 output rows.
// SPLIT = true (coarse levels: few rows, many channels): the four waves share one 32-row tile and split the
// (kernel offset, channel chunk) loop four ways; the partial sums meet in LDS and are added in wave order.
// The input rows of the whole kernel region are looked up once (K <= 27 indices per row, kept in LDS); the gathered
// A values and the weight fragment of step i+1 are loaded while the MFMAs of step i issue (register ping-pong).
template <int NCB, bool SPLIT>
__global__ __launch_bounds__(256) void spconv_kernel(SpConvArgs a) {
    __shared__ int srcl[4][SP_MAXK * 32];
    __shared__ float red[SPLIT ? 3 * NCB * 16 * 64 : 1];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int rbase = SPLIT ? blockIdx.x * 32 : (blockIdx.x * 4 + w) * 32;
    if (!SPLIT && rbase >= a.nout) return;
    const int cb0 = blockIdx.y * NCB;                      // this workgroup's first 32-channel output block
    const int row = rbase + li;
    const bool valid = row < a.nout;
    int* sl = srcl[w];
    for (int k = h; k < a.K; k += 2) sl[k * 32 + li] = valid ? (a.map ? a.map[(size_t)k * a.nout + row] : row) : -1;
    __builtin_amdgcn_wave_barrier();

    floatx16s acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    const int nchunk = a.cin / 32;
    const int total = a.K * nchunk;
    const int it0 = SPLIT ? (total * w) / 4 : 0, it1 = SPLIT ? (total * (w + 1)) / 4 : total;

    auto issue = [&](int it, float (&av)[16], float (&bv)[16 * NCB]) {
        const int k = it / nchunk, cc = it - k * nchunk;
        const int src = sl[k * 32 + li];
        const float* ip = a.in + (size_t)(src < 0 ? 0 : src) * a.ldin + cc * 32 + h * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src >= 0) v = *reinterpret_cast<const float4*>(ip + 4 * q);
            av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
        }
        const float* wp = a.W + ((size_t)k * a.cin + cc * 32 + h * 16) * a.cout + cb0 * 32 + li;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) bv[kk * NCB + cb] = wp[(size_t)kk * a.cout + cb * 32];
    };
    auto mma = [&](const float (&av)[16], const float (&bv)[16 * NCB]) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk], bv[kk * NCB + cb], acc[cb], 0, 0, 0);
    };
    float a0[16], a1[16], b0[16 * NCB], b1[16 * NCB];
    if (it0 < it1) issue(it0, a0, b0);
    for (int it = it0; it < it1; it += 2) {
        if (it + 1 < it1) issue(it + 1, a1, b1);
        mma(a0, b0);
        if (it + 1 < it1) {
            if (it + 2 < it1) issue(it + 2, a0, b0);
            mma(a1, b1);
        }
    }
    if (SPLIT) {
        if (w > 0) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((w - 1) * NCB + cb) * 16 + r) * 64 + lane] = acc[cb][r];
        }
        __syncthreads();
        if (w > 0) return;
#pragma unroll
        for (int ww = 0; ww < 3; ++ww)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[cb][r] += red[((ww * NCB + cb) * 16 + r) * 64 + lane];
    }
    // D[i = row][j = channel]: lane (j = lane & 31, half = lane >> 5), reg r -> row = (r & 3) + 8 (r >> 2) + 4 half
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int co = (cb0 + cb) * 32 + li;
        const float s = a.aff_s ? a.aff_s[co] : 1.f, t = a.aff_t ? a.aff_t[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = rbase + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (orow < a.nout) {
                float v = acc[cb][r] * s + t;
                if (a.res) v += a.res[(size_t)orow * a.ldres + a.rcoff + co];
                if (a.relu) v = fmaxf(v, 0.f);
                a.out[(size_t)orow * a.ldout + a.ocoff + co] = v;
            }
        }
    }
}

// fp16x2 split variant (same decomposition as spconv_kernel): every product as lo*hi + hi*lo + hi*hi on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation (3 MFMAs at 16x the fp32-MFMA rate, error <= 3 * 2^-22 per product).
// The gathered fp32 rows are split in registers (x * 16 = hi + lo; activations must stay below 4094); the weights are
// split once at load time and stored in B-fragment order
//     Wh[k][chunk32][K16 step 2][plane 2][cout block][lane = 32 kg + j][8]  =  W[k][32 chunk + 16 step + 8 kg + e][32 cb + j]
// so a fragment is one 16-byte load per lane.
typedef unsigned uintx4s __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8s __attribute__((ext_vector_type(8)));
typedef _Float16 halfx2s __attribute__((ext_vector_type(2)));
typedef float floatx2s __attribute__((ext_vector_type(2)));
constexpr float SP_ASCALE = 16.f;

__device__ __forceinline__ floatx16s mfma_sp16(uintx4s a, uintx4s b, floatx16s c) {
    union { uintx4s u; halfx8s h; } ca, cb;
    ca.u = a; cb.u = b;
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(ca.h, cb.h, c, 0, 0, 0);
}
__device__ __forceinline__ void split_pair_sp(float x0, float x1, unsigned& hi, unsigned& lo) {
    floatx2s x;
    x.x = x0 * SP_ASCALE; x.y = x1 * SP_ASCALE;
    const halfx2s h = __builtin_convertvector(x, halfx2s);
    const floatx2s r = x - __builtin_convertvector(h, floatx2s);
    const halfx2s l = __builtin_convertvector(r, halfx2s);
    __builtin_memcpy(&hi, &h, 4);
    __builtin_memcpy(&lo, &l, 4);
}

// Gathered input rows are read through a raw buffer descriptor over [in, in + 2 GiB): a lane whose region cell is empty
// uses an out-of-range offset and gets zeros without a memory access and without a branch (branches around loads make
// the compiler drain vmcnt at every join, which serialises the load pipeline).
constexpr unsigned SP_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sp_rsrc(const float* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ void sp_gather16(__amdgpu_buffer_rsrc_t rs, unsigned off, float (&av)[16]) {
    // K16 step s uses channels 32 cc + 16 s + 8 h + e: two 32-byte runs of this lane's input row (off points at 32 cc + 8 h)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uintx4s v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, ((q >> 1) * 16 + (q & 1) * 4) * 4, 0);
        __builtin_memcpy(&av[4 * q], &v, 16);      // not v.x .. v.w: hipcc 7.2 then narrows the load to one dword and replicates it
    }
}

// Coarse levels (few rows, many channels): the four waves share one 32-row tile and split the (offset, chunk) loop four
// ways, each fetching its own weight fragments; the partial sums meet in LDS and are added in wave order.
template <int NCB, int ND>
__global__ __launch_bounds__(256) void spconv16s_kernel(SpConvArgs a) {
    __shared__ int srcl[4][SP_MAXK * 32];
    __shared__ float red[3 * NCB * 16 * 64];
    __shared__ int prow[32];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, h = lane >> 5;
    const int rbase = blockIdx.x * 32;
    const int cb0 = blockIdx.y * NCB;
    const int slot = rbase + li;
    int row = -1;
    if (slot < a.nslots) row = a.rowperm ? a.rowperm[slot] : slot;
    const bool valid = row >= 0;
    if (w == 0 && h == 0) prow[li] = row;
    int* sl = srcl[w];
    unsigned actl = 0u;                                                   // offsets reached by any row of the tile (every wave computes it)
    for (int k = h; k < a.K; k += 2) {
        const int v = valid ? (a.map ? a.map[(size_t)k * a.nout + row] : row) : -1;
        sl[k * 32 + li] = v;
        const unsigned long long b = __ballot(v >= 0);
        if ((unsigned)b) actl |= 1u << (k - h);
        if (b >> 32) actl |= 2u << (k - h);
    }
    const unsigned act = __builtin_amdgcn_readfirstlane(actl) | __builtin_amdgcn_readlane(actl, 32);
    __builtin_amdgcn_wave_barrier();

    floatx16s acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    const int nchunk = a.cin / 32, ncbt = a.cout / 32;
    const int total = a.K * nchunk;
    const int it0 = (total * w) / 4, it1 = (total * (w + 1)) / 4;
    const uintx4s* Wh = reinterpret_cast<const uintx4s*>(a.Wh);

    // Loads are branch-free (empty cells: out-of-range buffer offset; steps past the end re-read the last one): the
    // compiler's vmcnt bookkeeping only keeps loads in flight across straight-line code.
    // this wave's steps: those of [it0, it1) whose offset is active, in ascending order (the fixed ranges keep the order
    // in which the partial sums meet independent of what the tile skips)
    int nit = 0, ik = 0, icc = 0, issued = 0;                            // wave-uniform position of the load pointer
    for (unsigned m = act; m;) {
        const int k = sp_next_offset(m);
        const int lo = max(it0, k * nchunk), hi = min(it1, (k + 1) * nchunk);
        if (hi > lo) {
            if (nit == 0) { ik = k; icc = lo - k * nchunk; }
            nit += hi - lo;
        }
    }
    auto next_active = [&](int k) { return __builtin_ctz(act & ~((2u << k) - 1u)); };
    const __amdgpu_buffer_rsrc_t rs = sp_rsrc(a.in);
    auto issue = [&](float (&av)[16], uintx4s (&bv)[4 * NCB]) {
        const int src = sl[ik * 32 + li];
        sp_gather16(rs, src < 0 ? SP_OOB : ((unsigned)src * (unsigned)a.ldin + icc * 32 + h * 8) * 4u, av);
        const uintx4s* wp = Wh + ((size_t)(ik * nchunk + icc) * 4 * ncbt + cb0) * 64 + lane;    // [it][step][plane][cb][lane]
#pragma unroll
        for (int sp = 0; sp < 4; ++sp)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) bv[sp * NCB + cb] = wp[((size_t)sp * ncbt + cb) * 64];
        if (++issued < nit && ++icc == nchunk) { icc = 0; ik = next_active(ik); }
    };
    auto mma = [&](const float (&av)[16], const uintx4s (&bv)[4 * NCB]) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            uintx4s ah, al;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned hh, ll;
                split_pair_sp(av[8 * st + 2 * p], av[8 * st + 2 * p + 1], hh, ll);
                ah[p] = hh; al[p] = ll;
            }
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_sp16(al, bv[(2 * st + 0) * NCB + cb], acc[cb]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_sp16(ah, bv[(2 * st + 1) * NCB + cb], acc[cb]);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_sp16(ah, bv[(2 * st + 0) * NCB + cb], acc[cb]);
        }
    };
    // register ring: the loads of step i + ND - 1 are in flight behind the MFMAs of step i (the gathers come from the
    // MALL / a remote L2, 1-2 us away, and a coarse level has only a few waves per SIMD to hide that)
    if (nit > 0) {
        float av[ND][16];
        uintx4s bv[ND][4 * NCB];
#pragma unroll
        for (int j = 0; j < ND - 1; ++j) issue(av[j], bv[j]);
        const int nmain = (nit / ND) * ND;
        for (int it = 0; it < nmain; it += ND) {
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                issue(av[(j + ND - 1) % ND], bv[(j + ND - 1) % ND]);
                mma(av[j], bv[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < ND - 1; ++j)
            if (nmain + j < nit) mma(av[j], bv[j]);              // already loaded by the ring
    }
    if (w > 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((w - 1) * NCB + cb) * 16 + r) * 64 + lane] = acc[cb][r];
    }
    __syncthreads();
    if (w > 0) return;
#pragma unroll
    for (int ww = 0; ww < 3; ++ww)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][r] += red[((ww * NCB + cb) * 16 + r) * 64 + lane];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int co = (cb0 + cb) * 32 + li;
        const float s = (a.aff_s ? a.aff_s[co] : 1.f) * a.descale, t = a.aff_t ? a.aff_t[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = prow[(r & 3) + 8 * (r >> 2) + 4 * h];
            if (orow >= 0) {
                float v = acc[cb][r] * s + t;
                if (a.res) v += a.res[(size_t)orow * a.ldres + a.rcoff + co];
                if (a.relu) v = fmaxf(v, 0.f);
                a.out[(size_t)orow * a.ldout + a.ocoff + co] = v;
            }
        }
    }
}

// Fine levels (many rows): the four waves of a workgroup own four 32-row tiles and walk the (offset, chunk) steps in
// lockstep, so the weight fragments of a step (4 NCB KiB) are shared: every thread fetches NCB 16-byte pieces two steps
// ahead, they go through a double-buffered LDS stage (one barrier per step) and each wave reads its fragments from
// there - the vector-memory pipe only carries the gathers (a quarter of the bytes of the per-wave weight loads).
// The gathered rows run NA - 1 steps ahead in a register ring.
template <int NCB, int DBG, int NA>
__device__ __forceinline__ void spconv16w_body(const SpConvArgs& a) {
    // one LDS block: region rows of the four waves | double-buffered weight stage; the epilogue lays its output tiles over it
    constexpr int SRCL_INTS = 4 * SP_MAXK * 32, BST_FRAGS = 2 * 4 * NCB * 64, EPI_LD = 36;       // EPI_LD: padded row of 32 floats
    static_assert(SRCL_INTS * 4 + BST_FRAGS * 16 >= 4 * 32 * EPI_LD * 4, "epilogue tiles must fit");
    __shared__ __attribute__((aligned(16))) char smem[SRCL_INTS * 4 + BST_FRAGS * 16];
    __shared__ int prow[4][32];
    __shared__ unsigned actm;
    int (*srcl)[SP_MAXK * 32] = reinterpret_cast<int (*)[SP_MAXK * 32]>(smem);
    uintx4s (*bst)[4 * NCB * 64] = reinterpret_cast<uintx4s (*)[4 * NCB * 64]>(smem + SRCL_INTS * 4);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int rbase = (blockIdx.x * 4 + w) * 32;
    const int cb0 = blockIdx.y * NCB;
    const int slot = rbase + li;
    int row = -1;
    if (slot < a.nslots) row = a.rowperm ? a.rowperm[slot] : slot;
    const bool valid = row >= 0;
    int* sl = srcl[w];
    if (tid == 0) actm = 0u;
    if (h == 0) prow[w][li] = row;
    __syncthreads();
    {
        // all map reads of the tile in flight at once (half h holds offsets h, h + 2, ...), then the LDS copies and the ballots
        constexpr int NV = (SP_MAXK + 1) / 2;
        int v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int k = 2 * j + h;
            v[j] = (valid && k < a.K) ? (a.map ? a.map[(size_t)k * a.nout + row] : row) : -1;
        }
        unsigned m = 0u;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int k = 2 * j + h;
            if (k < SP_MAXK) sl[k * 32 + li] = v[j];
            const unsigned long long b = __ballot(v[j] >= 0);        // low half: offset 2 j, high half: 2 j + 1
            if ((unsigned)b) m |= 1u << (2 * j);
            if (b >> 32) m |= 2u << (2 * j);
        }
        if (m && li == 0) atomicOr(&actm, m);
    }
    __syncthreads();
    const unsigned act = __builtin_amdgcn_readfirstlane(actm);        // offsets reached by any of the workgroup's 128 rows
    if (act == 0u && __syncthreads_or(valid) == 0) return;            // padding only

    floatx16s acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
    const int nchunk = a.cin / 32, ncbt = a.cout / 32;
    const int total = __builtin_popcount(act) * nchunk;
    const uintx4s* Wh = reinterpret_cast<const uintx4s*>(a.Wh);

    if (total > 0 && !(a.debug & 1)) {
        // branch-free loads, see spconv16_kernel
        unsigned amask = act, bmask = act;                                    // wave-uniform load pointers
        int ak = sp_next_offset(amask), acc_ = 0, aissued = 0;
        int bk = sp_next_offset(bmask), bcc = 0, bissued = 0;
        const __amdgpu_buffer_rsrc_t rs = sp_rsrc(a.in);
        auto loadA = [&](float (&av)[16]) {
            int src = sl[ak * 32 + li];
            if constexpr (DBG & 32) src = min(rbase + li, a.nout - 1);          // sequential rows instead of the neighbours
            if constexpr (DBG & 64) src = (rbase >> 5) & 1023;                  // one row for the whole wave
            if constexpr (DBG & 4) {
#pragma unroll
                for (int e = 0; e < 16; ++e) av[e] = __int_as_float(src + e);
            } else
            sp_gather16(rs, src < 0 ? SP_OOB : ((unsigned)src * (unsigned)a.ldin + acc_ * 32 + h * 8) * 4u, av);
            if (++aissued < total && ++acc_ == nchunk) { acc_ = 0; ak = sp_next_offset(amask); }
        };
        // stage image = [step-plane 4][cb NCB][lane 64] fragments; piece j of this thread = image index j * 256 + tid
        auto loadB = [&](uintx4s (&br)[NCB]) {
            const int it = bk * nchunk + bcc;                                 // past the end: the last step again
#pragma unroll
            for (int j = 0; j < NCB; ++j) {
                const int idx = j * 256 + tid, sp = idx / (NCB * 64), within = idx - sp * (NCB * 64);
                if constexpr (DBG & 16) { br[j][0] = it + idx; br[j][1] = sp; br[j][2] = within; br[j][3] = it; }
                else
                br[j] = Wh[(((size_t)it * 4 + sp) * ncbt + cb0) * 64 + within];
            }
            if (++bissued < total && ++bcc == nchunk) { bcc = 0; bk = sp_next_offset(bmask); }
        };
        auto storeB = [&](int buf, const uintx4s (&br)[NCB]) {
#pragma unroll
            for (int j = 0; j < NCB; ++j) bst[buf][j * 256 + tid] = br[j];
        };
        auto mma = [&](const float (&av)[16], int buf) {
            const uintx4s* bl = &bst[buf][lane];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                uintx4s ah, al, bh[NCB], bw[NCB];
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) { bh[cb] = bl[((2 * st + 0) * NCB + cb) * 64]; bw[cb] = bl[((2 * st + 1) * NCB + cb) * 64]; }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    unsigned hh, ll;
                    split_pair_sp(av[8 * st + 2 * p], av[8 * st + 2 * p + 1], hh, ll);
                    ah[p] = hh; al[p] = ll;
                }
                if constexpr (DBG & 8) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) { acc[cb][0] += __uint_as_float(al[0] ^ bh[cb][0] ^ ah[1] ^ bw[cb][1] ^ al[2] ^ ah[3] ^ bh[cb][2] ^ bw[cb][3] ^ al[1] ^ al[3] ^ ah[0] ^ ah[2]); }
                } else {
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_sp16(al, bh[cb], acc[cb]);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_sp16(ah, bw[cb], acc[cb]);
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_sp16(ah, bh[cb], acc[cb]);
                }
            }
        };
        static_assert(NA % 2 == 0, "the stage parity of ring slot j is j & 1");
        float av[NA][16];
        uintx4s br[2][NCB];
        loadB(br[0]);
        loadB(br[1]);
#pragma unroll
        for (int j = 0; j < NA - 1; ++j) loadA(av[j]);
        storeB(0, br[0]);
        // step s: barrier (stage s & 1 complete, the other one free) -> weights of s + 1 into the free stage, fetch the
        // weights of s + 2 and the rows of s + NA - 1, MFMAs of s
        const int nmain = (total / NA) * NA;
        for (int it = 0; it < nmain; it += NA) {
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                __syncthreads();
                storeB((j + 1) & 1, br[(j + 1) & 1]);
                loadB(br[j & 1]);
                loadA(av[(j + NA - 1) % NA]);
                mma(av[j], j & 1);
            }
        }
#pragma unroll
        for (int j = 0; j < NA - 1; ++j) {
            if (nmain + j < total) {                                          // uniform over the workgroup
                __syncthreads();
                storeB((j + 1) & 1, br[(j + 1) & 1]);
                loadB(br[j & 1]);
                mma(av[j], j & 1);
            }
        }
    }
    if (a.debug & 2) return;
    // Epilogue through LDS: the accumulator tile (a lane holds one channel of 16 rows) is turned into rows of 32 channels, so
    // that eight lanes move one row's 128 bytes with 16-byte accesses (residual read, affine, ReLU, store).
    __syncthreads();                                                      // every wave is done with the stage buffers
    float* et = reinterpret_cast<float*>(smem) + w * 32 * EPI_LD;
    const int er = lane >> 3, ep = lane & 7;                              // row within a group of eight, 4-channel piece
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        if (cb) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) et[((r & 3) + 8 * (r >> 2) + 4 * h) * EPI_LD + li] = acc[cb][r];
        __builtin_amdgcn_wave_barrier();
        const int co = (cb0 + cb) * 32 + 4 * ep;
        float4 sc = make_float4(a.descale, a.descale, a.descale, a.descale), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.aff_s) { const float4 t = *reinterpret_cast<const float4*>(a.aff_s + co); sc.x *= t.x; sc.y *= t.y; sc.z *= t.z; sc.w *= t.w; }
        if (a.aff_t) sh = *reinterpret_cast<const float4*>(a.aff_t + co);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int orow = prow[w][8 * g + er];
            if (orow >= 0) {
                float4 v = *reinterpret_cast<const float4*>(et + (8 * g + er) * EPI_LD + 4 * ep);
                v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                if (a.res) {
                    const float4 rr = *reinterpret_cast<const float4*>(a.res + (size_t)orow * a.ldres + a.rcoff + co);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                size_t drow = (size_t)orow;
                if constexpr (NCB == 1) {
                    if (a.norm) {
                        // the feature head: a row's 32 channels sit in the eight lanes of its group (4 each) - unit-normalise here
                        // (resunet.py:183-187, once more in fcgf_feat.py:48) instead of a pass of its own over the (n, 32) matrix
                        for (int pass = 0; pass < a.norm; ++pass) {
                            float ss = fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);      // explicit: the compiler's contraction must not differ between the two places this is written
                            ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
                            const float nr = sqrtf(ss);
                            v.x /= nr; v.y /= nr; v.z /= nr; v.w /= nr;
                        }
                        if (a.operm) drow = (size_t)a.operm[orow];
                    }
                }
                *reinterpret_cast<float4*>(a.out + drow * a.ldout + a.ocoff + co) = v;
            }
        }
    }
}

// Ring depth and register budget per variant, measured on the 15-copy pass (tools/ab_spconv_occ.sh, round 4: 8.61 -> 8.36 ms, same bits):
// with the gathers only one step ahead (ring 2) the 64-channel kernel fits 4 waves per SIMD (102 registers instead of 148 -> 3) and
// the 128-channel one 3 (156 instead of 204 -> 2), and the extra resident workgroup hides more than the deeper ring did; forcing the
// budget with the ring of 4 spills (9.36 ms), a ring of 6 at 3 waves is slower too (8.76), the 32-channel kernel does not care
// (ring 2 at 5 waves 8.60, ring 4 at 4 waves as it was)
constexpr int sp_ring(int ncb) { return ncb == 1 ? 4 : 2; }
constexpr int sp_wpe(int ncb) { return ncb == 4 ? 3 : 4; }
template <int NCB, int DBG = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(sp_wpe(NCB), sp_wpe(NCB)))) void spconv16w_kernel(SpConvArgs a) {
    spconv16w_body<NCB, DBG, sp_ring(NCB)>(a);
}
#ifdef YOHO_EXPERIMENTS
// occupancy experiments (YOHO_SPCONV_VAR, experiments build only): the same body under a register budget of WPE waves per SIMD
// (0 = none), with a gather ring of NA steps
template <int NCB, int NA, int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void spconv16w_occ_kernel(SpConvArgs a) { spconv16w_body<NCB, 0, NA>(a); }
template <int NCB>
__global__ __launch_bounds__(256) void spconv16w_ring4_kernel(SpConvArgs a) { spconv16w_body<NCB, 0, 4>(a); }      // the kernels up to round 4
#endif

// The decoder's two 1 x 1 heads in one kernel (resunet.py:181-187): f1 = relu(conv1_tr(x)) (32 NC1 -> 64 channels), out = final(f1) + bias
// (64 -> 32), rows /= |row| `norm` times, the caller's row order.  As two launches of spconv16w_kernel the 64-channel intermediate is
// written and read back once (2 x 336 MB of the 1.35 GB the two move per 1.3 M-voxel pass; both are HBM-bound).  Here it stays in LDS:
// the accumulator tile of the first head (a lane = one channel of 16 rows) is written as rows, and read back in the A-operand
// layout the gathers deliver (a lane = 8 + 8 channels of one row).  Both weight packs (24 + 8 KiB) stay in LDS for the life of the
// workgroup, which walks 128-row tiles with a stride of the grid.  The same products in the same order, the same epilogue
// expressions as the two launches: identical bits.
struct HeadsArgs {
    const float* in; int ldin;           // (n, ldin) rows; the first 32 NC1 columns are convolved
    int n;
    const void* W1; float descale1;      // pack_w16 planes, cout = 64
    const void* W2; float descale2;      // cout = 32
    const float* bias2;
    float* out;                          // (n, 32)
    int norm;
    const int* operm;                    // output row -> caller's row, or null
};
template <int NC1>
__global__ __launch_bounds__(256) void heads_fused_kernel(HeadsArgs a) {
    constexpr int W1F = NC1 * 4 * 2 * 64, W2F = 2 * 4 * 64, LD1 = 68, EPI_LD = 36;
    __shared__ uintx4s w1s[W1F];
    __shared__ uintx4s w2s[W2F];
    __shared__ __attribute__((aligned(16))) float tile[4][32 * LD1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    for (int i = tid; i < W1F; i += 256) w1s[i] = reinterpret_cast<const uintx4s*>(a.W1)[i];
    for (int i = tid; i < W2F; i += 256) w2s[i] = reinterpret_cast<const uintx4s*>(a.W2)[i];
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = sp_rsrc(a.in);
    float* et = tile[w];
    const int er = lane >> 3, ep = lane & 7;
    const int ntiles = (a.n + 127) / 128;
    // the rows of the next tile are fetched while this one is multiplied (a wave's tile is a dependent chain gather -> MFMA -> LDS -> MFMA
    // -> store, and only eight waves share a CU)
    float avn[NC1][16];
    auto fetch = [&](int t) {
        const int row = t * 128 + w * 32 + li;
        const bool valid = t < ntiles && row < a.n;
#pragma unroll
        for (int cc = 0; cc < NC1; ++cc) sp_gather16(rs, valid ? ((unsigned)row * (unsigned)a.ldin + cc * 32 + h * 8) * 4u : SP_OOB, avn[cc]);
    };
    fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int rbase = t * 128 + w * 32;
        float av[NC1][16];
#pragma unroll
        for (int cc = 0; cc < NC1; ++cc)
#pragma unroll
            for (int e = 0; e < 16; ++e) av[cc][e] = avn[cc][e];
        fetch(t + gridDim.x);
        floatx16s acc1[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[cb][r] = 0.f;
#pragma unroll
        for (int cc = 0; cc < NC1; ++cc) {
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                uintx4s ah, al, bh[2], bw[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) { bh[cb] = w1s[((cc * 4 + 2 * st + 0) * 2 + cb) * 64 + lane]; bw[cb] = w1s[((cc * 4 + 2 * st + 1) * 2 + cb) * 64 + lane]; }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    unsigned hh, ll;
                    split_pair_sp(av[cc][8 * st + 2 * p], av[cc][8 * st + 2 * p + 1], hh, ll);
                    ah[p] = hh; al[p] = ll;
                }
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc1[cb] = mfma_sp16(al, bh[cb], acc1[cb]);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc1[cb] = mfma_sp16(ah, bw[cb], acc1[cb]);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) acc1[cb] = mfma_sp16(ah, bh[cb], acc1[cb]);
            }
        }
        // f1 = relu(acc * descale + 0) as the first launch's epilogue writes it, kept as rows of 64 channels
        {
            const float sc = a.descale1, sh = 0.f;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc1[cb][r];
                    v = v * sc + sh;
                    et[((r & 3) + 8 * (r >> 2) + 4 * h) * LD1 + cb * 32 + li] = fmaxf(v, 0.f);
                }
        }
        __builtin_amdgcn_wave_barrier();
        floatx16s acc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
            float a2[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(et + li * LD1 + cc * 32 + h * 8 + (q >> 1) * 16 + (q & 1) * 4);
                a2[4 * q] = v.x; a2[4 * q + 1] = v.y; a2[4 * q + 2] = v.z; a2[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                uintx4s ah, al;
                const uintx4s bh = w2s[(cc * 4 + 2 * st + 0) * 64 + lane], bw = w2s[(cc * 4 + 2 * st + 1) * 64 + lane];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    unsigned hh, ll;
                    split_pair_sp(a2[8 * st + 2 * p], a2[8 * st + 2 * p + 1], hh, ll);
                    ah[p] = hh; al[p] = ll;
                }
                acc2 = mfma_sp16(al, bh, acc2);
                acc2 = mfma_sp16(ah, bw, acc2);
                acc2 = mfma_sp16(ah, bh, acc2);
            }
        }
        __builtin_amdgcn_wave_barrier();                                  // every lane has read its row of f1
#pragma unroll
        for (int r = 0; r < 16; ++r) et[((r & 3) + 8 * (r >> 2) + 4 * h) * EPI_LD + li] = acc2[r];
        __builtin_amdgcn_wave_barrier();
        {
            const int co = 4 * ep;
            float4 sc = make_float4(a.descale2, a.descale2, a.descale2, a.descale2), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias2) sh = *reinterpret_cast<const float4*>(a.bias2 + co);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int orow = rbase + 8 * g + er;
                if (orow < a.n) {
                    float4 v = *reinterpret_cast<const float4*>(et + (8 * g + er) * EPI_LD + 4 * ep);
                    v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
                    size_t drow = (size_t)orow;
                    if (a.norm) {
                        for (int pass = 0; pass < a.norm; ++pass) {
                            float ss = fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);      // as in spconv16w_body's epilogue
                            ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
                            const float nr = sqrtf(ss);
                            v.x /= nr; v.y /= nr; v.z /= nr; v.w /= nr;
                        }
                        if (a.operm) drow = (size_t)a.operm[orow];
                    }
                    *reinterpret_cast<float4*>(a.out + drow * 32 + co) = v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                                  // the tile is written again by the next iteration
    }
}

// Cin < 32 (the first convolution: one input channel, 5^3 / 7^3 offsets): plain fp32, one thread per (row, channel)
__global__ __launch_bounds__(256) void spconv_small_kernel(SpConvArgs a) {
    const int co = threadIdx.x % a.cout, rl = threadIdx.x / a.cout;
    const int rows_per = 256 / a.cout;
    const int row = blockIdx.x * rows_per + rl;
    if (rl >= rows_per || row >= a.nout) return;
    float acc = 0.f;
    constexpr int UB = 7;                                  // offsets per batch: independent map / feature loads in flight
    for (int k0 = 0; k0 < a.K; k0 += UB) {
        int src[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) src[u] = (k0 + u < a.K) ? (a.map ? a.map[(size_t)(k0 + u) * a.nout + row] : row) : -1;
        for (int c = 0; c < a.cin; ++c) {
            float xv[UB], wv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                xv[u] = src[u] >= 0 ? a.in[(size_t)src[u] * a.ldin + c] : 0.f;
                wv[u] = (k0 + u < a.K) ? a.W[((size_t)(k0 + u) * a.cin + c) * a.cout + co] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) acc = fmaf(xv[u], wv[u], acc);
        }
    }
    float v = acc * (a.aff_s ? a.aff_s[co] : 1.f) + (a.aff_t ? a.aff_t[co] : 0.f);
    if (a.res) v += a.res[(size_t)row * a.ldres + a.rcoff + co];
    if (a.relu) v = fmaxf(v, 0.f);
    a.out[(size_t)row * a.ldout + a.ocoff + co] = v;
}

// First convolution with the constant-one input feature (simple_yoho/fcgf_feat.py:41, one input channel, 32 outputs):
//     out[n][co] = sum over the occupied voxels of the K^3 region of W[k][0][co]
// fused with the neighbourhood lookup: a half-wave owns one output row, its 32 lanes probe the hash table for 32 kernel
// offsets at a time (ballot), then every lane (= output channel) adds the weights of the occupied offsets in kernel-index
// order from an LDS copy of W.  No K^3 x N kernel map is written or read.
constexpr int C1O_MAXK = 343;
__global__ __launch_bounds__(256) void conv1_ones_kernel(const int* __restrict__ coords, int n, const u64* __restrict__ keys, unsigned mask,
                                                         int ksize, const float* __restrict__ W, const float* __restrict__ aff_s,
                                                         const float* __restrict__ aff_t, float* __restrict__ out) {
    __shared__ float Wl[C1O_MAXK * 32];
    const int kv = ksize * ksize * ksize, hk = ksize / 2;
    for (int i = threadIdx.x; i < kv * 32; i += 256) Wl[i] = W[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, l32 = lane & 31, hw = lane >> 5;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + hw;
    const bool valid = row < n;
    const int4 c = valid ? reinterpret_cast<const int4*>(coords)[row] : make_int4(0, 0, 0, 0);
    float acc = 0.f;
    for (int k0 = 0; k0 < kv; k0 += 32) {
        const int k = k0 + l32;
        bool present = false;
        if (valid && k < kv) {
            const int ox = k % ksize - hk, oy = (k / ksize) % ksize - hk, oz = k / (ksize * ksize) - hk;
            present = hash_find_slot(keys, mask, pack_key(c.x + ox, c.y + oy, c.z + oz, c.w)) >= 0;
        }
        const unsigned long long m64 = __ballot(present);
        unsigned m = hw ? (unsigned)(m64 >> 32) : (unsigned)m64;
        while (m) {
            const int j = __ffs(m) - 1;
            m &= m - 1;
            acc += Wl[(k0 + j) * 32 + l32];
        }
    }
    if (valid) out[(size_t)row * 32 + l32] = acc * (aff_s ? aff_s[l32] : 1.f) + (aff_t ? aff_t[l32] : 0.f);
}

// Bounding boxes of the clouds of a pass.  A workgroup scans a run of <= rows_per_wg rows of ONE cloud (the clouds' row ranges are in
// the kernel arguments; workgroup -> (cloud, run) by walking the clouds' run counts) and leaves its box in part[block] = (cloud, lo,
// hi); bbox_reduce_kernel combines the blocks.  No atomics here: the first version let a run straddle clouds and flushed a thread's
// box with six atomics at the boundary - 256 threads x 6 atomics on one cache line per boundary, serialised at ~50 ns each, were
// 75 of the kernel's 80 us on a 15-cloud pass (the row loop without them: 5 us).
// FROM3: the rows come from the caller's (n,3) matrix and the (n,4) rows with the cloud index are written on the way.
template <bool FROM3>
__global__ __launch_bounds__(256) void bbox_kernel(const int* __restrict__ coords, int rows_per_wg, int* __restrict__ part, CloudOff o, int nb,
                                                   int* __restrict__ c4) {
    __shared__ int red[4][6];
    int b = 0, base = 0;
    for (; b < nb; ++b) {
        const int runs = (o.off[b + 1] - o.off[b] + rows_per_wg - 1) / rows_per_wg;
        if ((int)blockIdx.x < base + runs) break;
        base += runs;
    }
    int* p = part + 7 * blockIdx.x;
    if (b == nb) { if (threadIdx.x == 0) p[0] = -1; return; }
    const int r0 = o.off[b] + ((int)blockIdx.x - base) * rows_per_wg, r1 = min(o.off[b + 1], r0 + rows_per_wg);
    int lo[3] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int i = r0 + threadIdx.x; i < r1; i += 256) {
        int x, y, z;
        if constexpr (FROM3) {
            x = coords[3 * (size_t)i]; y = coords[3 * (size_t)i + 1]; z = coords[3 * (size_t)i + 2];
            reinterpret_cast<int4*>(c4)[i] = make_int4(x, y, z, b);
        } else {
            const int4 c = reinterpret_cast<const int4*>(coords)[i];
            x = c.x; y = c.y; z = c.z;
        }
        lo[0] = min(lo[0], x); lo[1] = min(lo[1], y); lo[2] = min(lo[2], z);
        hi[0] = max(hi[0], x); hi[1] = max(hi[1], y); hi[2] = max(hi[2], z);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int o2 = 32; o2 >= 1; o2 >>= 1) {
            lo[a] = min(lo[a], __shfl_xor(lo[a], o2));
            hi[a] = max(hi[a], __shfl_xor(hi[a], o2));
        }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w][0] = lo[0]; red[w][1] = lo[1]; red[w][2] = lo[2]; red[w][3] = hi[0]; red[w][4] = hi[1]; red[w][5] = hi[2]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int ww = 1; ww < 4; ++ww)
            for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], red[ww][a]); hi[a] = max(hi[a], red[ww][3 + a]); }
        p[0] = lo[0] <= hi[0] ? b : -1;
        p[1] = lo[0]; p[2] = lo[1]; p[3] = lo[2]; p[4] = hi[0]; p[5] = hi[1]; p[6] = hi[2];
    }
}

__global__ __launch_bounds__(1024) void bbox_reduce_kernel(const int* __restrict__ part, int nblocks, int nb, int* __restrict__ bb) {
    __shared__ int lb[64 * 6];
    for (int i = threadIdx.x; i < 64 * 6; i += 1024) lb[i] = (i % 6) < 3 ? 0x7FFFFFFF : (int)0x80000000;
    __syncthreads();
    for (int b = threadIdx.x; b < nblocks; b += 1024) {
        const int* p = part + 7 * b;
        const int cl = p[0];
        if (cl >= 0 && cl < 64) {
            atomicMin(&lb[cl * 6 + 0], p[1]); atomicMin(&lb[cl * 6 + 1], p[2]); atomicMin(&lb[cl * 6 + 2], p[3]);
            atomicMax(&lb[cl * 6 + 3], p[4]); atomicMax(&lb[cl * 6 + 4], p[5]); atomicMax(&lb[cl * 6 + 5], p[6]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb * 6; i += 1024) bb[i] = lb[i];
}

__global__ void bitmap_fill_kernel(const int* __restrict__ coords, int n, const BmDesc* __restrict__ desc, unsigned* __restrict__ bm) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    const BmDesc d = desc[c.w];
    const int bx = c.x - d.x0;
    atomicOr(bm + d.base + ((long long)(c.z - d.z0) * d.ny + (c.y - d.y0)) * d.wx + (bx >> 5), 1u << (bx & 31));
}

// Persistent workgroups (the 44 KiB weight table is loaded into LDS once per workgroup, not once per 8 rows); a half-wave
// owns a row per round.  All ceil(K^3 / 32) bitmap words of a row are requested before the first one is used.
constexpr int C1B_NIT = (C1O_MAXK + 31) / 32;
__global__ __launch_bounds__(256) void conv1_bitmap_kernel(const int* __restrict__ coords, int n, const BmDesc* __restrict__ desc,
                                                           const unsigned* __restrict__ bm, int ksize, const float* __restrict__ W,
                                                           const float* __restrict__ aff_s, const float* __restrict__ aff_t,
                                                           float* __restrict__ out) {
    __shared__ float Wl[(C1O_MAXK + 1) * 32];
    __shared__ int koff[C1B_NIT * 32];                   // offset k -> dx | dy << 8 | dz << 16 (each 0 .. K-1), -1 past the end
    const int kv = ksize * ksize * ksize, hk = ksize / 2;
    for (int i = threadIdx.x; i < kv * 32; i += 256) Wl[i] = W[i];
    if (threadIdx.x < 32) Wl[C1O_MAXK * 32 + threadIdx.x] = 0.f;
    for (int k = threadIdx.x; k < C1B_NIT * 32; k += 256)
        koff[k] = k < kv ? (k % ksize) | (((k / ksize) % ksize) << 8) | ((k / (ksize * ksize)) << 16) : -1;
    __syncthreads();
    const int lane = threadIdx.x & 63, l32 = lane & 31, hw = lane >> 5;
    const int nit = (kv + 31) / 32;
    const float sc = aff_s ? aff_s[l32] : 1.f, sh = aff_t ? aff_t[l32] : 0.f;
    for (int pair = blockIdx.x * 4 + (threadIdx.x >> 6); pair * 2 < n; pair += gridDim.x * 4) {
        const int row = pair * 2 + hw;
        const bool valid = row < n;
        const int4 c = valid ? reinterpret_cast<const int4*>(coords)[row] : make_int4(0, 0, 0, 0);
        const BmDesc d = desc[c.w];
        const unsigned* bmc = bm + d.base;
        const int bx = c.x - d.x0 - hk, by = c.y - d.y0 - hk, bz = c.z - d.z0 - hk;      // >= 0 by construction of the margin
        unsigned word[C1B_NIT];
        int shift[C1B_NIT];
#pragma unroll
        for (int it = 0; it < C1B_NIT; ++it) {
            const int ko = koff[it * 32 + l32];
            const bool use = valid && ko >= 0 && it < nit;
            const int x = bx + (ko & 255), y = by + ((ko >> 8) & 255), z = bz + (ko >> 16);
            word[it] = bmc[use ? (z * d.ny + y) * d.wx + (x >> 5) : 0];                   // a cloud's bitmap has < 2^24 words
            shift[it] = use ? (x & 31) : 32;
        }
        float acc = 0.f;
#pragma unroll
        for (int it = 0; it < C1B_NIT; ++it) {
            const bool present = shift[it] < 32 && ((word[it] >> shift[it]) & 1u);
            const unsigned long long m64 = __ballot(present);
            unsigned m = hw ? (unsigned)(m64 >> 32) : (unsigned)m64;
            while (m) {                                      // ascending offsets: fixed summation order; 4 LDS reads in flight
                float wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = m ? it * 32 + __ffs(m) - 1 : C1O_MAXK;                 // row C1O_MAXK of Wl is zero
                    m &= m - 1;
                    wv[u] = Wl[row * 32 + l32];
                }
                acc += wv[0]; acc += wv[1]; acc += wv[2]; acc += wv[3];
            }
        }
        if (valid) out[(size_t)row * 32 + l32] = acc * sc + sh;
    }
}

// The first convolution as a matrix product on the fp16 MFMA: out[row][32] = occ[row][K^3] * W[K^3][32] with the occupancy
// bits of the row's K^3 region as a 0 / 1 operand (exact in fp16) and the weights as fp16 hi + lo planes (W * 2^s = hi + lo,
// fp32 accumulation; |error| <= 2^-22 |w| per term).  The reduction axis is ordered (z, y, x) with x padded to 8: the eight
// x-neighbours of one (y, z) line are one lane's share of a 32x32x16 step, i.e. one unaligned 8-bit run of one bitmap row,
// so a step is two (y, z) lines (one per lane half) and K = 7 takes 25 steps of two MFMAs for 32 rows.  Persistent
// workgroups keep the weight planes (50 KiB for K = 7) in LDS; a wave's 2 x 25 bitmap words are requested before the first
// step.  Replaces the per-row bit scan of conv1_bitmap_kernel (1.3 ms -> see DESIGN 3.5 for 1.3 M rows).
constexpr int C1M_MAXSTEPS = 25;                                     // (7 * 7 + 1) / 2
__global__ __launch_bounds__(256) void conv1_mfma_kernel(const int* __restrict__ coords, int n, const BmDesc* __restrict__ desc,
                                                         const unsigned* __restrict__ bm, const unsigned* __restrict__ zero2, int ksize,
                                                         const uintx4s* __restrict__ planes, float descale, const float* __restrict__ aff_s, const float* __restrict__ aff_t,
                                                         float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uintx4s pl[C1M_MAXSTEPS * 2 * 64];
    const int nsteps = (ksize * ksize + 1) / 2, hk = ksize / 2;
    for (int i = threadIdx.x; i < nsteps * 128; i += 256) pl[i] = planes[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, li = lane & 31, h = lane >> 5;
    const unsigned runmask = (1u << ksize) - 1u;
    const float sc = (aff_s ? aff_s[li] : 1.f) * descale, sh = aff_t ? aff_t[li] : 0.f;
    const int ntiles = (n + 31) / 32;
    for (int tile = blockIdx.x * 4 + (threadIdx.x >> 6); tile < ntiles; tile += gridDim.x * 4) {
        const int row = tile * 32 + li;
        const bool valid = row < n;
        const int4 c = valid ? reinterpret_cast<const int4*>(coords)[row] : make_int4(0, 0, 0, 0);
        const BmDesc d = desc[c.w];
        const unsigned* bmc = bm + d.base;
        const int bx = c.x - d.x0 - hk, by = c.y - d.y0 - hk, bz = c.z - d.z0 - hk;      // >= 0 by construction of the margin
        const int wcol = bx >> 5, shift = bx & 31;
        // this lane's (y, z) lines: 2 s + h, s = 0 .. nsteps - 1
        unsigned w0[C1M_MAXSTEPS], w1[C1M_MAXSTEPS];
        {
            int dy = h, dz = 0;
#pragma unroll
            for (int st = 0; st < C1M_MAXSTEPS; ++st) {
                const bool use = valid && st < nsteps && dz < ksize;
                const unsigned* wp = use ? bmc + (((bz + dz) * d.ny + (by + dy)) * d.wx + wcol) : zero2;      // a cloud's bitmap has < 2^24 words
                w0[st] = wp[0];
                w1[st] = wp[1];                                                               // (two spare words behind the last bitmap)
                dy += 2;
                if (dy >= ksize) { dy -= ksize; ++dz; }
            }
        }
        floatx16s acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int st = 0; st < C1M_MAXSTEPS; ++st) {
            if (st < nsteps) {                                                                // uniform
                const unsigned run = (unsigned)((((unsigned long long)w1[st] << 32) | w0[st]) >> shift) & runmask;
                uintx4s af;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int b0 = __builtin_amdgcn_sbfe(run, 2 * j, 1), b1 = __builtin_amdgcn_sbfe(run, 2 * j + 1, 1);   // 0 / -1
                    af[j] = ((unsigned)b0 & 0x00003C00u) | ((unsigned)b1 & 0x3C000000u);                                   // 1.0 in fp16
                }
                acc = mfma_sp16(af, pl[(2 * st + 0) * 64 + lane], acc);
                acc = mfma_sp16(af, pl[(2 * st + 1) * 64 + lane], acc);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int orow = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (orow < n) out[(size_t)orow * 32 + li] = acc[r] * sc + sh;
        }
    }
}

static int launch_spconv(const SpConvArgs& a_in, hipStream_t s) {
    if (a_in.nout == 0) return 0;
    SpConvArgs a = a_in;
    if (!a.Wh || !a.rowperm) { a.rowperm = nullptr; a.nslots = a.nout; }     // the permutation is an optimisation of the fp16x2 kernels
    if (a.norm && !(a.Wh && a.cout == 32 && a.cin % 32 == 0 && (a.nslots + 31) / 32 >= 1024)) {
        set_error("sparse conv: the fused row normalisation exists in the 32-channel fine-level kernel only"); return YOHO_EINVAL;
    }
    const bool vec_ok = a.ldout % 4 == 0 && a.ocoff % 4 == 0 && (!a.res || (a.ldres % 4 == 0 && a.rcoff % 4 == 0));      // 16-byte epilogue accesses
    if (a.cin % 32 == 0 && a.cout % 32 == 0 && a.cout <= 256 && a.ldin % 4 == 0 && a.K <= SP_MAXK && vec_ok) {
        // Two 32-channel output blocks per wave where possible (halves the gather traffic).  Levels with fewer than ~1024
        // (row tile, channel group) units run the split variant: one unit per workgroup, the K loop over its 4 waves.
        const int ncbt = a.cout / 32, rowtiles = (a.nslots + 31) / 32;
        const int ncb = (ncbt % 2 == 0 && (long long)rowtiles * (ncbt / 2) >= 1024) ? 2 : 1;
        const bool split = (long long)rowtiles * (ncbt / ncb) < 1024;
        const dim3 blk(256);
        if (split) {
            const dim3 grid(rowtiles, ncbt / ncb);
            if (a.Wh && ncb == 2) hipLaunchKernelGGL((spconv16s_kernel<2, 2>), grid, blk, 0, s, a);
            else if (a.Wh) hipLaunchKernelGGL((spconv16s_kernel<1, 3>), grid, blk, 0, s, a);
            else if (ncb == 2) hipLaunchKernelGGL((spconv_kernel<2, true>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((spconv_kernel<1, true>), grid, blk, 0, s, a);
        } else {
            const dim3 grid((a.nslots + 127) / 128, ncbt / ncb);
#ifdef YOHO_SPCONV_ABLATE
            const int dbg = a.debug;
            if (a.Wh && ncb == 2 && (dbg & 4) && (dbg & 8) && (dbg & 16)) hipLaunchKernelGGL((spconv16w_kernel<2, 28>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 2 && (dbg & 32)) hipLaunchKernelGGL((spconv16w_kernel<2, 32>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 2 && (dbg & 64)) hipLaunchKernelGGL((spconv16w_kernel<2, 64>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 2 && (dbg & 4)) hipLaunchKernelGGL((spconv16w_kernel<2, 4>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 2 && (dbg & 8)) hipLaunchKernelGGL((spconv16w_kernel<2, 8>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 2 && (dbg & 16)) hipLaunchKernelGGL((spconv16w_kernel<2, 16>), grid, blk, 0, s, a);
            else
#endif
            // 128 output channels: all four channel blocks in one wave, so every row is gathered once instead of twice (the gathers'
            // lane requests are what bounds these kernels; measured -0.25 ms on a 1.3 M-voxel pass, no gain at 256 channels)
#ifdef YOHO_EXPERIMENTS
            // low nibble: the 64-channel kernel, second nibble: the 32-channel one, third: the 128-channel one; 0 = as shipped
            static const int var = [] { const char* e = experiment_env("YOHO_SPCONV_VAR"); return e ? std::atoi(e) : 0; }();
            const bool four = a.Wh && ncb == 2 && ncbt == 4 && rowtiles >= 1024;
            if (four && (var & 0xF00) == 0x100) hipLaunchKernelGGL((spconv16w_ring4_kernel<4>), dim3(grid.x, 1), blk, 0, s, a);
            else if (four) hipLaunchKernelGGL((spconv16w_kernel<4>), dim3(grid.x, 1), blk, 0, s, a);
            else if (a.Wh && ncb == 2 && (var & 0xF) == 1) hipLaunchKernelGGL((spconv16w_ring4_kernel<2>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 2 && (var & 0xF) == 2) hipLaunchKernelGGL((spconv16w_occ_kernel<2, 4, 4>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 2 && (var & 0xF) == 3) hipLaunchKernelGGL((spconv16w_occ_kernel<2, 6, 3>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 2 && (var & 0xF) == 4) hipLaunchKernelGGL((spconv16w_occ_kernel<2, 2, 5>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 1 && (var & 0xF0) == 0x10) hipLaunchKernelGGL((spconv16w_occ_kernel<1, 2, 5>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 1 && (var & 0xF0) == 0x20) hipLaunchKernelGGL((spconv16w_occ_kernel<1, 4, 5>), grid, blk, 0, s, a);
            else if (a.Wh && ncb == 1 && (var & 0xF0) == 0x30) hipLaunchKernelGGL((spconv16w_occ_kernel<1, 6, 4>), grid, blk, 0, s, a);
            else
#endif
            if (a.Wh && ncb == 2 && ncbt == 4 && rowtiles >= 1024)
                hipLaunchKernelGGL((spconv16w_kernel<4>), dim3(grid.x, 1), blk, 0, s, a);
            else if (a.Wh && ncb == 2) hipLaunchKernelGGL((spconv16w_kernel<2>), grid, blk, 0, s, a);
            else if (a.Wh) hipLaunchKernelGGL((spconv16w_kernel<1>), grid, blk, 0, s, a);
            else if (ncb == 2) hipLaunchKernelGGL((spconv_kernel<2, false>), grid, blk, 0, s, a);
            else hipLaunchKernelGGL((spconv_kernel<1, false>), grid, blk, 0, s, a);
        }
    } else {
        if (a.cout > 256 || a.cout < 1) { set_error("sparse conv: unsupported channel count %d", a.cout); return YOHO_EINVAL; }
        const int rows_per = 256 / a.cout;
        hipLaunchKernelGGL(spconv_small_kernel, dim3((a.nout + rows_per - 1) / rows_per), dim3(256), 0, s, a);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

__global__ void fill_ones_kernel(float* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 1.f;
}

// rows /= |row| (resunet.py:183-187), then once more (fcgf_feat.py:48).  c <= 32: a half-wave per row (the xor tree over
// 32 lanes gives the same sum as the 64-lane tree with zeros in the upper half), 8 rows per wave; else one wave per row.
__global__ __launch_bounds__(256) void row_normalize_kernel(const float* in, int n, int c, float* out, int twice, const int* __restrict__ operm) {
    const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c == 32) {
        // eight lanes per row, four channels each, the sum of squares in the order of the fused epilogue of spconv16w_kernel<1>
        // (a pass large enough for that kernel normalises there): the same bits whichever of the two runs
        const int ep = lane & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + (lane >> 3);
            const bool ok = row < n;
            float4 v = ok ? *reinterpret_cast<const float4*>(in + (size_t)row * 32 + 4 * ep) : make_float4(1.f, 0.f, 0.f, 0.f);
            for (int pass = 0; pass < (twice ? 2 : 1); ++pass) {
                float ss = fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);      // explicit: the compiler's contraction must not differ between the two places this is written
                ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
                const float nr = sqrtf(ss);
                v.x /= nr; v.y /= nr; v.z /= nr; v.w /= nr;
            }
            if (ok) *reinterpret_cast<float4*>(out + (size_t)(operm ? operm[row] : row) * 32 + 4 * ep) = v;
        }
        return;
    }
    if (c <= 32) {
        const int l32 = lane & 31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 2 + (lane >> 5);
            const bool ok = row < n && l32 < c;
            float v = ok ? in[(size_t)row * c + l32] : 0.f;
            for (int pass = 0; pass < (twice ? 2 : 1); ++pass) {
                float s = v * v;
                for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
                v = v / sqrtf(s);
            }
            if (ok) out[(size_t)(operm ? operm[row] : row) * c + l32] = v;
        }
        return;
    }
    const int row = wave;
    if (row >= n) return;
    float v = lane < c ? in[(size_t)row * c + lane] : 0.f;
    for (int pass = 0; pass < (twice ? 2 : 1); ++pass) {
        float s = v * v;
        for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
        v = v / sqrtf(s);
    }
    if (lane < c) out[(size_t)(operm ? operm[row] : row) * c + lane] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct BnAff { float* s = nullptr; float* t = nullptr; };
struct ConvW { float* w = nullptr; void* wh = nullptr; float descale = 1.f; };      // fp32 kernel, fp16x2 planes (or null)

struct FcgfNet {
    int C[5] = {0, 32, 64, 128, 256}, T[5] = {0, 64, 64, 64, 128};
    int out_ch = 32, k1 = 7, in_ch = 1, normalize = 1;
    // kernels in spec order
    ConvW conv[4];                                               // conv1..conv4
    BnAff norm[4];
    ConvW bconv[4][2];                                           // block1..4 conv1/conv2
    BnAff bnorm[4][2];
    ConvW conv_tr[3];                                            // conv4_tr, conv3_tr, conv2_tr
    BnAff norm_tr[3];
    ConvW bconv_tr[3][2];                                        // block4_tr, block3_tr, block2_tr
    BnAff bnorm_tr[3][2];
    ConvW conv1_tr;
    ConvW final_k;
    float* final_b = nullptr;
    void* c1planes = nullptr; float c1descale = 1.f;             // first convolution as an MFMA product (conv1_mfma_kernel)
    std::vector<void*> owned;
};

static int up(FcgfNet* n, const float* h, size_t cnt, float** d) {
    HIPCHK(hipMalloc((void**)d, cnt * sizeof(float)));
    n->owned.push_back(*d);
    HIPCHK(hipMemcpy(*d, h, cnt * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

static inline unsigned short f16bits(float x) {
    const _Float16 hh = (_Float16)x;
    unsigned short u;
    std::memcpy(&u, &hh, 2);
    return u;
}

// kernel (K, cin, cout) fp32 -> device copy + (if the shape fits the MFMA path) fp16x2 planes in B-fragment order
static int up_conv(FcgfNet* n, const float* h, int K, int cin, int cout, ConvW* o, bool use16) {
    int rc;
    if ((rc = up(n, h, (size_t)K * cin * cout, &o->w))) return rc;
    if (!use16 || cin % 32 || cout % 32 || K > 27) return 0;
    float wmax = 0.f;
    for (size_t i = 0; i < (size_t)K * cin * cout; ++i) wmax = std::fmax(wmax, std::fabs(h[i]));
    int ex = 0;
    if (wmax > 0.f && std::isfinite(wmax)) (void)std::frexp(wmax, &ex);
    const float wscale = std::ldexp(1.f, 10 - ex);
    o->descale = 1.f / (wscale * 16.f);
    const int nch = cin / 32, ncbt = cout / 32;
    std::vector<unsigned short> pl((size_t)K * cin * cout * 2);
    for (int k = 0; k < K; ++k)
        for (int cc = 0; cc < nch; ++cc)
            for (int st = 0; st < 2; ++st)
                for (int cb = 0; cb < ncbt; ++cb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int c = cc * 32 + st * 16 + (lane >> 5) * 8 + e, co = cb * 32 + (lane & 31);
                            const float x = h[((size_t)k * cin + c) * cout + co] * wscale;
                            const _Float16 hi = (_Float16)x;
                            const size_t it = (size_t)k * nch + cc;
                            const size_t base = (((it * 2 + st) * 2) * ncbt) * 512;              // plane 0 of (it, st)
                            pl[base + ((size_t)0 * ncbt + cb) * 512 + lane * 8 + e] = f16bits(x);
                            pl[base + ((size_t)1 * ncbt + cb) * 512 + lane * 8 + e] = f16bits(x - (float)hi);
                        }
    HIPCHK(hipMalloc(&o->wh, pl.size() * 2));
    n->owned.push_back(o->wh);
    HIPCHK(hipMemcpy(o->wh, pl.data(), pl.size() * 2, hipMemcpyHostToDevice));
    return 0;
}

static int up_bn(FcgfNet* n, const float* const* p, int c, BnAff* o) {      // p: weight, bias, running_mean, running_var
    std::vector<float> s(c), t(c);
    for (int i = 0; i < c; ++i) {
        const float sc = p[0][i] / std::sqrt(p[3][i] + 1e-5f);
        s[i] = sc; t[i] = p[1][i] - p[2][i] * sc;
    }
    int rc;
    if ((rc = up(n, s.data(), c, &o->s)) || (rc = up(n, t.data(), c, &o->t))) return rc;
    return 0;
}

void fcgf_free(FcgfNet* n) {
    if (!n) return;
    for (void* p : n->owned) (void)hipFree(p);
    delete n;
}

// tensors: host pointers in the order of yoho_amd.weights.fcgf_spec() with the num_batches_tracked entries left out
int fcgf_load(FcgfNet** out, const yoho_fcgf_config* cfg, const float* const* t, int ntensors, bool f32_kernels) {
    FcgfNet* n = new FcgfNet();
    for (int i = 1; i < 5; ++i) { n->C[i] = cfg->channels[i]; n->T[i] = cfg->tr_channels[i]; }
    n->out_ch = cfg->out_channels; n->k1 = cfg->conv1_kernel_size; n->in_ch = cfg->in_channels; n->normalize = cfg->normalize_feature;
    const int expect = 4 * (1 + 4 + 2 * (1 + 4)) + 3 * (1 + 4 + 2 * (1 + 4)) + 3;
    if (ntensors != expect) { fcgf_free(n); set_error("yoho_load_fcgf: expected %d tensors, got %d", expect, ntensors); return YOHO_EINVAL; }
    if (n->k1 % 2 == 0 || n->out_ch > 64 || n->in_ch > 31) { fcgf_free(n); set_error("yoho_load_fcgf: unsupported configuration"); return YOHO_EINVAL; }
    int ti = 0, rc = 0;
    auto fail = [&](int r) { fcgf_free(n); return r; };
    const bool use16 = !f32_kernels;                              // default: fp16x2 split MFMA; a context created under YOHO_FCGF=f32 keeps the fp32 MFMA kernels
    const int* C = n->C; const int* T = n->T;
    const int cin_enc[4] = {n->in_ch, C[1], C[2], C[3]};
    for (int l = 0; l < 4 && !rc; ++l) {
        const int kv = l == 0 ? n->k1 * n->k1 * n->k1 : 27, co = C[l + 1];
        if ((rc = up_conv(n, t[ti], kv, cin_enc[l], co, &n->conv[l], use16))) break; ti += 1;
        if ((rc = up_bn(n, t + ti, co, &n->norm[l]))) break; ti += 4;
        for (int j = 0; j < 2 && !rc; ++j) {
            if ((rc = up_conv(n, t[ti], 27, co, co, &n->bconv[l][j], use16))) break; ti += 1;
            if ((rc = up_bn(n, t + ti, co, &n->bnorm[l][j]))) break; ti += 4;
        }
    }
    if (rc) return fail(rc);
    const int cin_tr[3] = {C[4], C[3] + T[4], C[2] + T[3]}, cout_tr[3] = {T[4], T[3], T[2]};
    for (int l = 0; l < 3 && !rc; ++l) {
        if ((rc = up_conv(n, t[ti], 27, cin_tr[l], cout_tr[l], &n->conv_tr[l], use16))) break; ti += 1;
        if ((rc = up_bn(n, t + ti, cout_tr[l], &n->norm_tr[l]))) break; ti += 4;
        for (int j = 0; j < 2 && !rc; ++j) {
            if ((rc = up_conv(n, t[ti], 27, cout_tr[l], cout_tr[l], &n->bconv_tr[l][j], use16))) break; ti += 1;
            if ((rc = up_bn(n, t + ti, cout_tr[l], &n->bnorm_tr[l][j]))) break; ti += 4;
        }
    }
    if (rc) return fail(rc);
    if (use16 && n->in_ch == 1 && C[1] == 32 && n->k1 <= 7) {
        // weight planes of the first convolution in the order conv1_mfma_kernel walks: step s, lane half kg -> (y, z) line
        // 2 s + kg, element e -> x offset e (zero for e >= k1 and for the line past the end)
        const float* w = t[0];                                       // (k1^3, 1, 32), kernel index x fastest
        const int k1 = n->k1, kv = k1 * k1 * k1, nsteps = (k1 * k1 + 1) / 2;
        float wmax = 0.f;
        for (int i = 0; i < kv * 32; ++i) wmax = std::fmax(wmax, std::fabs(w[i]));
        int ex = 0;
        if (wmax > 0.f && std::isfinite(wmax)) (void)std::frexp(wmax, &ex);
        const float wscale = std::ldexp(1.f, 10 - ex);
        n->c1descale = 1.f / wscale;
        std::vector<unsigned short> pl((size_t)nsteps * 2 * 64 * 8, 0);
        for (int st = 0; st < nsteps; ++st)
            for (int lane = 0; lane < 64; ++lane) {
                const int line = 2 * st + (lane >> 5), ch = lane & 31;
                if (line >= k1 * k1) continue;
                for (int e = 0; e < k1; ++e) {
                    const float x = w[((size_t)line * k1 + e) * 32 + ch] * wscale;          // line = z * k1 + y
                    const _Float16 hi = (_Float16)x;
                    pl[(((size_t)st * 2 + 0) * 64 + lane) * 8 + e] = f16bits(x);
                    pl[(((size_t)st * 2 + 1) * 64 + lane) * 8 + e] = f16bits(x - (float)hi);
                }
            }
        HIPCHK(hipMalloc(&n->c1planes, pl.size() * 2));
        n->owned.push_back(n->c1planes);
        HIPCHK(hipMemcpy(n->c1planes, pl.data(), pl.size() * 2, hipMemcpyHostToDevice));
    }
    if ((rc = up_conv(n, t[ti], 1, C[1] + T[2], T[1], &n->conv1_tr, use16))) return fail(rc);
    ti += 1;
    if ((rc = up_conv(n, t[ti], 1, T[1], n->out_ch, &n->final_k, use16))) return fail(rc);
    ti += 1;
    if ((rc = up(n, t[ti++], n->out_ch, &n->final_b))) return fail(rc);
    *out = n;
    return 0;
}

struct Level {
    int n = 0, ts = 1;
    int* coords = nullptr;
    u64* keys = nullptr;
    int* vals = nullptr;
    unsigned mask = 0;
};

static unsigned table_cap(int n) {
    unsigned c = 64;
    while (c < 2u * (unsigned)(n > 0 ? n : 1)) c <<= 1;
    return c;
}

// bump allocator over the context workspace
struct Arena {
    char* p; size_t off = 0, cap;
    template <typename Tp> Tp* take(size_t cnt) {
        off = (off + 255) & ~(size_t)255;
        Tp* r = reinterpret_cast<Tp*>(p + off);
        off += cnt * sizeof(Tp);
        return r;
    }
};

static int build_table(const CoordSrc& src, int n, Level& L, hipStream_t s) {
    const unsigned cap = L.mask + 1;
    hipLaunchKernelGGL(hash_clear_kernel, dim3((cap + 255) / 256), dim3(256), 0, s, L.keys, L.vals, cap);
    if (n > 0) hipLaunchKernelGGL(hash_insert_min_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, n, L.keys, L.vals, L.mask);
    HIPCHK(hipGetLastError());
    return 0;
}

constexpr size_t C1BM_BUDGET = (size_t)64 << 20;
size_t fcgf_workspace_bytes(const FcgfNet* net, int n0) {
    // generous bound: every level sized like level 0
    const size_t N = (size_t)n0 + 256;
    const int k1 = net->k1 * net->k1 * net->k1;
    size_t b = 0;
    b += 4 * (N * 4 * 4 + (size_t)table_cap(n0) * 12) + 8192;                 // coords + tables
    b += ((size_t)k1 + 27 * 10) * N * 4;                                      // kernel maps
    b += 3 * (N + 8 * 128 + 256) * 4;                                         // parity-sorted row orders
    b += N * 20 + (size_t)64 * 4096 * 8 + 8192;                               // cell-sorted level-0 rows
    const int* C = net->C; const int* T = net->T;
    size_t feat = 1 + 2 * C[1] + (T[2] + C[1]) + 2 * T[2] + T[1] + net->out_ch;
    feat += 2 * C[2] + (T[3] + C[2]) + 2 * T[3] + 2 * C[3] + (T[4] + C[3]) + 2 * T[4] + 3 * C[4];
    b += feat * N * 4 + 64 * 256;
    b += C1BM_BUDGET;                                                         // occupancy bitmaps of the first convolution (hash-table path)
    return b;
}

// coords0: (n0,3) int32 device, the distinct voxels of nb clouds stored one after the other (host row offsets off[0..nb],
// null = one cloud); out: (n0, out_ch).  The clouds share every launch (cloud index = 4th key component).
// One attempt.  ws_extra: bytes on top of fcgf_workspace_bytes (the rank-ordered bitmaps of a pass whose clouds are sparse in large
// boxes: their size hangs on the bounding boxes, which are only known once the pass has started).  Returns FCGF_RETRY with *rank_need
// set when the bitmaps would eat into the budget of the maps and features behind them - nothing of the pass has been kept then, and
// the caller starts it again on a workspace grown by that much (or on the hash-table path when allow_rank is false).
static constexpr int FCGF_RETRY = 2;
static int fcgf_forward_attempt(yoho_ctx* ctx, const FcgfNet* net, const int* coords0, int n0, const int* off_host, int nb, float* out, hipStream_t s,
                                size_t ws_extra, bool allow_rank, size_t* rank_need) {
    // The gathers address a feature matrix through a 2 GiB buffer window.  Level l's widest gathered matrix has ld[l] columns; level 0
    // is known now, the coarser levels are checked as their sizes come back (they hold a fraction of the rows, so in practice the
    // level-0 matrices - 96 columns: 5.5 M voxels - are what limits a pass).
    const int gather_ld[4] = {std::max(net->T[2] + net->C[1], net->C[1]), std::max(net->T[3] + net->C[2], net->C[2]),
                              std::max(net->T[4] + net->C[3], net->C[3]), net->C[4]};
    auto window_ok = [&](int level, long long rows) {
        if (rows * gather_ld[level] * 4 >= (1ll << 31)) {
            set_error("fcgf_forward: %lld voxels at level %d exceed the 2 GiB gather window (%d columns); split the batch", rows, level, gather_ld[level]);
            return false;
        }
        return true;
    };
    if (!window_ok(0, n0)) return YOHO_EINVAL;
    int rc;
    const size_t ws_base = fcgf_workspace_bytes(net, n0);
    if ((rc = ensure_ws(ctx, ws_base + ws_extra, s))) return rc;
    Arena ar{(char*)ctx->ws.p, 0, ctx->ws.bytes};
    const int* C = net->C; const int* T = net->T;
    Level L[4];
    int* dcount = ar.take<int>(4);
    phase_mark(ctx, 1, s);
    // ---- coordinate maps
    L[0].n = n0; L[0].ts = 1; L[0].coords = ar.take<int>((size_t)n0 * 4);
    int* dbb = ar.take<int>(64 * 6);
    int* dbbpart = ar.take<int>(7 * 1100);               // per-workgroup partial boxes of bbox_kernel (<= 1024 + nb <= 1088 runs)
    // <= 1024 + nb workgroups, each a run of rows of one cloud
    CloudOff ho;
    ho.off[0] = 0; ho.off[1] = n0;
    if (off_host) for (int b = 0; b <= nb; ++b) ho.off[b] = off_host[b];
    const int bb_rpw = std::max(1024, (n0 + 1023) / 1024);
    int bb_nblk = 0;
    for (int b = 0; b < nb; ++b) bb_nblk += (ho.off[b + 1] - ho.off[b] + bb_rpw - 1) / bb_rpw;
    // (n,3) -> (n,4) rows with the cloud index, and the partial bounding boxes of the clouds on the way
    if (bb_nblk) hipLaunchKernelGGL(bbox_kernel<true>, dim3(bb_nblk), dim3(256), 0, s, coords0, bb_rpw, dbbpart, ho, nb, L[0].coords);
    HIPCHK(hipGetLastError());
    bool boxes_pending = true;                           // dbbpart holds the partial boxes of the rows as they are now
    const bool conv1_fused = net->in_ch == 1 && C[1] == 32 && net->k1 * net->k1 * net->k1 <= C1O_MAXK;
    int hbb[64 * 6];
    auto bounding_boxes = [&](const int* c4) -> int {      // per-cloud boxes of the voxel indices -> hbb (waits for the stream)
        // the boxes do not depend on the order of the rows: the partials taken while the rows were written serve the first call,
        // a second one (the table path behind a bitmap path that handed over) reduces the rows again
        if (!boxes_pending && bb_nblk) hipLaunchKernelGGL(bbox_kernel<false>, dim3(bb_nblk), dim3(256), 0, s, c4, bb_rpw, dbbpart, ho, nb, nullptr);
        boxes_pending = false;
        hipLaunchKernelGGL(bbox_reduce_kernel, dim3(1), dim3(1024), 0, s, dbbpart, bb_nblk, nb, dbb);
        HIPCHK(hipMemcpyAsync(hbb, dbb, sizeof(int) * 6 * nb, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        // the packed 64-bit voxel keys hold 19 bits per axis: indices outside +-(2^18 - 16) (16 = reach of the coarsest kernel
        // offsets) would alias other voxels and give wrong kernel maps without any error - refuse them here
        for (int b = 0; b < nb; ++b) {
            const int* bb = hbb + 6 * b;
            if (bb[0] > bb[3]) continue;                        // empty cloud
            for (int a = 0; a < 3; ++a)
                if (bb[a] < -VOX_LIM || bb[3 + a] > VOX_LIM) {
                    set_error("FCGF backbone: voxel index %d of cloud %d is outside +-%d (cloud extent / voxel size too large, or a non-finite point)",
                              bb[a] < -VOX_LIM ? bb[a] : bb[3 + a], b, VOX_LIM);
                    return YOHO_EINVAL;
                }
        }
        return 0;
    };
    int* operm = nullptr;                          // internal level-0 row -> caller's row (null: same order)
    // ---- rank-ordered bitmaps (see RkDesc): every level's coordinate map without a hash table, when every cloud fits one
    const bool force_hash = ctx->fcgf_hash_coords != 0 || !allow_rank;
    bool rank_mode = false;
    bool has_dups = false;                         // the caller's rows repeat voxels: every row is then mapped by its own probes (full maps)
    RkDesc* rkd[4] = {nullptr, nullptr, nullptr, nullptr};      // device descriptors per level [nb]
    unsigned* rkbm[4] = {nullptr, nullptr, nullptr, nullptr};
    int* rkrank[4] = {nullptr, nullptr, nullptr, nullptr};
    BmDesc hdesc[64];
    BmDesc* ddesc = nullptr;
    unsigned* dbm = nullptr;
    long long dbm_words = 0;
    RkDesc hrk[4][64];
    if (conv1_fused && !force_hash) {
        if ((rc = bounding_boxes(L[0].coords))) return rc;
        const int hk = net->k1 / 2;
        long long words[4] = {0, 0, 0, 0}, ranks[4] = {0, 0, 0, 0};
        int blocks[4] = {0, 0, 0, 0};
        bool ok = true;
        auto floor16 = [](int v) { return v >= 0 ? v / 16 * 16 : -((-v + 15) / 16 * 16); };
        for (int b = 0; b < nb && ok; ++b) {
            const int* bb = hbb + 6 * b;
            const bool empty = bb[0] > bb[3];
            const int x0 = empty ? 0 : floor16(bb[0] - hk), y0 = empty ? 0 : floor16(bb[1] - hk), z0 = empty ? 0 : floor16(bb[2] - hk);
            long long dx = empty ? 1 : (long long)bb[3] + hk + 1 - x0, dy = empty ? 1 : (long long)bb[4] + hk + 1 - y0, dz = empty ? 1 : (long long)bb[5] + hk + 1 - z0;
            for (int l = 0; l < 4; ++l) {
                const long long wx = (dx + 31) / 32;
                if (l == 0 && wx * dy * dz > (1ll << 24)) { ok = false; break; }           // > 64 MiB for one cloud: the hash-table path
                RkDesc& d = hrk[l][b];
                d.base = words[l]; d.x0 = x0; d.y0 = y0; d.z0 = z0; d.wx = (int)wx; d.ny = (int)dy; d.nz = (int)dz;
                d.nyb = (int)((dy + 7) / 8);
                d.nrank = (int)(((dz + 7) / 8) * d.nyb * wx * 64);
                d.rbase = ranks[l]; d.blk0 = blocks[l];
                words[l] += wx * dy * dz; ranks[l] += d.nrank; blocks[l] += (d.nrank + 1023) / 1024;
                dx = (dx + 1) / 2; dy = (dy + 1) / 2; dz = (dz + 1) / 2;                    // cells of the next level
            }
        }
        const size_t mark = ar.off;
        int* btot[4]; int* lcoords[4];
        if (ok) {
            // The estimate behind the workspace knows n0 only; the bitmaps and rank arrays follow the VOLUME of the boxes (sparse clouds
            // in large boxes: 15 copies of 5 k voxels over 800 x 800 x 240 cells are 660 MB of ranks).  They must fit ON TOP of that
            // estimate, or the kernel maps and features taken later run out of room: ask for a larger workspace and start again.
            size_t rb = 22 * 256 + 2 * sizeof(RkDesc) * 64 + sizeof(BmDesc) * 64 + (size_t)n0 * 4;
            for (int l = 0; l < 4; ++l) rb += ((size_t)words[l] + 2) * 4 + ((size_t)ranks[l] + 1) * 4 + ((size_t)blocks[l] + 2) * 4 + (size_t)n0 * 16 + sizeof(RkDesc) * 64;
            if (ws_base + rb > ar.cap) {
                if (rank_need) { *rank_need = rb; return FCGF_RETRY; }
                ok = false;
            }
        }
        if (ok) {
            for (int l = 0; l < 4; ++l) {
                rkd[l] = reinterpret_cast<RkDesc*>(ar.take<char>(sizeof(RkDesc) * 64));
                rkbm[l] = ar.take<unsigned>((size_t)words[l] + 2);        // + spare zero words: conv1_mfma_kernel reads word pairs
                rkrank[l] = ar.take<int>((size_t)ranks[l] + 1);
                btot[l] = ar.take<int>((size_t)blocks[l] + 2);
                lcoords[l] = ar.take<int>((size_t)n0 * 4);
            }
            ddesc = reinterpret_cast<BmDesc*>(ar.take<char>(sizeof(BmDesc) * 64));
            operm = ar.take<int>((size_t)n0);
            if (ar.off > ar.cap) ok = false;
        }
        if (ok) {
            for (int b = 0; b < nb; ++b) hdesc[b] = BmDesc{hrk[0][b].base, hrk[0][b].x0, hrk[0][b].y0, hrk[0][b].z0, hrk[0][b].wx, hrk[0][b].ny, hrk[0][b].nz};
            // (the host arrays live on this frame; the copies complete with the synchronisation behind the level sizes below)
            for (int l = 0; l < 4; ++l) {
                HIPCHK(hipMemcpyAsync(rkd[l], hrk[l], sizeof(RkDesc) * nb, hipMemcpyHostToDevice, s));
                HIPCHK(hipMemsetAsync(rkbm[l], 0, ((size_t)words[l] + 2) * 4, s));
            }
            HIPCHK(hipMemcpyAsync(ddesc, hdesc, sizeof(BmDesc) * nb, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(rk_fill_kernel, dim3((n0 + 255) / 256), dim3(256), 0, s, L[0].coords, n0, rkd[0], rkbm[0]);
            long long maxw[4] = {1, 1, 1, 1};
            int maxr[4] = {1, 1, 1, 1};
            for (int l = 0; l < 4; ++l)
                for (int b = 0; b < nb; ++b) {
                    maxw[l] = std::max(maxw[l], (long long)hrk[l][b].wx * hrk[l][b].ny * hrk[l][b].nz);
                    maxr[l] = std::max(maxr[l], hrk[l][b].nrank);
                }
            for (int l = 0; l < 3; ++l)
                hipLaunchKernelGGL(rk_coarsen_kernel, dim3((unsigned)((maxw[l + 1] + 255) / 256), nb), dim3(256), 0, s, rkd[l], rkd[l + 1], rkbm[l], rkbm[l + 1]);
            for (int l = 0; l < 4; ++l) {
                hipLaunchKernelGGL(rk_count_kernel, dim3((maxr[l] + 1023) / 1024, nb), dim3(1024), 0, s, rkd[l], rkbm[l], rkrank[l], btot[l]);
                hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, s, btot[l], blocks[l], dcount + l);
            }
            int hn[4] = {0, 0, 0, 0};
            HIPCHK(hipMemcpyAsync(hn, dcount, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            has_dups = hn[0] != n0;
            if (!has_dups) {                                   // else: duplicate voxels in the input - the hash-table path keeps the first of each
                rank_mode = true;
                for (int l = 0; l < 4; ++l) {
                    L[l].ts = 1 << l; L[l].n = hn[l];
                    if (!window_ok(l, hn[l])) return YOHO_EINVAL;
                    hipLaunchKernelGGL(rk_rows_kernel, dim3((maxr[l] + 255) / 256, nb), dim3(256), 0, s, rkd[l], rkbm[l], rkrank[l], btot[l], 1 << l, lcoords[l]);
                }
                hipLaunchKernelGGL(rk_operm_kernel, dim3((n0 + 255) / 256), dim3(256), 0, s, L[0].coords, n0, rkd[0], rkbm[0], rkrank[0], operm);
                for (int l = 0; l < 4; ++l) L[l].coords = lcoords[l];
                dbm = rkbm[0]; dbm_words = words[0];
                HIPCHK(hipGetLastError());
            }
        }
        if (!rank_mode) { ar.off = mark; operm = nullptr; ddesc = nullptr; }
    }
    if (!rank_mode && (ctx->fcgf_cell_sort > 1 || (ctx->fcgf_cell_sort == 1 && n0 >= CELL_SORT_MIN_ROWS))) {
        const int ncell = nb * CELL_PER_CLOUD, nblk = ncell / 1024;
        int* cnt = ar.take<int>((size_t)2 * ncell);            // histogram -> in-block prefix | cursors
        int* btot = ar.take<int>(nblk + 1);
        int* sorted = ar.take<int>((size_t)n0 * 4);
        operm = ar.take<int>((size_t)n0);
        if (ar.off > ar.cap) { set_error("fcgf_forward: workspace estimate too small"); return YOHO_ENOMEM; }
        HIPCHK(hipMemsetAsync(cnt, 0, sizeof(int) * 2 * (size_t)ncell, s));
        hipLaunchKernelGGL(cell_count_kernel, dim3((n0 + 255) / 256), dim3(256), 0, s, L[0].coords, n0, cnt);
        hipLaunchKernelGGL(cell_scan_kernel, dim3(nblk), dim3(1024), 0, s, cnt, btot);
        hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, s, btot, nblk, btot + nblk);
        hipLaunchKernelGGL(cell_scatter_kernel, dim3((n0 + 255) / 256), dim3(256), 0, s, L[0].coords, n0, cnt, btot, cnt + ncell, operm, sorted);
        HIPCHK(hipGetLastError());
        L[0].coords = sorted;
    }
    if (!rank_mode) {
        // (the bounding boxes again when the rank path was tried: the rows may have been cell-sorted since - same boxes, one more
        // reduction on the rare path)
        if ((rc = bounding_boxes(L[0].coords))) return rc;
        HIPCHK(hipMemsetAsync(dcount, 0, sizeof(int) * 4, s));       // [0]: raised when the level-0 insert meets a voxel twice
        for (int l = 0; l < 4; ++l) {
            L[l].ts = 1 << l;
            const int nprev = l == 0 ? n0 : L[l - 1].n;
            L[l].mask = table_cap(nprev) - 1;
            L[l].keys = ar.take<u64>(L[l].mask + 1);
            L[l].vals = ar.take<int>(L[l].mask + 1);
            if (l == 0) {
                CoordSrc src{L[0].coords, nullptr, 1.0, 1, 0, {0}, nullptr, dcount};
                if ((rc = build_table(src, n0, L[0], s))) return rc;
                // value = first row of the voxel (atomicMin); the input voxels are normally distinct
            } else {
                CoordSrc src{L[l - 1].coords, nullptr, 1.0, L[l].ts, 0, {0}};
                if ((rc = build_table(src, nprev, L[l], s))) return rc;
                L[l].coords = ar.take<int>((size_t)nprev * 4);
                int* bsum = ar.take<int>((size_t)(nprev + 1023) / 1024 + 1);
                if ((rc = launch_first_compact(src, nprev, L[l].keys, L[l].vals, L[l].mask, bsum, L[l].coords, 4, nullptr, dcount + l, s))) return rc;
                int hd[2] = {0, 0};
                HIPCHK(hipMemcpyAsync(hd, dcount + l - 1, sizeof(int) * 2, hipMemcpyDeviceToHost, s));      // l = 1: [duplicate flag, n1]
                HIPCHK(hipStreamSynchronize(s));
                L[l].n = hd[1];
                if (l == 1 && hd[0]) has_dups = true;
                if (!window_ok(l, L[l].n)) return YOHO_EINVAL;
                hipLaunchKernelGGL(hash_set_rows_kernel, dim3((L[l].n + 255) / 256), dim3(256), 0, s, L[l].coords, L[l].n, L[l].keys, L[l].vals,
                                   L[l].mask);
                HIPCHK(hipGetLastError());
            }
        }
    }
    // ---- kernel maps
    phase_mark(ctx, 2, s);
    const BmDesc* map_desc = nullptr;          // set once the level-0 occupancy bitmaps exist
    const unsigned* map_bm = nullptr;
    auto make_map = [&](const Level& outL, const Level& inL, int ksize, int ts, int sign) -> int* {
        const int kv = ksize * ksize * ksize;
        int* m = ar.take<int>((size_t)kv * outL.n);
        const bool filter = map_bm && inL.ts == 1;             // the bitmap holds the level-0 voxels
        const int li = (int)(&inL - L);                        // level that is looked up
        if (outL.n > 0 && rank_mode)
            hipLaunchKernelGGL(build_map_kernel, dim3((outL.n + 255) / 256, kv), dim3(256), 0, s, outL.coords, outL.n, (const u64*)nullptr, (const int*)nullptr,
                               0u, ksize, ts, sign, inL.ts, (const BmDesc*)nullptr, (const unsigned*)rkbm[li], m, (const RkDesc*)rkd[li], (const int*)rkrank[li], li);
        else if (outL.n > 0)
            hipLaunchKernelGGL(build_map_kernel, dim3((outL.n + 255) / 256, kv), dim3(256), 0, s, outL.coords, outL.n, inL.keys, inL.vals,
                               inL.mask, ksize, ts, sign, inL.ts, filter ? map_desc : nullptr, filter ? map_bm : nullptr, m,
                               (const RkDesc*)nullptr, (const int*)nullptr, 0);
        return m;
    };
    int* M1 = conv1_fused ? nullptr : make_map(L[0], L[0], net->k1, 1, +1);
    // occupancy bitmaps for the first convolution (skipped if a cloud's bounding box is too large: hash probes then); the rank
    // path has them already
    if (conv1_fused && !rank_mode) {
        const int hk = net->k1 / 2;
        long long words = 0;
        bool ok = true;
        for (int b = 0; b < nb && ok; ++b) {
            const int* bb = hbb + 6 * b;
            if (bb[0] > bb[3]) { hdesc[b] = BmDesc{words, 0, 0, 0, 1, 1, 1}; continue; }       // empty cloud
            const long long dx = (long long)bb[3] - bb[0] + 1 + 2 * hk, dy = (long long)bb[4] - bb[1] + 1 + 2 * hk,
                            dz = (long long)bb[5] - bb[2] + 1 + 2 * hk;
            const long long wx = (dx + 31) / 32;
            if (wx * dy * dz > (1ll << 24)) ok = false;                                        // > 64 MiB for one cloud
            hdesc[b] = BmDesc{words, bb[0] - hk, bb[1] - hk, bb[2] - hk, (int)wx, (int)dy, (int)dz};
            words += wx * dy * dz;
        }
        // optional, and budgeted: fcgf_workspace_bytes reserves 64 MiB for these bitmaps (C1BM_BUDGET).  Sparse clouds in large boxes
        // need more than that (15 x 5 k voxels over 800 x 800 x 240 cells: 310 MB) - they are taken only if the workspace has that much
        // ON TOP of the estimate, or the maps and features allocated below would run out of room ("workspace estimate too small")
        const size_t bm_bytes = (size_t)words * 4 + 8192 + sizeof(BmDesc) * 64 + 512;
        const bool fits = bm_bytes <= C1BM_BUDGET || ws_base + (bm_bytes - C1BM_BUDGET) <= ar.cap;
        if (ok && words > 0 && !fits && rank_need) { *rank_need = bm_bytes - C1BM_BUDGET; return FCGF_RETRY; }      // once more on a workspace with room for them
        if (ok && words > 0 && fits) {
            dbm_words = words;
            dbm = ar.take<unsigned>((size_t)words + 2);            // + spare words: conv1_mfma_kernel reads word pairs
            ddesc = reinterpret_cast<BmDesc*>(ar.take<char>(sizeof(BmDesc) * 64));
            HIPCHK(hipMemsetAsync(dbm, 0, ((size_t)words + 2) * 4, s));
            HIPCHK(hipMemcpyAsync(ddesc, hdesc, sizeof(BmDesc) * nb, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(bitmap_fill_kernel, dim3((n0 + 255) / 256), dim3(256), 0, s, L[0].coords, n0, ddesc, dbm);
            HIPCHK(hipStreamSynchronize(s));                  // hdesc lives on this frame
        }
    }
    if (dbm) { map_desc = ddesc; map_bm = dbm; }
    int* Msame[4]; int* Mdown[3]; int* Mup[3];
    // every map by its own probes: A/B switch, and whenever the caller's rows repeat voxels - the mirrored entries of the symmetric
    // build and the inverted maps are only ever written for the FIRST row of a voxel
    const bool full_maps = ctx->env.fcgf_full_maps || has_dups;
    for (int l = 0; l < 4; ++l) {
        if (full_maps) { Msame[l] = make_map(L[l], L[l], 3, L[l].ts, +1); continue; }
        // symmetric 3^3 map: offsets 0..12 looked up, 14..26 mirrored, 13 = identity (build_map_sym_kernel)
        Msame[l] = ar.take<int>((size_t)27 * L[l].n);
        if (ar.off > ar.cap) { set_error("fcgf_forward: workspace estimate too small"); return YOHO_ENOMEM; }
        if (L[l].n == 0) continue;
        HIPCHK(hipMemsetAsync(Msame[l] + (size_t)14 * L[l].n, 0xFF, sizeof(int) * (size_t)13 * L[l].n, s));
        const bool filter = map_bm && L[l].ts == 1;
        if (rank_mode)
            hipLaunchKernelGGL(build_map_sym_kernel, dim3((L[l].n + 255) / 256, 14), dim3(256), 0, s, L[l].coords, L[l].n, (const u64*)nullptr, (const int*)nullptr, 0u,
                               L[l].ts, (const BmDesc*)nullptr, (const unsigned*)rkbm[l], Msame[l], (const RkDesc*)rkd[l], (const int*)rkrank[l], l);
        else
            hipLaunchKernelGGL(build_map_sym_kernel, dim3((L[l].n + 255) / 256, 14), dim3(256), 0, s, L[l].coords, L[l].n, L[l].keys, L[l].vals, L[l].mask,
                               L[l].ts, filter ? map_desc : nullptr, filter ? map_bm : nullptr, Msame[l], (const RkDesc*)nullptr, (const int*)nullptr, 0);
    }
    for (int l = 0; l < 3; ++l) {
        Mdown[l] = make_map(L[l + 1], L[l], 3, L[l].ts, +1);           // strided conv: offsets on the input (finer) stride
        if (full_maps) { Mup[l] = make_map(L[l], L[l + 1], 3, L[l].ts, -1); continue; }     // transposed: coarse row at coord(fine) - offset
        // ... which is the strided map with input and output exchanged (invert_map_kernel): no second probe pass
        Mup[l] = ar.take<int>((size_t)27 * L[l].n);
        if (ar.off > ar.cap) { set_error("fcgf_forward: workspace estimate too small"); return YOHO_ENOMEM; }
        if (L[l].n == 0) continue;
        HIPCHK(hipMemsetAsync(Mup[l], 0xFF, sizeof(int) * (size_t)27 * L[l].n, s));
        if (L[l + 1].n > 0)
            hipLaunchKernelGGL(invert_map_kernel, dim3((L[l + 1].n + 255) / 256, 27), dim3(256), 0, s, Mdown[l], L[l + 1].n, L[l].n, Mup[l]);
    }
    // parity-sorted row orders of levels 0..2 for the transposed convolutions
    int* perm[3]; int nperm[3];
    for (int l = 0; l < 3; ++l) {
        nperm[l] = L[l].n + 8 * PAR_PAD;
        perm[l] = ar.take<int>((size_t)nperm[l]);
        int* pc = ar.take<int>(16);
        if (ar.off > ar.cap) { set_error("fcgf_forward: workspace estimate too small"); return YOHO_ENOMEM; }
        if (L[l].n == 0) continue;
        HIPCHK(hipMemsetAsync(perm[l], 0xFF, sizeof(int) * (size_t)nperm[l], s));
        HIPCHK(hipMemsetAsync(pc, 0, sizeof(int) * 16, s));
        hipLaunchKernelGGL(parity_count_kernel, dim3((L[l].n + PAR_ROWS - 1) / PAR_ROWS), dim3(256), 0, s, L[l].coords, L[l].n, l, pc);
        hipLaunchKernelGGL(parity_scatter_kernel, dim3((L[l].n + PAR_ROWS - 1) / PAR_ROWS), dim3(256), 0, s, L[l].coords, L[l].n, l, pc, perm[l]);
    }
    HIPCHK(hipGetLastError());
    // ---- features
    float* ones = ar.take<float>((size_t)n0 * net->in_ch);
    hipLaunchKernelGGL(fill_ones_kernel, dim3((n0 * net->in_ch + 255) / 256), dim3(256), 0, s, ones, n0 * net->in_ch);
    float* x[4]; float* tmp[4]; float* cat[3]; float* enc3;
    const int catw[3] = {T[2] + C[1], T[3] + C[2], T[4] + C[3]}, catoff[3] = {T[2], T[3], T[4]};
    for (int l = 0; l < 4; ++l) {
        x[l] = ar.take<float>((size_t)L[l].n * C[l + 1]);
        tmp[l] = ar.take<float>((size_t)L[l].n * C[l + 1]);
        if (l < 3) cat[l] = ar.take<float>((size_t)L[l].n * catw[l]);
    }
    enc3 = ar.take<float>((size_t)L[3].n * C[4]);
    if (ar.off > ar.cap) { set_error("fcgf_forward: workspace estimate too small (%zu > %zu)", ar.off, ar.cap); return YOHO_ENOMEM; }

    int conv_cat = -1;                          // phase category of the next conv() calls (yoho_phase_read)
    int conv_norm = 0;                          // > 0: the next conv() normalises its output rows in its epilogue (and hands them back in the caller's order)
    auto conv = [&](const float* in, int ldin, int cin, const int* map, int K, int nout, const ConvW& W, int cout, float* o, int ldout,
                    int ocoff, const BnAff* bn, const float* bias, const float* res, int ldres, int rcoff, int relu,
                    const int* rowperm = nullptr, int nslots = 0) -> int {
        phase_mark(ctx, conv_cat, s);
        // fp16 MFMA flops the launch issues: every 32-row tile walks all K offsets x cin x cout, 3 split products (the
        // parity-sorted transposed convolutions skip offsets by data: not counted)
        if (W.wh && !rowperm && cin % 32 == 0 && cout % 32 == 0) phase_work(ctx, conv_cat, 2.0 * 3.0 * (double)((nout + 31) / 32 * 32) * K * cin * cout);
        SpConvArgs a;
        a.rowperm = rowperm; a.nslots = rowperm ? nslots : nout;
        a.in = in; a.ldin = ldin; a.cin = cin; a.map = map; a.K = K; a.nout = nout; a.W = W.w; a.Wh = W.wh; a.descale = W.descale; a.cout = cout;
        a.out = o; a.ldout = ldout; a.ocoff = ocoff; a.aff_s = bn ? bn->s : nullptr; a.aff_t = bn ? bn->t : bias;
        a.res = res; a.ldres = ldres; a.rcoff = rcoff; a.relu = relu;
        a.norm = conv_norm; a.operm = conv_norm ? operm : nullptr;
        a.debug = ctx->env.spconv_debug;
        return launch_spconv(a, s);
    };
    // BasicBlockBN: out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x), written at column `ocoff` of `o`
    auto block = [&](int l, const float* xin, int c, const ConvW* Wc, const BnAff* bnc, float* scratch, float* o, int ldout, int ocoff) -> int {
        int r;
        if ((r = conv(xin, c, c, Msame[l], 27, L[l].n, Wc[0], c, scratch, c, 0, &bnc[0], nullptr, nullptr, 0, 0, 1))) return r;
        return conv(scratch, c, c, Msame[l], 27, L[l].n, Wc[1], c, o, ldout, ocoff, &bnc[1], nullptr, xin, c, 0, 1);
    };

    // encoder (resunet.py:142-160).  The block outputs land in the decoder's concatenation buffers (right-hand columns).
    const int k1v = net->k1 * net->k1 * net->k1;
    phase_mark(ctx, 3, s);
    conv_cat = 3;
    if (conv1_fused) {
        if (dbm && net->c1planes)
            hipLaunchKernelGGL(conv1_mfma_kernel, dim3(std::min((n0 + 127) / 128, 3 * (ctx->nCU > 0 ? ctx->nCU : 256))), dim3(256), 0, s, L[0].coords, n0, ddesc, dbm,
                               dbm + dbm_words, net->k1, reinterpret_cast<const uintx4s*>(net->c1planes), net->c1descale, net->norm[0].s, net->norm[0].t, x[0]);
        else if (dbm)
            hipLaunchKernelGGL(conv1_bitmap_kernel, dim3(std::min((n0 + 7) / 8, 3 * (ctx->nCU > 0 ? ctx->nCU : 256))), dim3(256), 0, s, L[0].coords, n0, ddesc, dbm, net->k1, net->conv[0].w,
                               net->norm[0].s, net->norm[0].t, x[0]);
        else
            hipLaunchKernelGGL(conv1_ones_kernel, dim3((n0 + 7) / 8), dim3(256), 0, s, L[0].coords, n0, L[0].keys, L[0].mask, net->k1,
                               net->conv[0].w, net->norm[0].s, net->norm[0].t, x[0]);
        HIPCHK(hipGetLastError());
    } else if ((rc = conv(ones, net->in_ch, net->in_ch, M1, k1v, n0, net->conv[0], C[1], x[0], C[1], 0, &net->norm[0], nullptr, nullptr, 0, 0, 0)))
        return rc;
    conv_cat = 4;
    if ((rc = block(0, x[0], C[1], net->bconv[0], net->bnorm[0], tmp[0], cat[0], catw[0], catoff[0]))) return rc;
    for (int l = 1; l < 4; ++l) {
        const float* in = cat[l - 1] + catoff[l - 1];
        conv_cat = 7 + l;
        if ((rc = conv(in, catw[l - 1], C[l], Mdown[l - 1], 27, L[l].n, net->conv[l], C[l + 1], x[l], C[l + 1], 0, &net->norm[l], nullptr, nullptr,
                       0, 0, 0))) return rc;
        float* o = l < 3 ? cat[l] : enc3;
        conv_cat = 4 + l;
        if ((rc = block(l, x[l], C[l + 1], net->bconv[l], net->bnorm[l], tmp[l], o, l < 3 ? catw[l] : C[4], l < 3 ? catoff[l] : 0))) return rc;
    }
    // decoder (resunet.py:162-181): conv_tr -> norm -> block -> left-hand columns of the concatenation buffer
    const float* din = enc3; int dld = C[4], dcin = C[4];
    for (int j = 0; j < 3; ++j) {
        const int l = 2 - j;                       // output level of conv{4,3,2}_tr
        const int co = j == 0 ? T[4] : (j == 1 ? T[3] : T[2]);
        float* u = ar.take<float>((size_t)L[l].n * co);
        float* sc = ar.take<float>((size_t)L[l].n * co);
        if (ar.off > ar.cap) { set_error("fcgf_forward: workspace estimate too small"); return YOHO_ENOMEM; }
        conv_cat = 11 + l;
        if ((rc = conv(din, dld, dcin, Mup[l], 27, L[l].n, net->conv_tr[j], co, u, co, 0, &net->norm_tr[j], nullptr, nullptr, 0, 0, 0,
                       ctx->fcgf_parity_sort ? perm[l] : nullptr, nperm[l]))) return rc;
        conv_cat = 4 + l;
        if ((rc = block(l, u, co, net->bconv_tr[j], net->bnorm_tr[j], sc, cat[l], catw[l], 0))) return rc;
        din = cat[l]; dld = catw[l]; dcin = catw[l];
    }
    float* f1 = ar.take<float>((size_t)n0 * T[1]);
    float* f2 = ar.take<float>((size_t)n0 * net->out_ch);
    if (ar.off > ar.cap) { set_error("fcgf_forward: workspace estimate too small"); return YOHO_ENOMEM; }
    conv_cat = 14;
    // both heads in one launch where the fused kernel exists (96 -> 64 -> 32 channels, fp16x2 packs, a pass large enough for the fused
    // normalisation): the 64-channel intermediate never leaves the CU (YOHO_FCGF_HEADS=staged: the two launches below)
    const bool heads_fused = !ctx->env.fcgf_heads_staged && !ctx->env.fcgf_norm_staged && net->out_ch == 32 && T[1] == 64 && catw[0] == 96 &&
                             net->conv1_tr.wh && net->final_k.wh && (n0 + 31) / 32 >= 1024 && !ctx->env.spconv_debug;
    if (heads_fused) {
        phase_mark(ctx, conv_cat, s);
        HeadsArgs ha{cat[0], catw[0], n0, net->conv1_tr.wh, net->conv1_tr.descale, net->final_k.wh, net->final_k.descale, net->final_b, out,
                     net->normalize ? 2 : 1, operm};
        const int ntl = (n0 + 127) / 128;
        hipLaunchKernelGGL((heads_fused_kernel<3>), dim3(std::min(ntl, 2 * (ctx->nCU > 0 ? ctx->nCU : 256))), dim3(256), 0, s, ha);
        HIPCHK(hipGetLastError());
        phase_mark(ctx, -1, s);
        return 0;
    }
    if ((rc = conv(cat[0], catw[0], catw[0], nullptr, 1, n0, net->conv1_tr, T[1], f1, T[1], 0, nullptr, nullptr, nullptr, 0, 0, 1))) return rc;
    // the feature head: 1 x 1 convolution to out_ch, rows /= |row| (once more for fcgf_feat.py:48), the caller's row order.  Fused into
    // the convolution's epilogue where that kernel exists (32 features, fp16x2 fine-level kernel: passes of >= 32768 voxels); the
    // pass over the (n0, 32) matrix was 0.16 ms per 15-copy pass
    const bool fuse_norm = !ctx->env.fcgf_norm_staged && net->out_ch == 32 && net->final_k.wh && T[1] % 32 == 0 && (n0 + 31) / 32 >= 1024;
    conv_norm = fuse_norm ? (net->normalize ? 2 : 1) : 0;
    if ((rc = conv(f1, T[1], T[1], nullptr, 1, n0, net->final_k, net->out_ch, fuse_norm ? out : f2, net->out_ch, 0, nullptr, net->final_b, nullptr, 0, 0, 0))) return rc;
    conv_norm = 0;
    if (!fuse_norm)
        hipLaunchKernelGGL(row_normalize_kernel, dim3(net->out_ch == 32 ? (n0 + 127) / 128 : (net->out_ch < 32 ? (n0 + 31) / 32 : (n0 + 3) / 4)), dim3(256), 0, s, f2, n0, net->out_ch, out,
                           net->normalize ? 1 : 0, operm);
    HIPCHK(hipGetLastError());
    phase_mark(ctx, -1, s);
    return 0;
}

int fcgf_forward(yoho_ctx* ctx, const FcgfNet* net, const int* coords0, int n0, const int* off_host, int nb, float* out, hipStream_t s) {
    if (nb < 1 || nb > 64) { set_error("fcgf_forward: 1..64 clouds per call"); return YOHO_EINVAL; }
    if (n0 == 0) return 0;
    size_t need = 0;
    int rc = fcgf_forward_attempt(ctx, net, coords0, n0, off_host, nb, out, s, 0, true, &need);
    if (rc != FCGF_RETRY) return rc;
    // the bitmaps of this pass do not fit beside the budget of the maps and features: once more on a workspace with room for both (it
    // stays that size, so the next pass of the same shape starts there), and on the hash tables if even that cannot be had
    rc = fcgf_forward_attempt(ctx, net, coords0, n0, off_host, nb, out, s, need, true, nullptr);
    if (rc != YOHO_ENOMEM) return rc;
    rc = fcgf_forward_attempt(ctx, net, coords0, n0, off_host, nb, out, s, 0, false, nullptr);
    if (rc == 0) clear_error();             // recovered: the failed allocation's message must not outlive the pass that succeeded
    return rc;
}

// voxelisation (fcgf_feat.py:33-43): first point of every voxel in input order -> sel (ascending), integer coordinates
// the selected points, rotated like the voxelisation saw them, as fp32 (the reference's pcd[sel].float())
// m_dev (or null): the row count lives on the device (batched voxelisation: no host round trip between its stages)
__global__ void rotate_sel_kernel(CoordSrc src, const int64_t* __restrict__ sel, int m, float* __restrict__ out, const int* __restrict__ m_dev) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (m_dev ? *m_dev : m)) return;
    double p0, p1, p2;
    point_of(src, (int)sel[i], p0, p1, p2);
    out[3 * (size_t)i] = (float)p0; out[3 * (size_t)i + 1] = (float)p1; out[3 * (size_t)i + 2] = (float)p2;
}

int fcgf_rotate_select(const double* pts, const double* R_host, const int64_t* sel, int m, float* out, hipStream_t s) {
    if (m == 0) return 0;
    CoordSrc src{nullptr, pts, 1.0, 1, R_host ? 1 : 0, {0}};
    if (R_host) for (int i = 0; i < 9; ++i) src.R[i] = R_host[i];
    hipLaunchKernelGGL(rotate_sel_kernel, dim3((m + 255) / 256), dim3(256), 0, s, src, sel, m, out, (const int*)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

// voxelisation (fcgf_feat.py:33-43): first point of every voxel in input order -> sel (ascending), integer coordinates.
// R_host (9 doubles, row major) or null: the points are rotated (p' = R p, f64) on the fly; pts_sel (n,3) f32 or null
// receives the rotated selected points.
int fcgf_voxelize(yoho_ctx* ctx, const double* pts, int n, const double* R_host, double voxel, int64_t* sel, int* coords, float* pts_sel,
                  int* count_host, hipStream_t s) {
    if (n == 0) { *count_host = 0; return 0; }
    int rc;
    const unsigned cap = table_cap(n);
    if ((rc = ensure_ws(ctx, (size_t)cap * 12 + (size_t)n / 256 + 8192, s))) return rc;
    Arena ar{(char*)ctx->ws.p, 0, ctx->ws.bytes};
    Level L;
    L.mask = cap - 1; L.keys = ar.take<u64>(cap); L.vals = ar.take<int>(cap);
    int* dcount = ar.take<int>(2);                       // [0] number of voxels, [1] out-of-range flag
    HIPCHK(hipMemsetAsync(dcount, 0, 2 * sizeof(int), s));
    CoordSrc src{nullptr, pts, voxel, 1, R_host ? 1 : 0, {0}, dcount + 1};
    if (R_host) for (int i = 0; i < 9; ++i) src.R[i] = R_host[i];
    if ((rc = build_table(src, n, L, s))) return rc;
    int* bsum = ar.take<int>((size_t)(n + 1023) / 1024 + 1);
    if ((rc = launch_first_compact(src, n, L.keys, L.vals, L.mask, bsum, coords, 3, sel, dcount, s))) return rc;
    int hc[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(hc, dcount, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *count_host = hc[0];
    if (hc[1]) {
        *count_host = 0;
        set_error("voxelisation: a point's voxel index is outside +-%d (cloud extent / voxel size too large, or a non-finite point)", VOX_LIM);
        return YOHO_EINVAL;
    }
    if (pts_sel) return fcgf_rotate_select(pts, R_host, sel, *count_host, pts_sel, s);
    return 0;
}

// nb rotated copies of one cloud in one call: the stages of all copies are queued back to back (one hash table, block sums and
// counters per copy in the workspace) and the nb voxel counts come back with ONE read-back.  Outputs are laid out with n rows per
// copy: sel (nb, n), coords (nb, n, 3), pts_sel (nb, n, 3) or null; counts_host (nb).
// The batched voxelisation through rank-ordered bitmaps (RkDesc) instead of one hash table per copy: a copy's occupancy bitmap over a
// conservative box (the rotated corners of the cloud's bounds, two voxels of margin), ranks by prefix popcount, first[row] = smallest
// point index by one atomicMin per point into a dense array that stays in the L2 - the tables took a CAS and an atomicMin per point
// into 180 MB (4.5 M points of a 15-copy pass: 0.41 ms for the inserts alone).  The compaction in first-occurrence order is the table
// path's (vox_count / vox_scan / vox_scatter with the lookup swapped), so the outputs are the same rows in the same order.
// Returns 0 = done, < 0 = error, 1 = not applicable (a copy too large for a bitmap, non-finite points, indices near the key range):
// the caller runs the table path, which also owns the exact range check and its error message.
static int voxelize_batch_rank(yoho_ctx* ctx, const double* pts, int n, const double* R_host, int nb, double voxel, int64_t* sel, int* coords,
                               float* pts_sel, int* counts_host, hipStream_t s) {
    int rc;
    const int gblk = std::min(256, (n + 255) / 256);
    if ((rc = ensure_ws(ctx, 64 * 1024, s))) return rc;
    phase_mark(ctx, 0, s);
    double hpart[256 * 6];
    hipLaunchKernelGGL(aabb_kernel, dim3(gblk), dim3(256), 0, s, pts, n, reinterpret_cast<double*>(ctx->ws.p));
    HIPCHK(hipMemcpyAsync(hpart, ctx->ws.p, sizeof(double) * 6 * gblk, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int g = 0; g < gblk; ++g)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], hpart[6 * g + a]); hi[a] = std::max(hi[a], hpart[6 * g + 3 + a]); }
    for (int a = 0; a < 3; ++a) if (!(lo[a] > -1e290 && hi[a] < 1e290 && lo[a] <= hi[a])) return 1;
    RkDesc hd[64];
    long long words = 0, ranks = 0;
    int blocks = 0, maxr = 1;
    for (int b = 0; b < nb; ++b) {
        const double* R = R_host + 9 * (size_t)b;
        double bl[3] = {1e300, 1e300, 1e300}, bh[3] = {-1e300, -1e300, -1e300};
        for (int c = 0; c < 8; ++c) {
            const double px = (c & 1) ? hi[0] : lo[0], py = (c & 2) ? hi[1] : lo[1], pz = (c & 4) ? hi[2] : lo[2];
            for (int a = 0; a < 3; ++a) {
                const double v = R[3 * a] * px + R[3 * a + 1] * py + R[3 * a + 2] * pz;
                bl[a] = std::min(bl[a], v); bh[a] = std::max(bh[a], v);
            }
        }
        long long vlo[3], dim[3];
        for (int a = 0; a < 3; ++a) {
            const double l = std::floor(bl[a] / voxel) - 2.0, h = std::floor(bh[a] / voxel) + 2.0;
            if (!(l > -(double)VOX_LIM && h < (double)VOX_LIM)) return 1;
            vlo[a] = (long long)l; dim[a] = (long long)h - (long long)l + 1;
        }
        const long long wx = (dim[0] + 31) / 32;
        if (wx * dim[1] * dim[2] > (1ll << 24)) return 1;
        RkDesc& d = hd[b];
        d.base = words; d.x0 = (int)vlo[0]; d.y0 = (int)vlo[1]; d.z0 = (int)vlo[2]; d.wx = (int)wx; d.ny = (int)dim[1]; d.nz = (int)dim[2];
        d.nyb = (int)((dim[1] + 7) / 8);
        d.nrank = (int)(((dim[2] + 7) / 8) * d.nyb * wx * 64);
        d.rbase = ranks; d.blk0 = blocks;
        words += wx * dim[1] * dim[2]; ranks += d.nrank; blocks += (d.nrank + 1023) / 1024;
        maxr = std::max(maxr, d.nrank);
    }
    const int nblk = (n + 1023) / 1024;
    const size_t need = (size_t)words * 4 + (size_t)ranks * 4 + (size_t)blocks * 4 + (size_t)nb * n * 4 + ((size_t)nblk + 1) * 4 * nb + sizeof(RkDesc) * 64 + 65536;
    if ((rc = ensure_ws(ctx, need, s))) { if (rc == YOHO_ENOMEM) clear_error(); return rc == YOHO_ENOMEM ? 1 : rc; }      // multi-GB ranks that cannot be had: the table path needs 12 bytes per point and copy
    Arena ar{(char*)ctx->ws.p, 0, ctx->ws.bytes};
    int* dcount = ar.take<int>(2 * (size_t)nb + 2);      // per copy: [0] number of voxels, [1] flag: a voxel outside the bitmap / the key range
    RkDesc* dd = reinterpret_cast<RkDesc*>(ar.take<char>(sizeof(RkDesc) * 64));
    unsigned* bm = ar.take<unsigned>((size_t)words + 2);
    int* rank = ar.take<int>((size_t)ranks + 1);
    int* btot = ar.take<int>((size_t)blocks + 2);
    int* first = ar.take<int>((size_t)nb * n);
    int* bsum = ar.take<int>(((size_t)nblk + 1) * nb);
    if (ar.off > ar.cap) { set_error("fcgf_voxelize_batch: workspace estimate too small"); return YOHO_ENOMEM; }
    HIPCHK(hipMemcpyAsync(dd, hd, sizeof(RkDesc) * nb, hipMemcpyHostToDevice, s));      // hd lives until the synchronisation below
    HIPCHK(hipMemsetAsync(dcount, 0, sizeof(int) * (2 * nb + 2), s));
    HIPCHK(hipMemsetAsync(bm, 0, ((size_t)words + 2) * 4, s));
    HIPCHK(hipMemsetAsync(first, 0x7F, (size_t)nb * n * 4, s));
    auto batch = [&](int b0, int nbc) {
        VoxBatch a;
        a.pts = pts; a.n = n; a.voxel = voxel;
        for (int b = 0; b < nbc; ++b) for (int i = 0; i < 9; ++i) a.R[b][i] = R_host[9 * (size_t)(b0 + b) + i];
        a.keys = nullptr; a.vals = nullptr; a.cap = 0;
        a.bsum = bsum + (size_t)b0 * (nblk + 1); a.nblk = nblk;
        a.dcount = dcount + 2 * (size_t)b0;
        a.coords = coords + (size_t)b0 * n * 3; a.sel = sel + (size_t)b0 * n; a.pts_sel = pts_sel ? pts_sel + (size_t)b0 * n * 3 : nullptr;
        a.rk = dd; a.bm = bm; a.rank = rank; a.first = first; a.b0 = b0;
        return a;
    };
    for (int b0 = 0; b0 < nb; b0 += VOX_BATCH) {
        const int nbc = std::min(VOX_BATCH, nb - b0);
        hipLaunchKernelGGL(vox_fill_kernel, dim3((n + 255) / 256, nbc), dim3(256), 0, s, batch(b0, nbc), bm);
    }
    hipLaunchKernelGGL(rk_count_kernel, dim3((maxr + 1023) / 1024, nb), dim3(1024), 0, s, dd, bm, rank, btot);
    hipLaunchKernelGGL(block_scan_kernel, dim3(1), dim3(1024), 0, s, btot, blocks, dcount + 2 * nb);
    hipLaunchKernelGGL(rk_finish_kernel, dim3((maxr + 255) / 256, nb), dim3(256), 0, s, dd, rank, btot);
    for (int b0 = 0; b0 < nb; b0 += VOX_BATCH) {
        const int nbc = std::min(VOX_BATCH, nb - b0);
        hipLaunchKernelGGL(vox_first_kernel, dim3((n + 255) / 256, nbc), dim3(256), 0, s, batch(b0, nbc));
    }
    for (int b0 = 0; b0 < nb; b0 += VOX_BATCH) {
        const int nbc = std::min(VOX_BATCH, nb - b0);
        const VoxBatch a = batch(b0, nbc);
        hipLaunchKernelGGL(vox_count_kernel, dim3(nblk, nbc), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(vox_scan_kernel, dim3(nbc), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(vox_scatter_kernel, dim3(nblk, nbc), dim3(1024), 0, s, a);
    }
    HIPCHK(hipGetLastError());
    phase_mark(ctx, -1, s);
    int hc[130];
    HIPCHK(hipMemcpyAsync(hc, dcount, sizeof(int) * 2 * nb, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int b = 0; b < nb; ++b) if (hc[2 * b + 1]) return 1;           // outside the box or the key range: the table path decides
    for (int b = 0; b < nb; ++b) counts_host[b] = hc[2 * b];
    return 0;
}

int fcgf_voxelize_batch(yoho_ctx* ctx, const double* pts, int n, const double* R_host, int nb, double voxel, int64_t* sel, int* coords,
                        float* pts_sel, int* counts_host, hipStream_t s) {
    if (nb < 1 || nb > 64) { set_error("fcgf_voxelize_batch: 1..64 copies per call"); return YOHO_EINVAL; }
    for (int b = 0; b < nb; ++b) counts_host[b] = 0;
    if (n == 0) return 0;
    int rc;
    if (!ctx->fcgf_hash_coords) {
        rc = voxelize_batch_rank(ctx, pts, n, R_host, nb, voxel, sel, coords, pts_sel, counts_host, s);
        if (rc <= 0) return rc;
    }
    const unsigned cap = table_cap(n);
    const int nblk = (n + 1023) / 1024;
    const size_t per = (size_t)cap * 12 + ((size_t)nblk + 1) * 4 + 1024;
    if ((rc = ensure_ws(ctx, per * nb + 8192, s))) return rc;
    Arena ar{(char*)ctx->ws.p, 0, ctx->ws.bytes};
    int* dcount = ar.take<int>(2 * (size_t)nb);          // per copy: [0] number of voxels, [1] out-of-range flag
    u64* keys = ar.take<u64>((size_t)cap * nb);
    int* vals = ar.take<int>((size_t)cap * nb);
    int* bsum = ar.take<int>(((size_t)nblk + 1) * nb);
    if (ar.off > ar.cap) { set_error("fcgf_voxelize_batch: workspace estimate too small"); return YOHO_ENOMEM; }
    phase_mark(ctx, 0, s);
    HIPCHK(hipMemsetAsync(dcount, 0, 2 * sizeof(int) * nb, s));
    {
        const size_t total = (size_t)cap * nb;
        hipLaunchKernelGGL(vox_clear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, keys, vals, total);
    }
    // the stages of up to VOX_BATCH copies per launch (copy = blockIdx.y; the rotations travel in the kernel arguments)
    for (int b0 = 0; b0 < nb; b0 += VOX_BATCH) {
        const int nbc = nb - b0 < VOX_BATCH ? nb - b0 : VOX_BATCH;
        VoxBatch a;
        a.pts = pts; a.n = n; a.voxel = voxel;
        for (int b = 0; b < nbc; ++b) for (int i = 0; i < 9; ++i) a.R[b][i] = R_host[9 * (size_t)(b0 + b) + i];
        a.keys = keys + (size_t)b0 * cap; a.vals = vals + (size_t)b0 * cap; a.cap = cap;
        a.bsum = bsum + (size_t)b0 * (nblk + 1); a.nblk = nblk;
        a.dcount = dcount + 2 * (size_t)b0;
        a.coords = coords + (size_t)b0 * n * 3; a.sel = sel + (size_t)b0 * n; a.pts_sel = pts_sel ? pts_sel + (size_t)b0 * n * 3 : nullptr;
        a.rk = nullptr; a.bm = nullptr; a.rank = nullptr; a.first = nullptr; a.b0 = 0;
        hipLaunchKernelGGL(vox_insert_kernel, dim3((n + 255) / 256, nbc), dim3(256), 0, s, a);
        hipLaunchKernelGGL(vox_count_kernel, dim3(nblk, nbc), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(vox_scan_kernel, dim3(nbc), dim3(1024), 0, s, a);
        hipLaunchKernelGGL(vox_scatter_kernel, dim3(nblk, nbc), dim3(1024), 0, s, a);
    }
    HIPCHK(hipGetLastError());
    phase_mark(ctx, -1, s);
    int hc[128];
    HIPCHK(hipMemcpyAsync(hc, dcount, 2 * sizeof(int) * nb, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (int b = 0; b < nb; ++b) {
        if (hc[2 * b + 1]) {
            set_error("voxelisation: a point's voxel index is outside +-%d in rotated copy %d (cloud extent / voxel size too large, or a non-finite point)", VOX_LIM, b);
            return YOHO_EINVAL;
        }
        counts_host[b] = hc[2 * b];
    }
    return 0;
}

}  // namespace yoho
