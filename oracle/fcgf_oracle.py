"""CPU restatement of the FCGF backbone forward pass - TEST INFRASTRUCTURE ONLY.

Only tests/ and __graft_entry__.smoke() may import this file; nothing under yoho_amd/ does.

The backbone (reference fcgf_model/resunet.py:10-190, residual_block.py:9-52, simple_yoho/fcgf_feat.py:33-54) is written
against MinkowskiEngine 0.5.x, which is vendored in the reference tree as *source* but is not importable in this image
(CUDA / BLAS build, no cblas.h), so its sparse-tensor semantics are restated here from its sources:

  * sparse_quantize / SparseTensor construction: rows = first occurrence of every voxel in input order
    (CPU coordinate map insert_and_map, MinkowskiEngine/utils/quantization.py:263-300);
  * strided coordinate map: floor(c / s) * s, unique, tensor stride multiplied (src/coordinate_map.hpp:58-76);
  * kernel offsets of a hyper-cube region: kernel index k -> x fastest, offset_i = (k_i - K/2) * tensor_stride for odd K
    (src/kernel_region.hpp:196-216); kernel weights are (K^3, Cin, Cout), out[o] += in[i] @ W[k] for every
    (i, o, k) with coord(i) = coord(o) + offset(k) on the INPUT tensor stride (src/coordinate_map_manager.cpp:718-755);
  * transposed convolution: output coordinates = the existing coordinate map of the finer tensor stride
    (src/coordinate_map_manager.cpp:436-462), kernel map = the finer->coarser map of the same kernel with in/out
    swapped (:756-800), i.e. out[f] += in[c] @ W[k] whenever coord(c) = coord(f) + offset(k) on the finer stride;
  * MinkowskiBatchNorm = BatchNorm1d over the feature rows (eval: running statistics, eps 1e-5); ME.cat = channel concat.

Pinning (tests/test_fcgf_oracle.py):
  * coordinate maps and kernel maps - which duplicate survives quantisation and in which order, the strided maps, and the kernel
    map of EVERY convolution of ResUNetBN2C including the strided and the TRANSPOSED ones - against the REFERENCE ITSELF:
    oracle/build_me_ref.py compiles MinkowskiEngine's own src/coordinate_map_manager.cpp (CPU_ONLY, no BLAS needed) where it lies,
    oracle/me_ref/me_maps_driver.cpp drives it exactly as src/convolution_cpu.cpp / convolution_transpose_cpu.cpp do, and
    oracle/gen_golden_me.py stores the maps in tests/golden/me_maps.npz (round 3).  Equal as sets of (kernel index, input
    coordinate, output coordinate) triples on every map; the reference numbers the rows of a strided map in hash-table order, the
    oracle in first-occurrence order - internal, no result depends on it (level 0 is equal row by row);
  * the same semantics on the known-answer vectors of MinkowskiEngine's own tests (region order and map direction:
    tests/cpp/kernel_region_cpu_test.py:22-42, 100-116; strided maps incl. negative coordinates and batches:
    tests/cpp/coordinate_map_cpu_test.py:12-15,47-65,80-86,95-125, tests/python/coordinate_manager.py:33-58,183-200,255-258;
    quantisation collisions: tests/python/quantization.py:104-113, tests/python/sparse_tensor.py:91-98).  The pair count asserted in
    tests/python/kernel_map.py:78,109 (16) is stale: the reference's own manager yields 26 on that figure, as this oracle does;
  * the convolution ARITHMETIC given the maps (out[o] += in[i] W[k], kernel indices in order) against torch.nn.functional.conv3d /
    conv_transpose3d on densified inputs - MinkowskiEngine's own CPU convolution needs a BLAS with cblas.h
    (src/math_functions_cpu.cpp), which this image lacks, so that translation unit is not built (no stand-in is written for it).
What remains unpinned is the END-TO-END network output against a running MinkowskiEngine (summation order inside its sgemm calls,
its BatchNorm wrapper) and against the pretrained backbone checkpoint, which is absent from the tree.
"""
import numpy as np

EPS = 1e-5


def sparse_quantize(coords):
    """coords (N,3) int -> indices of the first occurrence of every distinct row, ascending (simple_yoho/fcgf_feat.py:36-39)."""
    seen = {}
    for i, c in enumerate(map(tuple, np.asarray(coords).tolist())):
        if c not in seen:
            seen[c] = i
    return np.fromiter(seen.values(), dtype=np.int64, count=len(seen))


def voxelize(pc, voxel_size):
    """simple_yoho/fcgf_feat.py:33-43: floor(pc / voxel) (through int32), first point per voxel -> (sel, integer coords)."""
    coords = np.floor(np.asarray(pc, dtype=np.float64) / voxel_size).astype(np.int32)
    sel = sparse_quantize(coords)
    xyz = np.asarray(pc)[sel]
    return sel, np.floor(xyz / voxel_size).astype(np.int32)


def stride_coords(coords, ts_out):
    """coordinate map of tensor stride ts_out from a finer one: floor(c / ts) * ts, unique in first-occurrence order."""
    q = (np.floor(coords.astype(np.float32) / ts_out) * ts_out).astype(np.int32)
    keep = sparse_quantize(q)
    return q[keep]


def kernel_offsets(ksize, ts):
    """(K^3, 3) integer offsets, kernel index with x fastest (src/kernel_region.hpp:196-216); odd kernel sizes only."""
    assert ksize % 2 == 1
    k = np.arange(ksize ** 3)
    out = np.stack([k % ksize, (k // ksize) % ksize, k // (ksize * ksize)], 1) - ksize // 2
    return out.astype(np.int32) * ts


def kernel_map(coords_in, coords_out, ksize, ts_region, transpose=False):
    """(K^3, Nout) input row of every (kernel index, output row), -1 where the region cell is empty
    (src/coordinate_map_cpu.hpp:572-660 with kernel_region.hpp:196-216).  ts_region: tensor stride the kernel offsets live
    on (input stride for a convolution, the finer = OUTPUT stride for a transposed one).
    transpose=False: in = out + off(k); transpose=True: in + off(k) = out."""
    index = {tuple(c): i for i, c in enumerate(np.asarray(coords_in).tolist())}
    offs = kernel_offsets(ksize, ts_region)
    coords_out = np.asarray(coords_out)
    rows = np.full((len(offs), len(coords_out)), -1, dtype=np.int64)
    for k, off in enumerate(offs):
        src = coords_out - off if transpose else coords_out + off
        rows[k] = np.fromiter((index.get(tuple(c), -1) for c in src.tolist()), dtype=np.int64, count=len(src))
    return rows


def conv(feat_in, coords_in, coords_out, W, ksize, ts_region, transpose=False):
    """feat_in (Nin,Cin), W (K^3,Cin,Cout) or (Cin,Cout) -> (Nout,Cout) float32, accumulated in kernel-index order
    (out[o] += in[i] @ W[k] over the pairs of kernel_map)."""
    W = np.asarray(W, dtype=np.float32)
    if W.ndim == 2:
        assert len(coords_in) == len(coords_out)
        return feat_in.astype(np.float32) @ W
    rows_all = kernel_map(coords_in, coords_out, ksize, ts_region, transpose)
    out = np.zeros((len(coords_out), W.shape[2]), dtype=np.float32)
    for k, rows in enumerate(rows_all):
        m = rows >= 0
        if m.any():
            out[m] += feat_in[rows[m]].astype(np.float32) @ W[k]
    return out


def bn(x, p):
    s = p["weight"] / np.sqrt(p["running_var"] + np.float32(EPS))
    return (x * s + (p["bias"] - p["running_mean"] * s)).astype(np.float32)


def relu(x):
    return np.maximum(x, 0).astype(np.float32)


def _bn_params(sd, name):
    return {k: sd[f"{name}.bn.{k}"] for k in ("weight", "bias", "running_mean", "running_var")}


def block(x, coords, ts, sd, name):
    """BasicBlockBN (fcgf_model/residual_block.py:37-52)."""
    out = relu(bn(conv(x, coords, coords, sd[f"{name}.conv1.kernel"], 3, ts), _bn_params(sd, f"{name}.norm1")))
    out = bn(conv(out, coords, coords, sd[f"{name}.conv2.kernel"], 3, ts), _bn_params(sd, f"{name}.norm2"))
    return relu(out + x)


def resunet_forward(coords, sd, conv1_kernel_size=7, normalize_feature=True):
    """ResUNet2.forward (fcgf_model/resunet.py:141-190) on one cloud: coords (N,3) int32 distinct, input feature = ones
    (simple_yoho/fcgf_feat.py:41).  Returns (N,Cout) float32, rows in input order."""
    c1 = np.asarray(coords, dtype=np.int32)
    x = np.ones((len(c1), 1), dtype=np.float32)
    c2 = stride_coords(c1, 2)
    c4 = stride_coords(c2, 4)
    c8 = stride_coords(c4, 8)

    out_s1 = block(bn(conv(x, c1, c1, sd["conv1.kernel"], conv1_kernel_size, 1), _bn_params(sd, "norm1")), c1, 1, sd, "block1")
    out = relu(out_s1)
    out_s2 = block(bn(conv(out, c1, c2, sd["conv2.kernel"], 3, 1), _bn_params(sd, "norm2")), c2, 2, sd, "block2")
    out = relu(out_s2)
    out_s4 = block(bn(conv(out, c2, c4, sd["conv3.kernel"], 3, 2), _bn_params(sd, "norm3")), c4, 4, sd, "block3")
    out = relu(out_s4)
    out_s8 = block(bn(conv(out, c4, c8, sd["conv4.kernel"], 3, 4), _bn_params(sd, "norm4")), c8, 8, sd, "block4")
    out = relu(out_s8)

    out = block(bn(conv(out, c8, c4, sd["conv4_tr.kernel"], 3, 4, transpose=True), _bn_params(sd, "norm4_tr")), c4, 4, sd, "block4_tr")
    out = np.concatenate([relu(out), out_s4], 1)
    out = block(bn(conv(out, c4, c2, sd["conv3_tr.kernel"], 3, 2, transpose=True), _bn_params(sd, "norm3_tr")), c2, 2, sd, "block3_tr")
    out = np.concatenate([relu(out), out_s2], 1)
    out = block(bn(conv(out, c2, c1, sd["conv2_tr.kernel"], 3, 1, transpose=True), _bn_params(sd, "norm2_tr")), c1, 1, sd, "block2_tr")
    out = np.concatenate([relu(out), out_s1], 1)
    out = relu(conv(out, c1, c1, sd["conv1_tr.kernel"], 1, 1))
    out = conv(out, c1, c1, sd["final.kernel"], 1, 1) + sd["final.bias"].astype(np.float32)
    if normalize_feature:
        out = out / np.linalg.norm(out, axis=1, keepdims=True)
    return out.astype(np.float32)


def extract_features(pc, voxel_size, sd, conv1_kernel_size=7, normalize_feature=True):
    """fcgf_extractor.extract_features (simple_yoho/fcgf_feat.py:33-49): -> (sel, F) with F = L2-normalised rows."""
    sel, coords = voxelize(pc, voxel_size)
    F = resunet_forward(coords, sd, conv1_kernel_size, normalize_feature)
    F = F / np.linalg.norm(F, axis=1, keepdims=True)
    return sel, F.astype(np.float32)
