"""CPU oracle for the YOHO hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference's algorithm for the 60-rotation
descriptor path and the YOHO-O / YOHO-C estimators.  It is the *checker*: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``yoho_amd/`` imports it, and the product path raises if the
HIP library is missing - it never falls back to this code.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference from
``/root/reference`` (pure Python on this path), runs every stage on seeded inputs and
stores inputs + the reference's outputs under ``tests/golden/``; ``tests/test_oracle.py``
checks every function below against those vectors.  Upstream itself has no tests or golden
vectors for this path, so the pin is "outputs of the reference run here".

Every function cites the reference file:line it restates (paths relative to the
reference root).
"""
import numpy as np

G, NTAP = 60, 13
BN_EPS = np.float32(1e-5)          # torch.nn.BatchNorm2d default eps


# ----------------------------------------------------------------------------------------
# group conv building blocks
# ----------------------------------------------------------------------------------------
def data_process(x, N):
    """utils/network.py:80-84 (PartI_network.data_process) / :46-52 / :243-249.
    x (B,C,60) -> (B,C,60,13): neighbour gather over the flat 780-index."""
    B, C, _ = x.shape
    return x[:, :, N.reshape(-1)].reshape(B, C, G, NTAP)


def bn_eval(x, sd, prefix):
    """nn.BatchNorm2d in eval mode on a (B,C,...) tensor (utils/network.py:16,28,33)."""
    g = sd[prefix + ".weight"]; b = sd[prefix + ".bias"]
    m = sd[prefix + ".running_mean"]; v = sd[prefix + ".running_var"]
    shape = (1, -1) + (1,) * (x.ndim - 2)
    inv = (np.float32(1.0) / np.sqrt(v + BN_EPS)).astype(np.float32)
    return ((x - m.reshape(shape)) * inv.reshape(shape) * g.reshape(shape) + b.reshape(shape)).astype(np.float32)


def relu(x):
    return np.maximum(x, np.float32(0))


def conv_1xk(xg, w, b):
    """nn.Conv2d(Cin,Cout,(1,K)) applied to (B,Cin,60,K) -> (B,Cout,60)
    (utils/network.py:18,30,35,76): out[b,o,g] = bias[o] + sum_{c,k} w[o,c,0,k] xg[b,c,g,k]."""
    B, C, Gn, K = xg.shape
    O = w.shape[0]
    a = np.ascontiguousarray(xg.transpose(0, 2, 1, 3)).reshape(B * Gn, C * K)       # rows (b,g), cols (c,k)
    wt = np.ascontiguousarray(w.reshape(O, C * K).T)
    out = a @ wt + b[None, :]
    return np.ascontiguousarray(out.reshape(B, Gn, O).transpose(0, 2, 1)).astype(np.float32)


def comb_conv(x, N, sd, prefix):
    """Comb_Conv (utils/network.py:12-21): gather -> BN -> ReLU -> Conv(1,13); prefix.0 = BN, prefix.2 = conv."""
    t = relu(bn_eval(data_process(x, N), sd, prefix + ".0"))
    return conv_1xk(t, sd[prefix + ".2.weight"], sd[prefix + ".2.bias"])


def residual_comb_conv(x, N, sd, prefix):
    """Residual_Comb_Conv.forward (utils/network.py:54-65) with in_dim == out_dim (identity shortcut)."""
    t = comb_conv(x, N, sd, prefix + ".comb_layer_in")
    t = comb_conv(t, N, sd, prefix + ".comb_layer_out")
    return t + x


# ----------------------------------------------------------------------------------------
# a5 / a6: PartI
# ----------------------------------------------------------------------------------------
def partI_forward(x, sd, N):
    """PartI_network.forward (utils/network.py:86-105) through PartI_test (:140-147).
    x (B,32,60) f32 -> eqv (B,32,60), inv (B,32)."""
    x = x.astype(np.float32)
    p = "PartI_net."
    h = conv_1xk(data_process(x, N), sd[p + "Conv_in.0.weight"], sd[p + "Conv_in.0.bias"])   # :88 (no BN/ReLU)
    h = residual_comb_conv(h, N, sd, p + "SO3_Conv_layers.0")                                    # :89-90
    y = comb_conv(h, N, sd, p + "Conv_out.comb_layer")                                           # :91-92
    eqv = y + x                                                                                  # :98
    inv = np.mean(eqv, axis=-1, dtype=np.float32)                                                # :99
    n_e = np.maximum(np.sqrt(np.sum(eqv * eqv, axis=1, keepdims=True)), np.float32(1e-4))        # :102
    n_i = np.maximum(np.sqrt(np.sum(inv * inv, axis=1, keepdims=True)), np.float32(1e-4))        # :103
    return (eqv / n_e).astype(np.float32), (inv / n_i).astype(np.float32)


def partI_extract(x, sd, N, batch=900):
    """extractor_PartI.Extract inner loop (tests/extractor.py:49-59): chunks of test_batch_size, keeps eqv only."""
    out = []
    for s in range(0, x.shape[0], batch):
        out.append(partI_forward(x[s:s + batch], sd, N)[0])
    return np.concatenate(out, axis=0)


def partI_forward_torch(x, sd, N):
    """Same op sequence as the reference on torch CPU kernels (index-gather -> batch_norm -> relu ->
    F.conv2d); used as the timed CPU baseline because that is what the reference's CPU path runs."""
    import torch
    import torch.nn.functional as Fn
    t = lambda k: torch.from_numpy(sd[k])
    Nf = torch.from_numpy(N.reshape(-1))
    def gather(v):
        return v[:, :, Nf].reshape(v.shape[0], v.shape[1], G, NTAP)
    def bnrelu(v, pre):
        return Fn.relu(Fn.batch_norm(v, t(pre + ".running_mean"), t(pre + ".running_var"),
                                     t(pre + ".weight"), t(pre + ".bias"), False, 0.1, 1e-5))
    def conv(v, pre):
        return Fn.conv2d(v, t(pre + ".weight"), t(pre + ".bias"))[:, :, :, 0]
    p = "PartI_net."
    with torch.no_grad():
        xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        h0 = conv(gather(xt), p + "Conv_in.0")
        r = p + "SO3_Conv_layers.0."
        m = conv(bnrelu(gather(h0), r + "comb_layer_in.0"), r + "comb_layer_in.2")
        h2 = conv(bnrelu(gather(m), r + "comb_layer_out.0"), r + "comb_layer_out.2") + h0
        y = conv(bnrelu(gather(h2), p + "Conv_out.comb_layer.0"), p + "Conv_out.comb_layer.2")
        eqv = y + xt
        inv = torch.mean(eqv, dim=-1)
        eqv = eqv / torch.clamp_min(torch.norm(eqv, dim=1, keepdim=True), min=1e-4)
        inv = inv / torch.clamp_min(torch.norm(inv, dim=1, keepdim=True), min=1e-4)
    return eqv.numpy(), inv.numpy()


# ----------------------------------------------------------------------------------------
# a7: matcher
# ----------------------------------------------------------------------------------------
def group_mean_np(eqv):
    """tests/matcher.py:35-36: np.mean(feats, axis=-1) in fp32 (numpy pairwise sum, NOT renormalised)."""
    return np.mean(eqv.astype(np.float32), axis=-1).astype(np.float32)


def group_mean_np_explicit(eqv):
    """The same value written out: numpy's pairwise kernel for n=60 (< 128) keeps 8 running sums over
    x[8b+j], combines them as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), adds the 4 leftovers in order,
    then divides by 60 in fp32.  This is the recipe the HIP kernel implements."""
    x = eqv.astype(np.float32)
    r = [x[..., j].copy() for j in range(8)]
    for b in range(1, 7):
        for j in range(8):
            r[j] = r[j] + x[..., 8 * b + j]
    s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    for j in range(56, 60):
        s = s + x[..., j]
    return (s / np.float32(60.0)).astype(np.float32)


def pdist_l2(A, B, squared=False):
    """modified_knn_matcher.pdist, dist_type='L2' (utils/knn_search.py:17-20):
    sqrt(sum((a-b)^2) + 1e-7) in fp32.  The 32-term sum is written in the order torch's CPU
    reduction uses for a contiguous 32-float row (8 lane accumulators over x[l], x[8+l], x[16+l],
    x[24+l]; lanes then added 0..7 in sequence) - verified bit-identical to torch.sum in
    oracle/gen_golden.py - so that argmin ties resolve exactly like the reference run.
    The sqrt here is correctly rounded; the reference's torch-CPU build (MKL VML) returns a
    value 1 ulp low for ~0.6 % of inputs, CUDA's sqrtf is correctly rounded - the golden test
    therefore checks D2 bit-exactly and the distance to 1 ulp."""
    A = A.astype(np.float32); B = B.astype(np.float32)
    d = A[:, None, :] - B[None, :, :]
    sq = d * d
    F = sq.shape[2]
    if F % 8 == 0 and F >= 8:
        lanes = sq[:, :, 0:8].copy()
        for b in range(1, F // 8):
            lanes = lanes + sq[:, :, 8 * b:8 * b + 8]
        D2 = lanes[:, :, 0].copy()
        for l in range(1, 8):
            D2 = D2 + lanes[:, :, l]
    else:                                   # 3-D points (gather path): plain left-to-right
        D2 = sq[:, :, 0].copy()
        for f in range(1, F):
            D2 = D2 + sq[:, :, f]
    if squared:
        return D2
    return np.sqrt(D2 + np.float32(1e-7)).astype(np.float32)


def find_nn(src, tgt, chunk=500):
    """modified_knn_matcher.find_nn_gpu / __call__ with k=1 (utils/knn_search.py:26-66,138-154):
    for every src row the index of the nearest tgt row; dist.min(dim=1) returns the first minimum."""
    idx = np.empty(src.shape[0], dtype=np.int64)
    for s in range(0, src.shape[0], chunk):
        idx[s:s + chunk] = np.argmin(pdist_l2(src[s:s + chunk], tgt), axis=1)
    return idx


def mutual_match(inv0, inv1):
    """matcher_dual.match (tests/matcher.py:37-48): KNN(feats1,feats0) = NN of each row of 0 in 1,
    KNN(feats0,feats1) = NN of each row of 1 in 0, keep i with back[fwd[i]] == i, ascending i.
    Returns (M,2) int64."""
    fwd = find_nn(inv0, inv1)
    back = find_nn(inv1, inv0)
    i0 = np.arange(inv0.shape[0], dtype=np.int64)
    keep = back[fwd] == i0
    return np.stack([i0[keep], fwd[keep]], axis=1).astype(np.int64)


# ----------------------------------------------------------------------------------------
# a8: coarse rotation index
# ----------------------------------------------------------------------------------------
def des2r_cor(d1, d2, P):
    """extractor_dr_index.Batch_Des2R_torch (tests/extractor.py:74-78):
    cor[b,a] = sum_f sum_g d1[b,f,P[a,g]] * d2[b,f,g]  (fp32)."""
    B, Fd, _ = d1.shape
    d1p = d1[:, :, P.reshape(-1)].reshape(B, Fd, G, G)           # [b,f,a,g]
    return np.einsum("bfag,bfg->ba", d1p.astype(np.float32), d2.astype(np.float32), optimize=False).astype(np.float32)


def des2r(d1, d2, P):
    return np.argmax(des2r_cor(d1, d2, P), axis=1).astype(np.int64)


# ----------------------------------------------------------------------------------------
# a9 / a10: PartII
# ----------------------------------------------------------------------------------------
def partII_forward(bf0, bf1, af0, af1, pre_idx, sd, N, P):
    """PartII_test.forward (utils/network.py:259-278).  Inputs are NOT modified (the reference
    permutes bf0/af0 in place, :266-268)."""
    B = bf0.shape[0]
    perm = P[pre_idx.astype(np.int64)]                                 # (B,60)
    bi = np.arange(B)[:, None, None]; fi = np.arange(bf0.shape[1])[None, :, None]
    bf0p = bf0[bi, fi, perm[:, None, :]]
    af0p = af0[bi, fi, perm[:, None, :]]
    x = np.concatenate([bf0p, bf1, af0p, af1], axis=1).astype(np.float32)      # :269 (B,128,60)
    h = comb_conv(x, N, sd, "Conv_init.comb_layer")                            # :252-253
    h = residual_comb_conv(h, N, sd, "PartII_SO3_Conv_layers.0")               # :254-255
    # :273-276: 1x1 MLP on all 60 group elements, keep [:, :, 0, 0]
    def fc(v, pre):                                                             # v (B,C,60)
        w = sd[pre + ".weight"][:, :, 0, 0]
        return np.einsum("oc,bcg->bog", w, v).astype(np.float32) + sd[pre + ".bias"][None, :, None]
    t = fc(h, "PartII_To_R_FC.0")
    t = relu(bn_eval(t, sd, "PartII_To_R_FC.1"))
    t = fc(t, "PartII_To_R_FC.3")
    t = relu(bn_eval(t, sd, "PartII_To_R_FC.4"))
    t = fc(t, "PartII_To_R_FC.6")
    q = t[:, :, 0]
    return (q / np.sqrt(np.sum(q * q, axis=1))[:, None]).astype(np.float32)    # :277 (no clamp)


def des2r_torch(d1, d2, P):
    """Batch_Des2R_torch on torch's CPU kernels, the op sequence of tests/extractor.py:74-78 (index gather, einsum,
    argmax): the timed CPU baseline of this stage (the numpy einsum of des2r_cor is several times slower)."""
    import torch
    with torch.no_grad():
        a, b = torch.from_numpy(np.ascontiguousarray(d1, dtype=np.float32)), torch.from_numpy(np.ascontiguousarray(d2, dtype=np.float32))
        B, Fd, _ = a.shape
        idx = torch.from_numpy(np.ascontiguousarray(P.reshape(-1), dtype=np.int64))
        cor = torch.einsum('bfag,bfg->ba', a[:, :, idx].reshape([B, Fd, G, G]), b)
        return torch.argmax(cor, dim=1).numpy().astype(np.int64)


def partII_forward_torch(bf0, bf1, af0, af1, pre_idx, sd, N, P):
    """PartII_test.forward on torch's CPU kernels, the reference's op sequence (utils/network.py:259-278 with :243-256 and
    :12-65): per-match permutation, cat, [index-gather -> batch_norm -> relu -> conv2d (1,13)] x 3 with the residual, then the
    1x1 conv / BN / ReLU head on all 60 group elements, [:, :, 0, 0], normalise.  Timed CPU baseline of the stage."""
    import torch
    import torch.nn.functional as Fn
    t = lambda k: torch.from_numpy(sd[k])
    Nf = torch.from_numpy(np.ascontiguousarray(N.reshape(-1), dtype=np.int64))
    Pt = torch.from_numpy(np.ascontiguousarray(P, dtype=np.int64))

    def gather(v):
        return v[:, :, Nf].reshape(v.shape[0], v.shape[1], G, NTAP)

    def bn(v, pre):
        return Fn.batch_norm(v, t(pre + ".running_mean"), t(pre + ".running_var"), t(pre + ".weight"), t(pre + ".bias"), False, 0.1, 1e-5)

    def comb(v, pre):                                   # Comb_Conv: BN -> ReLU -> Conv2d on the gathered tensor
        return Fn.conv2d(Fn.relu(bn(gather(v), pre + ".0")), t(pre + ".2.weight"), t(pre + ".2.bias"))[:, :, :, 0]
    with torch.no_grad():
        b0, b1, a0, a1 = (torch.from_numpy(np.array(x, dtype=np.float32)) for x in (bf0, bf1, af0, af1))
        idx = torch.from_numpy(np.ascontiguousarray(pre_idx, dtype=np.int64))
        for i in range(b0.shape[0]):                    # :266-268 (on copies: the callers' arrays stay untouched)
            b0[i] = b0[i, :, Pt[idx[i]]]
            a0[i] = a0[i, :, Pt[idx[i]]]
        x = torch.cat([b0, b1, a0, a1], dim=1)
        h = comb(x, "Conv_init.comb_layer")
        r = "PartII_SO3_Conv_layers.0"
        h = comb(comb(h, r + ".comb_layer_in"), r + ".comb_layer_out") + h
        v = h.unsqueeze(-1)
        f = "PartII_To_R_FC."
        v = Fn.relu(bn(Fn.conv2d(v, t(f + "0.weight"), t(f + "0.bias")), f + "1"))
        v = Fn.relu(bn(Fn.conv2d(v, t(f + "3.weight"), t(f + "3.bias")), f + "4"))
        q = Fn.conv2d(v, t(f + "6.weight"), t(f + "6.bias"))[:, :, 0, 0]
        q = q / torch.norm(q, dim=1)[:, None]
    return q.numpy()


def batch_create(feats0_fcgf, feats1_fcgf, feats0_yoho, feats1_yoho, index_pre):
    """extractor_PartII.batch_create (tests/extractor.py:125-138): note the 0<->1 exchange."""
    return dict(before_eqv0=feats1_fcgf, before_eqv1=feats0_fcgf,
                after_eqv0=feats1_yoho, after_eqv1=feats0_yoho, pre_idx=index_pre)


def matrix_from_quaternion(q):
    """utils/r_eval.py:94-110.  q is a float32 vector, so every entry is evaluated in fp32
    (numpy scalar arithmetic) in exactly this operation order, then stored into an f64 matrix."""
    w, x, y, z = q[0], q[1], q[2], q[3]
    mat = np.eye(3)
    mat[0, 0] = 1 - 2 * y * y - 2 * z * z
    mat[0, 1] = 2 * x * y - 2 * z * w
    mat[0, 2] = 2 * x * z + 2 * y * w
    mat[1, 0] = 2 * x * y + 2 * z * w
    mat[1, 1] = 1 - 2 * x * x - 2 * z * z
    mat[1, 2] = 2 * y * z - 2 * x * w
    mat[2, 0] = 2 * x * z - 2 * y * w
    mat[2, 1] = 2 * y * z + 2 * x * w
    mat[2, 2] = 1 - 2 * x * x - 2 * y * y
    return mat


def hyp_from_quat(quat, idx, keys0, keys1, R32):
    """tests/extractor.py:187-199: R = quat2mat(q) @ Rgroup_f32[idx] (f64 result), t = k0 - k1 @ R.T."""
    M = quat.shape[0]
    T = np.empty((M, 3, 4), dtype=np.float64)
    for i in range(M):
        R = matrix_from_quaternion(quat[i].astype(np.float32)) @ R32[int(idx[i])]
        T[i, :, :3] = R
        T[i, :, 3] = keys0[i] - keys1[i] @ R.T
    return T


# ----------------------------------------------------------------------------------------
# a11: YOHO-O, a12: YOHO-C
# ----------------------------------------------------------------------------------------
def transform_points(pts, T):
    """utils/utils.py:42-50 ((3,4) branch)."""
    return pts @ T[:, :3].T + T[:, 3:].T


def inlier_count(k0, k1, T, d):
    """overlap_cal (tests/estimator.py:286-290 / :66-70) as an integer count; overlap = count / M."""
    diff = np.sum(np.square(k0 - transform_points(k1, T)), axis=-1)
    return int(np.sum(diff < d * d))


def yohoo_select(k0, k1, Trans, order, d, max_iter=1000):
    """yohoo.ransac inner part (tests/estimator.py:321-336) with the np.random.shuffle result passed
    in as `order`.  Returns (best_t_id, best_count, trans(3,4) or eye(4) if nothing beats 0)."""
    Tr = Trans[order[0:max_iter]]
    best, best_id, best_T = 0, 0, np.eye(4)
    for t_id in range(Tr.shape[0]):
        c = inlier_count(k0, k1, Tr[t_id], d)
        if c > best:                      # strict: first maximum wins (overlap = c / M, same M)
            best, best_id, best_T = c, t_id, Tr[t_id]
    return best_id, best, best_T


def dr_statistic(dr_idx):
    """yohoc.DR_statictic (tests/estimator.py:34-51): buckets + p_i ~ n(n-.01)(n-.02), n=count/100."""
    buckets = {i: [] for i in range(G)}
    for t in range(dr_idx.shape[0]):
        buckets[int(dr_idx[t])].append(t)
    prob = []
    for i in range(G):
        if len(buckets[i]) < 2:
            prob.append(0)
        else:
            num = float(len(buckets[i])) / 100.0
            prob.append(num * (num - 0.01) * (num - 0.02))
    prob = np.array(prob)
    if np.sum(prob) < 1e-4:
        return None, None
    return buckets, prob / np.sum(prob)


def threepps2tran(kps0, kps1, proper=False):
    """yohoc.Threepps2Tran (tests/estimator.py:55-63).  R = Vt.T @ U.T with NO determinant fix, so
    for 3 centred points (rank-2 covariance) the sign of the null direction is LAPACK-defined.
    proper=True flips that direction when det < 0 (what the HIP kernel computes by default);
    returns (T(3,4), det_sign_of_reference_R)."""
    c0 = np.mean(kps0, 0, keepdims=True)
    c1 = np.mean(kps1, 0, keepdims=True)
    m = (kps1 - c1).T @ (kps0 - c0)
    U, S, VT = np.linalg.svd(m)
    R = VT.T @ U.T
    det = np.linalg.det(R)
    if proper and det < 0:
        VT = VT.copy(); VT[2] = -VT[2]
        R = VT.T @ U.T
    off = c0 - (c1 @ R.T)
    return np.concatenate([R, off.T], 1), (1 if det >= 0 else -1)


def yohoc_select(k0, k1, triples, d, proper=False):
    """yohoc.ransac loop body (tests/estimator.py:119-137) for an explicit (I,3) sample sequence.
    Returns (best_iter (1-based, 0 = none), best_count, trans, det_signs (I,))."""
    best, best_it, best_T = 0, 0, np.eye(4)
    dets = np.zeros(len(triples), dtype=np.int8)
    for it, tri in enumerate(triples):
        T, s = threepps2tran(k0[tri], k1[tri], proper=proper)
        dets[it] = s
        c = inlier_count(k0, k1, T, d)
        if c > best:
            best, best_it, best_T = c, it + 1, T
    return best_it, best, best_T, dets


def yohoc_draw_triples(dr_idx, max_iter, rng):
    """The sampling half of yohoc.ransac (tests/estimator.py:119-128) with an explicit RandomState:
    weighted draw of a coarse index, then 3 matches (with replacement) from that bucket."""
    buckets, prob = dr_statistic(dr_idx)
    if prob is None:
        return None
    tri, it, execs = [], 0, 0
    while it < max_iter:
        if execs > 50000:
            break
        execs += 1
        r = rng.choice(range(G), p=prob)
        if len(buckets[r]) < 2:
            continue
        it += 1
        tri.append(rng.choice(np.array(buckets[r]), 3))
    return np.array(tri, dtype=np.int64).reshape(-1, 3)


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter = 4 and key = 2 words
    of 32 bits, plain Python integers.  The generator of the build's device-side YOHO-C sampling (csrc/estim.hip)."""
    c0, c1, c2, c3 = (int(v) & 0xFFFFFFFF for v in counter)
    k0, k1 = (int(v) & 0xFFFFFFFF for v in key)
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def yohoc_device_triples(dr_idx, max_iter, seed):
    """The sampling half of yohoc.ransac (tests/estimator.py:34-51,119-128) as yoho_c_ransac_device draws it: same
    statistic (buckets in ascending match order, weights n(n-.01)(n-.02) for buckets of >= 2 matches), same kind of draw
    (a cdf search for the coarse index, then three members of its bucket with replacement), but from a counter-based
    Philox stream keyed by `seed` instead of numpy's global MT19937 stream:
        iteration it: (w0, w1, .., ..) = philox((it,0,0,0), seed) -> u = ((w0 << 21) | (w1 >> 11)) / 2^53,
                      bucket = first b with cumsum(p)[b] > u * sum(p)   (p unnormalised, summed b = 0..59 in f64);
                      (v0, v1, v2, ..) = philox((it,1,0,0), seed) -> member j = bucket[(v_j * len(bucket)) >> 32].
    Returns the (max_iter,3) int64 triples, or None when no bucket has two matches (reference: recalltime 50001)."""
    dr = np.clip(np.asarray(dr_idx, dtype=np.int64), 0, G - 1)
    buckets = [np.nonzero(dr == b)[0] for b in range(G)]
    run, cdf = 0.0, []
    for b in range(G):
        p = 0.0
        if len(buckets[b]) >= 2:
            num = float(len(buckets[b])) / 100.0
            p = (num * (num - 0.01)) * (num - 0.02)
        run = run + p
        cdf.append(run)
    if run < 1e-4:
        return None
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    tri = np.zeros((max_iter, 3), dtype=np.int64)
    for it in range(max_iter):
        w = philox4x32_10((it, 0, 0, 0), key)
        v = philox4x32_10((it, 1, 0, 0), key)
        u = float((w[0] << 21) | (w[1] >> 11)) * 2.0 ** -53
        thr = u * cdf[-1]
        b = next((j for j in range(G) if cdf[j] > thr), G - 1)
        n = len(buckets[b])
        for p_ in range(3):
            tri[it, p_] = buckets[b][(v[p_] * n) >> 32]
    return tri


# ----------------------------------------------------------------------------------------
# a2: 60-fold FCGF feature gather
# ----------------------------------------------------------------------------------------
def group_gather_one(keys, pts, feat, Rg):
    """YOHO_testset.py:153-160 for one group element: rotate the keypoints (f64), 1-NN of every
    rotated key among the down-sampled rotated cloud (brute force, KNN 'L2' form; torch promotes
    f64 keys vs f32 pts to f64), gather the 32-D feature rows."""
    kr = keys @ Rg.T                                           # f64
    d = kr[:, None, :] - pts[None, :, :].astype(np.float64)
    sq = d * d
    D2 = (sq[:, :, 0] + sq[:, :, 1]) + sq[:, :, 2]
    idx = np.argmin(np.sqrt(D2 + 1e-7), axis=1)
    return feat[idx, :], idx


def group_gather(keys, pts_list, feat_list, R64):
    """YOHO_testset.py:153-166: stack the 60 gathered (K,32) blocks on the last axis, ordered by g."""
    out = np.empty((keys.shape[0], feat_list[0].shape[1], G), dtype=np.float32)
    for g in range(G):
        out[:, :, g] = group_gather_one(keys, pts_list[g], feat_list[g], R64[g])[0]
    return out


# ----------------------------------------------------------------------------------------
# a13: pre.log
# ----------------------------------------------------------------------------------------
def r_pre_log_text(pc_ids, pair_ids, trans_by_pair):
    """R_pre_log (tests/estimator.py:12-24) as a string."""
    n = int(len(pc_ids))
    out = []
    for (a, b) in pair_ids:
        T = trans_by_pair[(a, b)]
        out.append(f"{int(a)}\t{int(b)}\t{n}\n")
        for r in range(3):
            out.append(f"{T[r][0]}\t{T[r][1]}\t{T[r][2]}\t{T[r][3]}\n")
        out.append(f"{0.0}\t{0.0}\t{0.0}\t{1.0}\n")
    return "".join(out)
