"""Synthetic workloads for tests, golden fixtures, smoke() and bench.py.

3DMatch data, FCGF features and the pretrained checkpoints are absent from the reference
tree (.MISSING_LARGE_BLOBS), so every workload is synthetic:
unit-norm group features (FCGF features are L2-normalised, fcgf_model/resunet.py:187-190) and
fragment pairs built by permuting the group axis with 60_60.npy[i] + noise + outlier rows, with
keypoints related by a rigid motion.

Everything is derived from ``weights.hash_uniform`` with only element-wise IEEE ops
(+,-,*,/,sqrt) and explicitly ordered reductions, so the same bits come out on any machine
and the golden fixtures can store seeds instead of megabytes of inputs.
"""
import numpy as np
from .weights import hash_uniform
from .tables import default_tables, G, F


def _gauss(seed, name, n, dtype=np.float32):
    """Approximately N(0,1): Irwin-Hall sum of 4 uniforms, exact arithmetic only."""
    u = hash_uniform(seed, name, 4 * n).astype(np.float64).reshape(4, n)
    v = (((u[0] + u[1]) + u[2]) + u[3] - 2.0) * 1.7320508075688772
    return v.astype(dtype)


def _unit_norm_rows(x):
    """L2-normalise (K,32,60) over axis 1 with a fixed (sequential) summation order."""
    s = x[:, 0, :] * x[:, 0, :]
    for c in range(1, x.shape[1]):
        s = s + x[:, c, :] * x[:, c, :]
    return (x / np.sqrt(s)[:, None, :]).astype(np.float32)


def unit_features(K, seed=0, name="feat"):
    """(K,32,60) f32, unit norm over the 32 channels for every (keypoint, group element)."""
    x = _gauss(seed, name, K * F * G).reshape(K, F, G)
    return _unit_norm_rows(x)


def quat_to_mat64(q):
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
                     [2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w],
                     [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y]], dtype=np.float64)


def _apply_rt(k, R, t):
    """k @ R.T + t with an explicit operation order (no BLAS)."""
    out = np.empty_like(k)
    for r in range(3):
        out[:, r] = ((k[:, 0] * R[r, 0] + k[:, 1] * R[r, 1]) + k[:, 2] * R[r, 2]) + t[r]
    return out


def make_pair(K, seed=0, tables=None, noise=0.02, outlier_frac=0.4, key_noise=0.01, max_res_deg=15.0):
    """Synthetic fragment pair.

    fragment 1 = random unit features / keys U[0,3]^3;  fragment 0 = fragment 1 with the group
    axis permuted by P[gi] (+noise, renormalised), a fraction of rows replaced by fresh random
    features (outliers), rows shuffled; keys0 = keys1 @ (R_res R_gi)^T + t + noise.
    Returns dict(feat0, feat1 (K,32,60) f32, keys0, keys1 (K,3) f64, gi, gt (3,4), perm).
    """
    tb = tables or default_tables()
    u = hash_uniform(seed, "pairmeta", 16).astype(np.float64)
    gi = int(u[0] * G) % G
    feat1 = unit_features(K, seed, "feat1")
    keys1 = hash_uniform(seed, "keys1", K * 3).astype(np.float64).reshape(K, 3) * 3.0

    # residual rotation: small-angle quaternion about a hashed axis (<= max_res_deg)
    ax = u[1:4] - 0.5
    ax = ax / np.sqrt((ax[0] * ax[0] + ax[1] * ax[1]) + ax[2] * ax[2])
    half = 0.5 * np.deg2rad(max_res_deg) * u[4]
    s = half - half ** 3 / 6.0                      # sin/cos by short series: exact-op only
    c = np.sqrt(1.0 - s * s)
    Rres = quat_to_mat64(np.array([c, ax[0] * s, ax[1] * s, ax[2] * s]))
    Rg = tb.R64[gi]
    R = np.empty((3, 3))
    for i in range(3):
        for j in range(3):
            R[i, j] = (Rres[i, 0] * Rg[0, j] + Rres[i, 1] * Rg[1, j]) + Rres[i, 2] * Rg[2, j]
    t = (u[5:8] - 0.5) * 2.0

    # fragment 0 rows: a hashed permutation of fragment 1's rows (argsort of uniforms: integer result)
    perm = np.argsort(hash_uniform(seed, "rowperm", K), kind="stable")
    f0 = feat1[perm][:, :, tb.P[gi]] + np.float32(noise) * _gauss(seed, "fnoise", K * F * G).reshape(K, F, G)
    fresh = unit_features(K, seed, "outliers")
    is_out = hash_uniform(seed, "isout", K) < outlier_frac
    f0[is_out] = fresh[is_out]
    feat0 = _unit_norm_rows(f0)
    keys0 = _apply_rt(keys1[perm], R, t) + key_noise * _gauss(seed, "knoise", K * 3, np.float64).reshape(K, 3)
    fresh_k = hash_uniform(seed, "outkeys", K * 3).astype(np.float64).reshape(K, 3) * 3.0
    keys0[is_out] = fresh_k[is_out]
    gt = np.concatenate([R, t[:, None]], axis=1)
    return dict(feat0=np.ascontiguousarray(feat0), feat1=np.ascontiguousarray(feat1),
                keys0=np.ascontiguousarray(keys0), keys1=np.ascontiguousarray(keys1),
                gi=gi, gt=gt, perm=perm, is_out=is_out)


def estimator_case(M=1500, H=1000, seed=0, inlier_frac=0.3, tables=None):
    """Estimator micro-benchmark inputs: M matches, M per-match hypotheses of which
    ~inlier_frac are close to the true motion."""
    tb = tables or default_tables()
    u = hash_uniform(seed, "estmeta", 16).astype(np.float64)
    gi = int(u[0] * G) % G
    R = tb.R64[gi]
    t = (u[5:8] - 0.5) * 2.0
    k1 = hash_uniform(seed, "ek1", M * 3).astype(np.float64).reshape(M, 3) * 3.0
    k0 = _apply_rt(k1, R, t) + 0.01 * _gauss(seed, "eknoise", M * 3, np.float64).reshape(M, 3)
    inl = hash_uniform(seed, "einl", M) < inlier_frac
    rnd = hash_uniform(seed, "ernd", M * 3).astype(np.float64).reshape(M, 3) * 3.0
    k0[~inl] = rnd[~inl]
    # per-match hypotheses: inliers get R (+ tiny perturbation through t), outliers random group rot
    gidx = (hash_uniform(seed, "egidx", M) * G).astype(np.int64) % G
    gidx[inl] = gi
    T = np.empty((M, 3, 4))
    T[:, :, :3] = tb.R64[gidx]
    for r in range(3):
        T[:, r, 3] = k0[:, r] - ((k1[:, 0] * T[:, r, 0] + k1[:, 1] * T[:, r, 1]) + k1[:, 2] * T[:, r, 2])
    return dict(k0=k0, k1=k1, T=T, dr=gidx, gi=gi, gt=np.concatenate([R, t[:, None]], 1), inl=inl)


def make_scene(nfrag=4, K=64, seed=0, tables=None, noise=0.02, outlier_frac=0.3, key_noise=0.01, res_deg=None):
    """A synthetic multi-fragment scene for the evaluator / Registration-Recall rows.

    Every fragment f is a copy of one base fragment moved by its own rigid motion T_f = [Rres_f R_{g_f} | t_f] with
    the group axis of its features permuted by P[g_f] (+ noise, outliers, row shuffle).  Returns
    dict(feats [nfrag](K,32,60) f32, keys [nfrag](K,3) f64, poses [nfrag](3,4), pairs [(i,j) i<j],
         gt {(i,j): (4,4) with keys_i = R keys_j + t}).
    res_deg: optional list of the residual rotation angle (degrees) of every fragment instead of a hashed angle below
    15 degrees; pairs of fragments with small residuals are then recoverable from the coarse group rotation alone.
    """
    tb = tables or default_tables()
    base_f = unit_features(K, seed, "scene_base")
    base_k = hash_uniform(seed, "scene_keys", K * 3).astype(np.float64).reshape(K, 3) * 3.0
    feats, keys, poses, perms = [], [], [], []
    for f in range(nfrag):
        u = hash_uniform(seed, f"scene_meta{f}", 16).astype(np.float64)
        gi = int(u[0] * G) % G
        ax = u[1:4] - 0.5
        ax = ax / np.sqrt((ax[0] * ax[0] + ax[1] * ax[1]) + ax[2] * ax[2])
        half = 0.5 * np.deg2rad(15.0) * u[4] if res_deg is None else 0.5 * np.deg2rad(float(res_deg[f]))
        s = half - half ** 3 / 6.0
        c = np.sqrt(1.0 - s * s)
        Rres = quat_to_mat64(np.array([c, ax[0] * s, ax[1] * s, ax[2] * s]))
        Rg = tb.R64[gi]
        R = np.empty((3, 3))
        for i in range(3):
            for j in range(3):
                R[i, j] = (Rres[i, 0] * Rg[0, j] + Rres[i, 1] * Rg[1, j]) + Rres[i, 2] * Rg[2, j]
        t = (u[5:8] - 0.5) * 2.0
        perm = np.argsort(hash_uniform(seed, f"scene_perm{f}", K), kind="stable")
        ff = base_f[perm][:, :, tb.P[gi]] + np.float32(noise) * _gauss(seed, f"scene_fn{f}", K * F * G).reshape(K, F, G)
        is_out = hash_uniform(seed, f"scene_out{f}", K) < outlier_frac
        fresh = unit_features(K, seed, f"scene_fresh{f}")
        ff[is_out] = fresh[is_out]
        kk = _apply_rt(base_k[perm], R, t) + key_noise * _gauss(seed, f"scene_kn{f}", K * 3, np.float64).reshape(K, 3)
        fk = hash_uniform(seed, f"scene_ok{f}", K * 3).astype(np.float64).reshape(K, 3) * 3.0
        kk[is_out] = fk[is_out]
        feats.append(np.ascontiguousarray(_unit_norm_rows(ff)))
        keys.append(np.ascontiguousarray(kk))
        poses.append(np.concatenate([R, t[:, None]], 1))
        perms.append(perm)
    pairs = [(i, j) for i in range(nfrag) for j in range(i + 1, nfrag)]
    gt = {}
    for (i, j) in pairs:                       # keys_i = Ri Rj^T (keys_j - tj) + ti
        Ri, ti, Rj, tj = poses[i][:, :3], poses[i][:, 3], poses[j][:, :3], poses[j][:, 3]
        Rij = np.empty((3, 3))
        for a in range(3):
            for b in range(3):
                Rij[a, b] = (Ri[a, 0] * Rj[b, 0] + Ri[a, 1] * Rj[b, 1]) + Ri[a, 2] * Rj[b, 2]
        tij = np.array([ti[a] - ((Rij[a, 0] * tj[0] + Rij[a, 1] * tj[1]) + Rij[a, 2] * tj[2]) for a in range(3)])
        T = np.eye(4)
        T[:3, :3] = Rij
        T[:3, 3] = tij
        gt[(i, j)] = T
    return dict(feats=feats, keys=keys, poses=poses, pairs=pairs, gt=gt)


def write_scene_files(scene, root, cache_scene_dir=None, lo_pairs=None):
    """Lay a make_scene() result out as the reference expects: {root}/PointCloud/gt.log + gt.info (Redwood format),
    {root}/Keypoints_PC/cloud_bin_{k}Keypoints.npy and, optionally, the FCGF_Input_Group_feature cache.
    lo_pairs: a second ground-truth pair list written as gtLo.log / gtLo.info beside gt.log - the layout of 3DLoMatch, which
    shares 3DMatch's scene directories and differs only in its pair list (utils/dataset.py:176-182)."""
    import os
    n = len(scene["feats"])
    os.makedirs(f"{root}/PointCloud", exist_ok=True)
    os.makedirs(f"{root}/Keypoints_PC", exist_ok=True)
    for stem, plist in (("gt", scene["pairs"]), ("gtLo", lo_pairs)):
        if plist is None:
            continue
        with open(f"{root}/PointCloud/{stem}.log", "w") as f:
            for (i, j) in plist:
                T = scene["gt"][(i, j)]
                f.write(f"{i}\t{j}\t{n}\n")
                for r in range(4):
                    f.write("\t".join(repr(float(v)) for v in T[r]) + "\n")
        with open(f"{root}/PointCloud/{stem}.info", "w") as f:
            for (i, j) in plist:
                f.write(f"{i}\t{j}\t{n}\n")
                for r in range(6):
                    f.write("\t".join(repr(float(1.0 + 0.25 * r if r == c else 0.0)) for c in range(6)) + "\n")
    for k in range(n):
        np.save(f"{root}/Keypoints_PC/cloud_bin_{k}Keypoints.npy", scene["keys"][k])
    if cache_scene_dir is not None:
        os.makedirs(f"{cache_scene_dir}/FCGF_Input_Group_feature", exist_ok=True)
        for k in range(n):
            np.save(f"{cache_scene_dir}/FCGF_Input_Group_feature/{k}.npy", scene["feats"][k])


def surface_cloud(n, seed=0, extent=1.2):
    """n points (f64) on a few random planes / a sphere inside a cube of side `extent` metres: a stand-in for an indoor
    scan (surface-like occupancy, ~1/3 of a voxel neighbourhood filled) for the FCGF backbone tests and benches."""
    rs = np.random.RandomState(seed)
    parts = []
    for _ in range(4):
        o = rs.rand(3) * extent
        u, v = rs.randn(3), rs.randn(3)
        u /= np.linalg.norm(u)
        v -= u * (u @ v)
        v /= np.linalg.norm(v)
        ab = (rs.rand(n // 5, 2) - 0.5) * extent
        parts.append(o + ab[:, :1] * u + ab[:, 1:] * v)
    d = rs.randn(n - 4 * (n // 5), 3)
    parts.append(extent * 0.5 + 0.3 * extent * d / np.linalg.norm(d, axis=1, keepdims=True))
    pc = np.concatenate(parts) + rs.randn(n, 3) * 0.002
    rs.shuffle(pc)
    return np.ascontiguousarray(pc.astype(np.float64))


def train_batch(bn, tables_P, seed=0):
    """One synthetic training batch in the layout of the reference's Train_val_list items (utils/dataset.py:263-270 after
    the DataLoader's batch dimension of 1): feats0 / feats1 (1,bn,32,60), true_idx (1,bn), deltaR (1,bn,4).  feats1 is
    feats0 with the group axis permuted by P[true_idx] plus noise, so the coarse-rotation index is learnable."""
    rs = np.random.RandomState(seed)
    f0 = unit_features(bn, seed=seed + 1000)
    idx = rs.randint(0, 60, size=bn).astype(np.int64)
    f1 = np.stack([f0[i][:, tables_P[idx[i]]] for i in range(bn)]) + 0.05 * rs.randn(bn, 32, 60).astype(np.float32)
    f1 = (f1 / np.linalg.norm(f1, axis=1, keepdims=True)).astype(np.float32)
    q = np.concatenate([np.ones((bn, 1)), 0.1 * rs.randn(bn, 3)], 1)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    return {"feats0": f0[None], "feats1": f1[None], "true_idx": idx[None], "deltaR": q[None]}


def tensor_digest(a):
    """small fingerprint of a (large) tensor for golden files: [l2 norm, sum, dot with a fixed pattern] + first 16 values"""
    from .weights import hash_uniform
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    pat = hash_uniform(12345, "digest", a.size).astype(np.float64) - 0.5
    head = np.zeros(16)
    head[:min(16, a.size)] = a[:16]
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum(), (a * pat).sum()], head])
