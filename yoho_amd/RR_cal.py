"""Registration-Recall benchmark (3DMatch / Redwood protocol) - drop-in for the reference's utils/RR_cal.py
(the Gojcic/Huang script), numpy only.

Differences from the reference, none of them numerical:
  * `nibabel.quaternions.mat2quat` is restated here (symmetric 4x4 K matrix + eigh, the algorithm nibabel and the
    reference's own utils/r_eval.py:67-84 share), so nibabel is not needed;
  * `np.float` / `np.int` (removed from numpy >= 1.24) are spelled float / int;
  * rotation / translation errors are computed in numpy f64 instead of torch.
Cited line numbers refer to utils/RR_cal.py.
"""
import math
import os
from collections import defaultdict

import numpy as np


def mat2quat(M):
    """nibabel.quaternions.mat2quat (w, x, y, z), w >= 0.  Used by computeTransformationErr (:61)."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M, dtype=np.float64).flat
    K = np.array([[Qxx - Qyy - Qzz, 0, 0, 0],
                  [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
                  [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
                  [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q = q * -1
    return q


def rotation_error(R1, R2):
    """:13-33  r_e = arccos((trace(R1^T R2) - 1) / 2) in degrees, (b,3,3) x (b,3,3) -> (b,1)."""
    R1 = np.asarray(R1, dtype=np.float64); R2 = np.asarray(R2, dtype=np.float64)
    R_ = np.matmul(np.transpose(R1, (0, 2, 1)), R2)
    e = ((np.trace(R_, axis1=1, axis2=2) - 1) / 2)[:, None]
    e = np.clip(e, -1, 1)
    return 180.0 * np.arccos(e) / math.pi


def translation_error(t1, t2):
    """:35-45  (b,3,1) x (b,3,1) -> (b,)"""
    d = np.asarray(t1, dtype=np.float64) - np.asarray(t2, dtype=np.float64)
    return np.sqrt(np.sum(d * d, axis=(1, 2)))


def computeTransformationErr(trans, info):
    """:47-66  RMSE approximation of the Redwood protocol."""
    t = trans[:3, 3]
    r = trans[:3, :3]
    q = mat2quat(r)
    er = np.concatenate([t, q[1:]], axis=0)
    p = er.reshape(1, 6) @ info @ er.reshape(6, 1) / info[0, 0]
    return p.item()


def read_trajectory(filename, dim=4):
    """Redwood .log: blocks of one header line "id0 <tab> id1 <tab> n" + dim matrix rows (:68-102).
    -> (keys (n,3) str array, traj (n,dim,dim))"""
    with open(filename) as fh:
        rows = fh.read().splitlines()
    block = dim + 1
    headers = [[tok.strip() for tok in rows[b].split('\t')[:3]] for b in range(0, len(rows), block)]
    body = [rows[b + 1 + r].split('\t')[:dim] for b in range(0, len(rows) - dim, block) for r in range(dim)]
    return np.asarray(headers), np.asarray(body, dtype=float).reshape(-1, dim, dim)


def read_pre_trajectory(filename, dim=4):
    """pre.log has the same layout (:104-140)"""
    return read_trajectory(filename, dim)


def read_trajectory_info(filename, dim=6):
    """Redwood .info: per pair one header "i j n_frames" + a dim x dim information matrix (:142-170) -> (n_frames, (n,dim,dim))"""
    with open(filename) as fh:
        rows = [r for r in fh.read().splitlines()]
    per = dim + 1
    if len(rows) % per:
        raise AssertionError("malformed .info file")
    n_frames = 0
    mats = np.empty((len(rows) // per, dim, dim))
    for b in range(len(rows) // per):
        n_frames = int(rows[b * per].split()[2])
        mats[b] = [[float(v) for v in rows[b * per + 1 + r].split()] for r in range(dim)]
    return n_frames, mats


def extract_corresponding_trajectors(est_pairs, gt_pairs, gt_traj):
    """ground-truth pose of every estimated pair (:172-191); like the reference it overwrites the third header column of
    est_pairs with the ground truth's fragment count before matching rows"""
    out = np.zeros((len(est_pairs), 4, 4))
    n_frag = gt_pairs[0][2]
    for k, pair in enumerate(est_pairs):
        pair[2] = n_frag
        out[k] = gt_traj[np.where((gt_pairs == pair).all(axis=1))[0]]
    return out


def write_trajectory(traj, metadata, filename, dim=4):
    """:193-212"""
    with open(filename, 'w') as fh:
        for meta, pose in zip(metadata, traj):
            if not meta[2]:
                continue
            fh.write('\t'.join(str(m) for m in meta) + '\n')
            fh.write('\n'.join('\t'.join('{0:.12f}'.format(v) for v in pose[r]) for r in range(dim)) + '\n')


def evaluate_registration(num_fragment, result, result_pairs, gt_pairs, gt, gt_info, err2=0.2, nonconsecutive=True):
    """3DMatch registration benchmark for one scene (:236-317) -> precision, recall, flags, errors.
    flags: 0 = correct, 1 = wrong, 2 = pair not evaluated.  Ground-truth pairs are addressed through their row number in
    the .log, and - as in the reference - row 0 can therefore never be looked up (its table entry is the "empty" value 0);
    in the consecutive protocol (WHU-TLS) the first estimate is scored against row 0 directly instead."""
    thr = err2 ** 2
    row_of = {}
    for row in range(gt_pairs.shape[0]):
        i, j = int(gt_pairs[row, 0]), int(gt_pairs[row, 1])
        if row > 0 and (not nonconsecutive or abs(j - i) > 1):
            row_of[(i, j)] = row
    n_gt = len(row_of) + (0 if nonconsecutive else 1)
    flags, errors = [], []
    good = scored = 0

    def score(est_pose, row):
        nonlocal good, scored
        scored += 1
        e2 = computeTransformationErr(np.linalg.inv(gt[row]) @ est_pose, gt_info[row])
        errors.append(np.sqrt(e2))
        good += e2 <= thr
        flags.append(0 if e2 <= thr else 1)

    first = 0
    if not nonconsecutive:
        score(result[0], 0)
        first = 1
    for k in range(first, result_pairs.shape[0]):
        row = row_of.get((int(result_pairs[k, 0]), int(result_pairs[k, 1])))
        if row is None:
            flags.append(2)
        else:
            score(result[k], row)
    precision = good * 1.0 / (scored if scored else 1e6)
    # n_gt is a numpy integer in the reference (np.sum of the mask): a scene without a scorable gt pair gives nan (0/0, numpy
    # warning), not a ZeroDivisionError
    return precision, good * 1.0 / np.int64(n_gt), flags, errors


def benchmark(cfg, datasets, max_iter, yoho_sign='YOHO_O'):
    """All scenes of a test set (:321-399): writes
    {output_cache_fn}/Testset/{wholesetname}/Eval_results/{yoho_sign}_RR/{max_iter}iters/result.txt and returns
    (mean registration recall over the scenes, per-scene flags, per-scene errors)."""
    whole = datasets['wholesetname']
    nonconsecutive = whole != 'WHU-TLS'
    out_dir = f'{cfg.output_cache_fn}/Testset/{whole}/Eval_results/{yoho_sign}_RR/{max_iter}iters'
    os.makedirs(out_dir, exist_ok=True)
    c_flags, c_errors = {}, {}
    med_re, med_te, precisions, recalls, n_valids = [], [], [], [], []
    with open(f'{out_dir}/result.txt', 'w') as rep:
        rep.write("Scene\t prec.\t rec.\t re\t te\t samples\t\n")
        for scene, ds in datasets.items():
            if scene == 'wholesetname':
                continue
            pre_log = os.path.join(f'{cfg.output_cache_fn}/Testset/{ds.name}/Match/{yoho_sign}/{max_iter}iters', 'pre.log')
            gt_stem = ds.gt_dir[:ds.gt_dir.rfind('.')]
            gt_pairs, gt_traj = read_trajectory(gt_stem + '.log')
            n_valid = sum(1 for p in gt_pairs if (not nonconsecutive) or abs(int(p[0]) - int(p[1])) > 1)
            n_frag, gt_cov = read_trajectory_info(gt_stem + '.info')
            print(pre_log)
            est_pairs, est_traj = read_pre_trajectory(pre_log)
            prec, rec, flags, errs = evaluate_registration(n_frag, est_traj, est_pairs, gt_pairs, gt_traj, gt_cov,
                                                           err2=cfg.RR_dist_threshold, nonconsecutive=nonconsecutive)
            c_flags[ds.name], c_errors[ds.name] = flags, errs
            gt_of_est = extract_corresponding_trajectors(est_pairs, gt_pairs, gt_traj)
            ok = np.array(flags) == 0
            re = rotation_error(gt_of_est[:, :3, :3], est_traj[:, :3, :3])[ok]
            te = translation_error(gt_of_est[:, :3, 3:4], est_traj[:, :3, 3:4])[ok]
            if re.shape[0] == 0:
                re = np.full([n_valid], 180.0)
            if te.shape[0] == 0:
                te = np.ones([n_valid])
            med_re.append(np.median(re)); med_te.append(np.median(te))
            precisions.append(prec); recalls.append(rec); n_valids.append(n_valid)
            rep.write("{}\t {:.3f}\t {:.3f}\t {:.3f}\t {:.3f}\t {:3d}\n".format(ds.name, prec, rec, np.median(re), np.median(te), n_valid))
            rep.write("Mean precision: {:.3f}".format(prec))
            rep.write("Registration Recall: {:.3f}\n".format(rec))
            rep.write("Mean median RRE: {:.3f}: +- {:.3f}\n".format(np.mean(re), np.median(re)))
            rep.write("Mean median RTE: {:.3F}: +- {:.3f}\n".format(np.mean(te), np.median(te)))
        recall_mean = np.mean(np.array(recalls))                     # the number the papers report
        rep.write("Mean precision: {:.3f}: +- {:.3f}\n".format(np.mean(precisions), np.std(precisions)))
        rep.write("Weighted precision: {:.3f}\n".format((np.array(n_valids) * np.array(precisions)).sum() / np.sum(n_valids)))
        rep.write("Registration Recall: {:.3f}: +- {:.3f}\n".format(recall_mean, np.std(np.array(recalls))))
        rep.write("Mean median RRE: {:.3f}: +- {:.3f}\n".format(np.mean(med_re), np.std(med_re)))
        rep.write("Mean median RTE: {:.3F}: +- {:.3f}\n".format(np.mean(med_te), np.std(med_te)))
    return recall_mean, c_flags, c_errors
