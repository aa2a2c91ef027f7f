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
    """:68-102  -> (keys (n,3) str array, traj (n,dim,dim))"""
    with open(filename) as f:
        lines = f.readlines()
    keys = lines[0::(dim + 1)]
    final_keys = []
    for k in keys:
        a = k.split('\t')[0:3]
        final_keys.append([a[0].strip(), a[1].strip(), a[2].strip()])
    traj = []
    for i in range(len(lines)):
        if i % 5 != 0:
            traj.append(lines[i].split('\t')[0:dim])
    traj = np.asarray(traj, dtype=float).reshape(-1, dim, dim)
    return np.asarray(final_keys), traj


def read_pre_trajectory(filename, dim=4):
    """:104-140 (identical parsing for pre.log)"""
    return read_trajectory(filename, dim)


def read_trajectory_info(filename, dim=6):
    """:142-170  -> (n_frame, cov (n,6,6))"""
    with open(filename) as fid:
        contents = fid.readlines()
    n_pairs = len(contents) // 7
    assert (len(contents) == 7 * n_pairs)
    info_list = []
    n_frame = 0
    for i in range(n_pairs):
        frame_idx0, frame_idx1, n_frame = [int(item) for item in contents[i * 7].strip().split()]
        info_matrix = np.concatenate(
            [np.array(item.split(), dtype=float).reshape(1, -1) for item in contents[i * 7 + 1:i * 7 + 7]], axis=0)
        info_list.append(info_matrix)
    cov_matrix = np.asarray(info_list, dtype=float).reshape(-1, dim, dim)
    return n_frame, cov_matrix


def extract_corresponding_trajectors(est_pairs, gt_pairs, gt_traj):
    """:172-191 (mutates est_pairs[:,2] like the reference)"""
    ext_traj = np.zeros((len(est_pairs), 4, 4))
    for est_idx, pair in enumerate(est_pairs):
        pair[2] = gt_pairs[0][2]
        gt_idx = np.where((gt_pairs == pair).all(axis=1))[0]
        ext_traj[est_idx, :, :] = gt_traj[gt_idx, :, :]
    return ext_traj


def write_trajectory(traj, metadata, filename, dim=4):
    """:193-212"""
    with open(filename, 'w') as f:
        for idx in range(traj.shape[0]):
            if metadata[idx][2]:
                p = traj[idx, :, :].tolist()
                f.write('\t'.join(map(str, metadata[idx])) + '\n')
                f.write('\n'.join('\t'.join(map('{0:.12f}'.format, p[i])) for i in range(dim)))
                f.write('\n')


def evaluate_registration(num_fragment, result, result_pairs, gt_pairs, gt, gt_info, err2=0.2, nonconsecutive=True):
    """:236-317  -> precision, recall, flags, errors"""
    err2 = err2 ** 2
    gt_mask = np.zeros((num_fragment, num_fragment), dtype=int)
    flags = []
    errors = []
    if nonconsecutive:
        for idx in range(gt_pairs.shape[0]):
            i = int(gt_pairs[idx, 0]); j = int(gt_pairs[idx, 1])
            if abs(j - i) > 1:                       # only non consecutive pairs are tested
                gt_mask[i, j] = idx
        n_gt = np.sum(gt_mask > 0)
    else:
        for idx in range(gt_pairs.shape[0]):
            i = int(gt_pairs[idx, 0]); j = int(gt_pairs[idx, 1])
            gt_mask[i, j] = idx
        n_gt = np.sum(gt_mask > 0) + 1
    good = 0
    n_res = 0
    if not nonconsecutive:
        start_check = 1
        n_res += 1
        pose = result[0, :, :]
        p = computeTransformationErr(np.linalg.inv(gt[0, :, :]) @ pose, gt_info[0, :, :])
        errors.append(np.sqrt(p))
        if p <= err2:
            good += 1
            flags.append(0)
        else:
            flags.append(1)
    else:
        start_check = 0
    for idx in range(start_check, result_pairs.shape[0]):
        i = int(result_pairs[idx, 0]); j = int(result_pairs[idx, 1])
        pose = result[idx, :, :]
        if gt_mask[i, j] > 0:
            n_res += 1
            gt_idx = gt_mask[i, j]
            p = computeTransformationErr(np.linalg.inv(gt[gt_idx, :, :]) @ pose, gt_info[gt_idx, :, :])
            errors.append(np.sqrt(p))
            if p <= err2:
                good += 1
                flags.append(0)
            else:
                flags.append(1)
        else:
            flags.append(2)
    if n_res == 0:
        n_res += 1e6
    precision = good * 1.0 / n_res
    recall = good * 1.0 / n_gt
    return precision, recall, flags, errors


def benchmark(cfg, datasets, max_iter, yoho_sign='YOHO_O'):
    """:321-399  writes {output_cache_fn}/Testset/{wholesetname}/Eval_results/{yoho_sign}_RR/{max_iter}iters/result.txt"""
    c_flags = {}
    c_errors = {}
    re_per_scene = defaultdict(list)
    te_per_scene = defaultdict(list)
    re_all, te_all, precision, recall = [], [], [], []
    n_valids = []
    nonconsecutive = True
    wholesetname = datasets['wholesetname']
    if wholesetname == 'WHU-TLS':
        nonconsecutive = False
    result_dir = f'{cfg.output_cache_fn}/Testset/{wholesetname}/Eval_results/{yoho_sign}_RR/{max_iter}iters'
    if not os.path.exists(result_dir):
        os.makedirs(result_dir)
    f = open(f'{result_dir}/result.txt', 'w')
    f.write(("Scene\t prec.\t rec.\t re\t te\t samples\t\n"))
    for scene, dataset in datasets.items():
        if scene == 'wholesetname':
            continue
        pre_dir = f'{cfg.output_cache_fn}/Testset/{dataset.name}/Match/{yoho_sign}/{max_iter}iters'
        gt_dir_loc = str.rfind(dataset.gt_dir, '.')
        gt_dir = dataset.gt_dir[0:gt_dir_loc]
        gt_pairs, gt_traj = read_trajectory(f'{gt_dir}.log')
        n_valid = 0
        for ele in gt_pairs:
            if nonconsecutive:
                diff = abs(int(ele[0]) - int(ele[1]))
                n_valid += diff > 1
            else:
                n_valid += 1
        n_valids.append(n_valid)
        n_fragments, gt_traj_cov = read_trajectory_info(f'{gt_dir}.info')
        print(os.path.join(pre_dir, 'pre.log'))
        est_pairs, est_traj = read_pre_trajectory(os.path.join(pre_dir, 'pre.log'))
        temp_precision, temp_recall, c_flag, c_error = evaluate_registration(
            n_fragments, est_traj, est_pairs, gt_pairs, gt_traj, gt_traj_cov, err2=cfg.RR_dist_threshold, nonconsecutive=nonconsecutive)
        c_flags[dataset.name] = c_flag
        c_errors[dataset.name] = c_error
        ext_gt_traj = extract_corresponding_trajectors(est_pairs, gt_pairs, gt_traj)
        ok = np.array(c_flag) == 0
        re = rotation_error(ext_gt_traj[:, 0:3, 0:3], est_traj[:, 0:3, 0:3])[ok]
        te = translation_error(ext_gt_traj[:, 0:3, 3:4], est_traj[:, 0:3, 3:4])[ok]
        if re.shape[0] == 0:
            re = np.ones([n_valid]) * 180
        if te.shape[0] == 0:
            te = np.ones([n_valid])
        for d, v in ((re_per_scene, re), (te_per_scene, te)):
            d['mean'].append(np.mean(v)); d['median'].append(np.median(v)); d['min'].append(np.min(v)); d['max'].append(np.max(v))
        re_all.extend(re.reshape(-1).tolist())
        te_all.extend(te.reshape(-1).tolist())
        precision.append(temp_precision)
        recall.append(temp_recall)
        f.write("{}\t {:.3f}\t {:.3f}\t {:.3f}\t {:.3f}\t {:3d}\n".format(dataset.name, temp_precision, temp_recall, np.median(re), np.median(te), n_valid))
        f.write("Mean precision: {:.3f}".format(temp_precision))
        f.write("Registration Recall: {:.3f}\n".format(temp_recall))
        f.write("Mean median RRE: {:.3f}: +- {:.3f}\n".format(np.mean(re), np.median(re)))
        f.write("Mean median RTE: {:.3F}: +- {:.3f}\n".format(np.mean(te), np.median(te)))
    weighted_precision = (np.array(n_valids) * np.array(precision)).sum() / np.sum(n_valids)
    Registration_Recall = np.mean(np.array(recall))      # most important registration recall for eval
    f.write("Mean precision: {:.3f}: +- {:.3f}\n".format(np.mean(precision), np.std(precision)))
    f.write("Weighted precision: {:.3f}\n".format(weighted_precision))
    f.write("Registration Recall: {:.3f}: +- {:.3f}\n".format(Registration_Recall, np.std(np.array(recall))))
    f.write("Mean median RRE: {:.3f}: +- {:.3f}\n".format(np.mean(re_per_scene['median']), np.std(re_per_scene['median'])))
    f.write("Mean median RTE: {:.3F}: +- {:.3f}\n".format(np.mean(te_per_scene['median']), np.std(te_per_scene['median'])))
    f.close()
    return Registration_Recall, c_flags, c_errors
