"""Drop-in for tests/estimator.py: YOHO-C (3-match Kabsch RANSAC) and YOHO-O (one-shot hypothesis
vote) on the HIP library, same class / method names, .npz outputs and pre.log.

Randomness: the reference draws from the global ``np.random`` stream (tests/estimator.py:122,126,
322).  These classes draw from the same stream, in the same order, on the host (a permutation /
<= 1000 index triples per pair) and hand the indices to the GPU, so that with the same seed the
same hypotheses are scored.

YOHO-C reflections: the reference's Kabsch has no determinant fix (tests/estimator.py:59-60) and a
3-point covariance has rank <= 2, so LAPACK decides the sign of the null direction - about half of the
reference's hypotheses are reflections.  With ``cfg.yohoc_lapack_parity`` (default True) the sign
is taken from a batched ``np.linalg.svd`` on the host (a few ms per pair) and passed to the kernel
as a mask, reproducing the reference's result; set it False to get proper rotations for every
sample (a strictly stronger estimator, no host SVD).
"""
import numpy as np
import torch

from . import hip
from .utils import transform_points, make_non_exists_dir, dataset_feature_name


def R_pre_log(dataset, save_dir):
    """tests/estimator.py:12-24 (Redwood-format trajectory)."""
    writer = open(f'{save_dir}/pre.log', 'w')
    pair_num = int(len(dataset.pc_ids))
    for pair in dataset.pair_ids:
        pc0, pc1 = pair
        ransac_result = np.load(f'{save_dir}/{pc0}-{pc1}.npz', allow_pickle=True)
        transform_pr = ransac_result['trans']
        writer.write(f'{int(pc0)}\t{int(pc1)}\t{pair_num}\n')
        writer.write(f'{transform_pr[0][0]}\t{transform_pr[0][1]}\t{transform_pr[0][2]}\t{transform_pr[0][3]}\n')
        writer.write(f'{transform_pr[1][0]}\t{transform_pr[1][1]}\t{transform_pr[1][2]}\t{transform_pr[1][3]}\n')
        writer.write(f'{transform_pr[2][0]}\t{transform_pr[2][1]}\t{transform_pr[2][2]}\t{transform_pr[2][3]}\n')
        writer.write(f'{0.0}\t{0.0}\t{0.0}\t{1.0}\n')
    writer.close()


def _cu(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def _compute_R_diff(R_gt, R):
    """utils/r_eval.py:112-119 via the trace (same angle as the quaternion formula)."""
    c = (np.trace(R_gt.T @ R) - 1.0) / 2.0
    return np.rad2deg(np.abs(np.arccos(np.clip(c, -1.0, 1.0))))


class _Base:
    def overlap_cal(self, key_m0, key_m1, T):
        key_m1 = transform_points(key_m1, T)
        diff = np.sum(np.square(key_m0 - key_m1), axis=-1)
        return np.mean(diff < self.inliner_dist * self.inliner_dist)

    def transdiff(self, gt, pre):
        Rdiff = _compute_R_diff(gt[0:3:, 0:3], pre[0:3:, 0:3])
        tdiff = np.sqrt(np.sum(np.square(gt[0:3, 3] - pre[0:3, 3])))
        return Rdiff, tdiff


class yohoc(_Base):
    def __init__(self, cfg):
        self.cfg = cfg
        self.inliner_dist = cfg.ransac_c_inlinerdist
        self.lapack_parity = bool(getattr(cfg, "yohoc_lapack_parity", True))
        self.ctx = hip.get_context(so3_dir=getattr(cfg, "SO3_related_files", None))

    def DR_statictic(self, DR_indexs):
        """tests/estimator.py:34-51"""
        R_index_pre_statistic = {i: [] for i in range(60)}
        for t in range(DR_indexs.shape[0]):
            R_index_pre_statistic[int(DR_indexs[t])].append(t)
        R_index_pre_probability = []
        for i in range(60):
            if len(R_index_pre_statistic[i]) < 2:
                R_index_pre_probability.append(0)
            else:
                num = float(len(R_index_pre_statistic[i])) / 100.0
                R_index_pre_probability.append(num * (num - 0.01) * (num - 0.02))
        R_index_pre_probability = np.array(R_index_pre_probability)
        if np.sum(R_index_pre_probability) < 1e-4:
            return None, None
        R_index_pre_probability = R_index_pre_probability / np.sum(R_index_pre_probability)
        return R_index_pre_statistic, R_index_pre_probability

    def Threepps2Tran(self, kps0_init, kps1_init):
        """tests/estimator.py:55-63 for one triple, computed by the Kabsch kernel (proper rotation
        unless LAPACK parity asks for the reference's reflection)."""
        k0 = np.ascontiguousarray(kps0_init, dtype=np.float64)
        k1 = np.ascontiguousarray(kps1_init, dtype=np.float64)
        tri = torch.arange(3, dtype=torch.int64, device="cuda")[None].contiguous()
        refl = _cu(self._reflect_mask(k0[None], k1[None]), np.uint8) if self.lapack_parity else None
        _, _, T_all, _ = self.ctx.c_ransac(_cu(k0, np.float64), _cu(k1, np.float64), tri, refl, self.inliner_dist, want_all=True)
        return T_all[0].cpu().numpy()

    @staticmethod
    def _reflect_mask(k0s, k1s):
        """det sign LAPACK gives the reference's R = Vt.T @ U.T for every triple ((I,3,3) inputs)."""
        c0 = np.mean(k0s, 1, keepdims=True)
        c1 = np.mean(k1s, 1, keepdims=True)
        m = np.stack([(k1s[i] - c1[i]).T @ (k0s[i] - c0[i]) for i in range(k0s.shape[0])])
        U, S, VT = np.linalg.svd(m)
        det = np.linalg.det(np.transpose(VT, (0, 2, 1)) @ np.transpose(U, (0, 2, 1)))
        return (det < 0).astype(np.uint8)

    def _draw(self, R_index_pre_statistic, R_index_pre_probability, max_iter):
        """the sampling half of tests/estimator.py:119-128 (consumes np.random exactly like the loop)."""
        triples, iter_ransac, exec_time, max_time = [], 0, 0, 50000
        while iter_ransac < max_iter:
            if exec_time > max_time:
                break
            exec_time += 1
            R_index = np.random.choice(range(60), p=R_index_pre_probability)
            if len(R_index_pre_statistic[R_index]) < 2:
                continue
            iter_ransac += 1
            triples.append(np.random.choice(np.array(R_index_pre_statistic[R_index]), 3))   # guarantee the same index
        return np.array(triples, dtype=np.int64).reshape(-1, 3)

    def _ransac_pair(self, dataset, max_iter, pair, Save_dir, match_dir, Index_dir, Keys_dir):
        id0, id1 = pair
        Keys0 = np.load(f'{Keys_dir}/cloud_bin_{id0}Keypoints.npy')
        Keys1 = np.load(f'{Keys_dir}/cloud_bin_{id1}Keypoints.npy')
        pps = np.load(f'{match_dir}/{id0}-{id1}.npy')
        Keys_m0 = Keys0[pps[:, 0]]
        Keys_m1 = Keys1[pps[:, 1]]
        Index = np.load(f'{Index_dir}/{id0}-{id1}.npy')
        R_index_pre_statistic, R_index_pre_probability = self.DR_statictic(Index)
        if R_index_pre_probability is None:
            np.savez(f'{Save_dir}/{id0}-{id1}.npz', trans=np.eye(4), center=0, axis=0, recalltime=50001)
            return
        triples = self._draw(R_index_pre_statistic, R_index_pre_probability, max_iter)
        best_trans_ransac, recall_time = np.eye(4), 0
        best_3p_in_0, best_3p_in_1 = np.ones([3, 3]), np.ones([3, 3])
        if triples.shape[0] > 0:
            refl = _cu(self._reflect_mask(Keys_m0[triples], Keys_m1[triples]), np.uint8) if self.lapack_parity else None
            best_T, res, _, _ = self.ctx.c_ransac(_cu(Keys_m0, np.float64), _cu(Keys_m1, np.float64), _cu(triples, np.int64),
                                                  refl, self.inliner_dist)
            it, cnt = (int(v) for v in res.cpu().numpy())
            if cnt > 0:
                best_trans_ransac, recall_time = best_T.cpu().numpy(), it
                best_3p_in_0, best_3p_in_1 = Keys_m0[triples[it - 1]], Keys_m1[triples[it - 1]]
        np.savez(f'{Save_dir}/{id0}-{id1}.npz', trans=best_trans_ransac,
                 center=np.concatenate([best_3p_in_0, best_3p_in_1], axis=0), recalltime=recall_time)

    def _dirs(self, dataset, max_iter):
        match_dir = f'{self.cfg.output_cache_fn}/Testset/{dataset.name}/Match'
        Index_dir = f'{match_dir}/DR_index'
        Save_dir = f'{match_dir}/YOHO_C/{max_iter}iters'
        Keys_dir = f'{self.cfg.origin_data_dir}/{dataset_feature_name(dataset.name)}/Keypoints_PC'
        return match_dir, Index_dir, Save_dir, Keys_dir

    def ransac(self, dataset, max_iter=1000):
        match_dir, Index_dir, Save_dir, Keys_dir = self._dirs(dataset, max_iter)
        make_non_exists_dir(Save_dir)
        print(f'Ransac with YOHO-C on {dataset.name}:')
        for pair in dataset.pair_ids:
            self._ransac_pair(dataset, max_iter, pair, Save_dir, match_dir, Index_dir, Keys_dir)
        R_pre_log(dataset, Save_dir)


class yohoc_mul(yohoc):
    """tests/estimator.py:145-275.  The reference forks one process per pair, so every pair starts
    from the SAME global np.random state and the parent's state is left untouched; reproduced here
    by re-seeding from the saved state before each pair (no processes: the GPU does the work)."""

    def ransac_once(self, dataset, max_iter, pair):
        match_dir, Index_dir, Save_dir, Keys_dir = self._dirs(dataset, max_iter)
        self._ransac_pair(dataset, max_iter, pair, Save_dir, match_dir, Index_dir, Keys_dir)

    def ransac(self, dataset, max_iter=1000):
        match_dir, Index_dir, Save_dir, Keys_dir = self._dirs(dataset, max_iter)
        make_non_exists_dir(Save_dir)
        print(f'Ransac with YOHO-C on {dataset.name}:')
        state = np.random.get_state()
        for pair in dataset.pair_ids:
            np.random.set_state(state)
            self.ransac_once(dataset, max_iter, pair)
        np.random.set_state(state)
        R_pre_log(dataset, Save_dir)
        print('Done')


class yohoo(_Base):
    def __init__(self, cfg):
        self.cfg = cfg
        self.inliner_dist = cfg.ransac_o_inlinerdist
        self.ctx = hip.get_context(so3_dir=getattr(cfg, "SO3_related_files", None))
        self.Nei_in_SO3 = self.ctx.tables.P.astype(np.float64)
        self.Rgroup = self.ctx.tables.R64

    def ransac(self, dataset, max_iter=1000):
        match_dir = f'{self.cfg.output_cache_fn}/Testset/{dataset.name}/Match'
        Trans_dir = f'{match_dir}/Trans_pre'
        Save_dir = f'{match_dir}/YOHO_O/{max_iter}iters'
        make_non_exists_dir(Save_dir)
        print(f'Ransac with YOHO-O on {dataset.name}:')
        for pair in dataset.pair_ids:
            id0, id1 = pair
            Keys0 = dataset.get_kps(id0)
            Keys1 = dataset.get_kps(id1)
            pps = np.load(f'{match_dir}/{id0}-{id1}.npy')
            Keys_m0 = Keys0[pps[:, 0]]
            Keys_m1 = Keys1[pps[:, 1]]
            Trans = np.load(f'{Trans_dir}/{id0}-{id1}.npy')
            index = np.arange(Trans.shape[0])
            np.random.shuffle(index)
            H = min(max_iter, Trans.shape[0])      # Trans[index[0:max_iter]] (tests/estimator.py:323)
            recall_time, best_trans_ransac = 0, np.eye(4)
            if H > 0:
                res, _ = self.ctx.o_score(_cu(Keys_m0, np.float64), _cu(Keys_m1, np.float64), _cu(Trans, np.float64),
                                          _cu(index, np.int64), H, self.inliner_dist)
                bh, cnt = (int(v) for v in res.cpu().numpy())
                if cnt > 0:
                    recall_time, best_trans_ransac = bh, Trans[index[bh]]
            np.savez(f'{Save_dir}/{id0}-{id1}.npz', trans=best_trans_ransac, recalltime=recall_time)
        R_pre_log(dataset, Save_dir)


name2estimator = {
    'yohoc': yohoc,
    'yohoc_mul': yohoc_mul,
    'yohoo': yohoo
}
