"""Drop-in for tests/estimator.py: YOHO-C (3-match Kabsch RANSAC) and YOHO-O (one-shot hypothesis
vote) on the HIP library, same class / method names, .npz outputs and pre.log.

Two sampling modes for YOHO-C:

* parity (default).  The reference draws from the global ``np.random`` stream (tests/estimator.py:122,126).
  The same draws are made on the host, in the same order (<= 1000 index triples per pair; the library continues numpy's
  legacy MT19937 stream in C, ``yoho_c_draw_np``), and handed to the GPU (``yoho_c_ransac``), so that with the same seed the
  same hypotheses are scored.  The reference's
  Kabsch has no determinant fix (:59-60) and a 3-point covariance has rank <= 2, so LAPACK decides the sign
  of the null direction - about half of the reference's hypotheses are reflections.  With
  ``cfg.yohoc_lapack_parity`` (default True) that sign comes from one batched ``np.linalg.svd`` on the host
  and is passed as a mask.  It cannot come from the device: the sign is that of a singular value which is
  zero up to rounding, i.e. it is decided by the last bits of one particular LAPACK build's
  bidiagonalisation (reflector count) and QR sweep - reproducing it needs that library, not its algorithm.
* throughput (``cfg.yohoc_device_sampling = True``).  Nothing but one 64-bit seed (drawn from
  ``np.random``) crosses to the device: statistic, sampling, Kabsch and vote run there
  (``yoho_c_ransac_device``; Philox stream, proper rotations only).  Same estimator, a different random
  sequence - what the multi-GPU driver and ``bench.py``'s YOHO-C leg use.

YOHO-O shuffles the hypothesis order on the host (one permutation per pair, :321-323) and scores on the GPU.
"""
import numpy as np
import torch

from . import hip
from .utils import transform_points, make_non_exists_dir, dataset_feature_name

G = 60
NO_ESTIMATE = 50001            # recalltime the reference stores when no rotation bucket has two matches (:107)


def format_log_entry(id0, id1, n_fragments, trans):
    """one Redwood-trajectory record of pre.log (tests/estimator.py:18-23): header, the 3 rows of trans, '0 0 0 1'"""
    rows = "".join("\t".join(f"{v}" for v in trans[r][:4]) + "\n" for r in range(3))
    return f"{int(id0)}\t{int(id1)}\t{int(n_fragments)}\n{rows}{0.0}\t{0.0}\t{0.0}\t{1.0}\n"


def write_pre_log(path, n_fragments, entries):
    """entries: iterable of (id0, id1, trans) in the order they are to appear"""
    with open(path, "w") as f:
        f.write("".join(format_log_entry(a, b, n_fragments, T) for a, b, T in entries))


def R_pre_log(dataset, save_dir):
    """tests/estimator.py:12-24: collect the per-pair .npz results of save_dir into save_dir/pre.log, dataset.pair_ids order"""
    n = len(dataset.pc_ids)
    write_pre_log(f"{save_dir}/pre.log", n,
                  ((a, b, np.load(f"{save_dir}/{a}-{b}.npz", allow_pickle=True)["trans"]) for a, b in dataset.pair_ids))


def _cu(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).cuda()


def rotation_angle_deg(R_gt, R):
    """angle of R_gt^T R in degrees (utils/r_eval.py:112-119 computes the same angle from quaternions)"""
    c = (np.trace(R_gt.T @ R) - 1.0) / 2.0
    return np.rad2deg(np.abs(np.arccos(np.clip(c, -1.0, 1.0))))


def _det3(A):
    """determinants of a stack of 3x3 matrices by cofactor expansion (only their sign is used)"""
    return (A[:, 0, 0] * (A[:, 1, 1] * A[:, 2, 2] - A[:, 1, 2] * A[:, 2, 1]) - A[:, 0, 1] * (A[:, 1, 0] * A[:, 2, 2] - A[:, 1, 2] * A[:, 2, 0])
            + A[:, 0, 2] * (A[:, 1, 0] * A[:, 2, 1] - A[:, 1, 1] * A[:, 2, 0]))


def draw_seed():
    """a 63-bit seed for the device sampler, taken from the global np.random stream (so np.random.seed governs it)"""
    return int(np.random.randint(0, 2 ** 63 - 1, dtype=np.int64))


class _Base:
    inliner_dist = None

    def overlap_cal(self, key_m0, key_m1, T):
        """inlier ratio of the matches under T (tests/estimator.py:66-70 / :286-290)"""
        residual = key_m0 - transform_points(key_m1, T)
        return np.mean(np.sum(residual * residual, axis=-1) < self.inliner_dist * self.inliner_dist)

    def transdiff(self, gt, pre):
        """(rotation error in degrees, translation error) between two [R|t] (tests/estimator.py:72-75)"""
        shift = gt[:3, 3] - pre[:3, 3]
        return rotation_angle_deg(gt[:3, :3], pre[:3, :3]), np.sqrt(np.sum(shift * shift))


class yohoc(_Base):
    """tests/estimator.py:28-141.  TWO modes, one contract each (timed side by side in bench.py's `yohoc` leg):

    * host-parity (default; cfg.yohoc_lapack_parity=True, cfg.yohoc_device_sampling=False): REFERENCE-EXACT.  The host consumes
      the global np.random stream exactly as the reference's loop does (:119-128: one weighted draw of a coarse rotation, three
      matches of its bucket with replacement, per accepted iteration) and decides, with one batched np.linalg.svd over the sampled
      triples, where LAPACK's sign of the null direction makes the reference's R = Vt.T @ U.T a reflection (:55-63 has no
      determinant fix; 3 centred points give a rank-2 covariance).  The device does everything else: Kabsch for all iterations,
      the optional reflection, the inlier vote, the first strict maximum.  Per-iteration counts, winner and transform are the
      reference's (atol 1e-9, goldens chain*.npz / scene*.npz).  Cost per pair at M = 3233, 1000 iterations: ~3.5 ms of host work (the
      draws continue numpy's stream in C, hip.c_draw_np: 0.3 ms where 2000 np.random.choice calls took 54; one stacked
      np.linalg.svd: 2.5 ms) + the device call.
    * device sampling (cfg.yohoc_device_sampling=True; pipeline.run_pair(estimator="yohoc"), the dataset driver, bench.py):
      STATISTICAL parity.  Statistic, sampling (Philox4x32-10 keyed by a seed), Kabsch and vote run on the device, no host work but
      the launches; proper rotations only.  The random stream is not numpy's MT19937, so results equal the reference's in
      distribution (same success flags / RR on the goldens), and bit-exactly equal oracle/yoho_oracle.yohoc_device_triples +
      yohoc_select for the same seed.
    """

    def __init__(self, cfg):
        self.cfg = cfg
        self.inliner_dist = cfg.ransac_c_inlinerdist
        self.lapack_parity = bool(getattr(cfg, "yohoc_lapack_parity", True))
        self.device_sampling = bool(getattr(cfg, "yohoc_device_sampling", False))
        self.ctx = hip.get_context(so3_dir=getattr(cfg, "SO3_related_files", None))

    def DR_statictic(self, DR_indexs):
        """tests/estimator.py:34-51: matches bucketed by coarse rotation (ascending match index) and the sampling
        weight of every bucket, n (n - 0.01)(n - 0.02) with n = size / 100 for buckets of at least two matches."""
        idx = np.asarray(DR_indexs).astype(np.int64).reshape(-1)
        size = np.bincount(idx, minlength=G)[:G]
        by_rot = np.split(np.argsort(idx, kind="stable"), np.cumsum(size)[:-1])
        buckets = {i: by_rot[i].tolist() for i in range(G)}
        n = size / 100.0
        weight = np.where(size >= 2, n * (n - 0.01) * (n - 0.02), 0.0)
        total = np.sum(weight)
        if total < 1e-4:
            return None, None
        return buckets, weight / total

    def _statistic_flat(self, DR_indexs):
        """DR_statictic without the per-rotation Python lists: (prob or None, (bucket_start, bucket_members))"""
        start, members = self.bucket_table(DR_indexs)
        size = np.diff(start)
        n = size / 100.0
        weight = np.where(size >= 2, n * (n - 0.01) * (n - 0.02), 0.0)
        total = np.sum(weight)
        if total < 1e-4:
            return None, (start, members)
        return weight / total, (start, members)

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
        """1 where LAPACK's SVD makes the reference's R = Vt.T @ U.T a reflection, for (I,3,3) sampled triples.
        The sign hangs on the rounding of the covariance, so each 3x3 has to carry the bits of the expression the reference
        evaluates (tests/estimator.py:56-58: centre with np.mean, then a 3x3 matmul): np.mean over the first axis of a (3,3)
        array adds the rows in order and divides by 3, which is what the reduction over axis 1 of the (I,3,3) stack does per
        triple; np.matmul hands every 3x3 of a stack to the same BLAS call a single (3,3).T @ (3,3) gets (transposed view
        included).  tests/test_host_cpu.py checks both against the one-triple-at-a-time expression, bit for bit."""
        c0 = k0s - np.mean(k0s, axis=1, keepdims=True)
        c1 = k1s - np.mean(k1s, axis=1, keepdims=True)
        cov = np.matmul(np.swapaxes(c1, 1, 2), c0)
        U, _, VT = np.linalg.svd(cov)
        return (_det3(U) * _det3(VT) < 0).astype(np.uint8)          # orthogonal factors: the determinants are +-1 up to rounding

    @staticmethod
    def _reflect_mask_one_by_one(k0s, k1s):
        """_reflect_mask with every covariance formed by the reference's own expression on one triple (what the vectorised form
        is tested against)"""
        cov = np.empty((k0s.shape[0], 3, 3))
        for i in range(k0s.shape[0]):
            cov[i] = (k1s[i] - np.mean(k1s[i], 0, keepdims=True)).T @ (k0s[i] - np.mean(k0s[i], 0, keepdims=True))
        U, _, VT = np.linalg.svd(cov)
        return (np.linalg.det(U) * np.linalg.det(VT) < 0).astype(np.uint8)

    @staticmethod
    def bucket_table(DR_indexs):
        """(bucket_start (61,), bucket_members (M,)): the matches of coarse rotation r in ascending order are
        bucket_members[bucket_start[r]:bucket_start[r + 1]] (the lists DR_statictic returns, flat)"""
        idx = np.asarray(DR_indexs).astype(np.int64).reshape(-1)
        size = np.bincount(idx, minlength=G)[:G]
        return np.concatenate([[0], np.cumsum(size)]).astype(np.int64), np.argsort(idx, kind="stable").astype(np.int64)

    def _draw(self, buckets, prob, max_iter, table=None):
        """the sampling half of tests/estimator.py:119-128: consumes np.random exactly as the reference's loop does (one weighted
        draw of a rotation, then three matches of its bucket with replacement, per accepted iteration).  The stream is continued
        in C (hip.c_draw_np: numpy's legacy choice / random_sample / randint restated, pinned against numpy in
        tests/test_host_cpu.py) - 2000 interpreter-level np.random.choice calls per pair cost 54 ms, this costs 0.05."""
        if table is None:
            start = np.concatenate([[0], np.cumsum([len(buckets[i]) for i in range(G)])]).astype(np.int64)
            members = np.array([m for i in range(G) for m in buckets[i]], dtype=np.int64)
        else:
            start, members = table
        return hip.c_draw_np(prob, start, members, max_iter)[0]

    def _draw_numpy(self, buckets, prob, max_iter):
        """_draw through np.random.choice itself, call by call as the reference makes them (the checker of _draw)"""
        triples, draws = [], 0
        while len(triples) < max_iter and draws <= 50000:
            draws += 1
            rot = np.random.choice(range(G), p=prob)
            if len(buckets[rot]) >= 2:
                triples.append(np.random.choice(np.array(buckets[rot]), 3))
        return np.array(triples, dtype=np.int64).reshape(-1, 3)

    def estimate_host_sampled(self, km0, km1, dr, max_iter, timings=None):
        """The host-parity mode for one pair's matched keypoints (M,3) f64 and coarse rotations (M,): np.random draws (+ the LAPACK
        sign mask) on the host, Kabsch + vote for all iterations in one device call.  -> None if no rotation bucket has two matches
        (the reference's recalltime 50001), else (trans (3,4) or eye(4), recalltime, winning triple or None).
        timings: optional dict receiving the seconds spent in the host draws, the host SVDs and the device call."""
        import time
        t0 = time.perf_counter()
        prob, table = self._statistic_flat(dr)
        if prob is None:
            return None
        triples = self._draw(None, prob, max_iter, table=table)
        t1 = time.perf_counter()
        trans, recall, tri = np.eye(4), 0, None
        t2 = t1
        if triples.shape[0] > 0:
            refl = _cu(self._reflect_mask(km0[triples], km1[triples]), np.uint8) if self.lapack_parity else None
            t2 = time.perf_counter()
            T, res, _, _ = self.ctx.c_ransac(_cu(km0, np.float64), _cu(km1, np.float64), _cu(triples, np.int64), refl, self.inliner_dist)
            it, cnt = (int(v) for v in res.cpu().numpy())
            if cnt > 0:
                trans, recall, tri = T.cpu().numpy(), it, triples[it - 1]
        if timings is not None:
            t3 = time.perf_counter()
            for k, v in (("draw_s", t1 - t0), ("svd_mask_s", t2 - t1), ("device_call_s", t3 - t2)):
                timings[k] = timings.get(k, 0.0) + v
        return trans, recall, tri

    def _ransac_pair(self, dataset, max_iter, pair, Save_dir, match_dir, Index_dir, Keys_dir):
        id0, id1 = pair
        keys0 = np.load(f'{Keys_dir}/cloud_bin_{id0}Keypoints.npy')
        keys1 = np.load(f'{Keys_dir}/cloud_bin_{id1}Keypoints.npy')
        pps = np.load(f'{match_dir}/{id0}-{id1}.npy')
        dr = np.load(f'{Index_dir}/{id0}-{id1}.npy')
        out = f'{Save_dir}/{id0}-{id1}.npz'
        trans, recall, tri = np.eye(4), 0, None
        if self.device_sampling:
            if len(pps):
                T, res, tris = self.ctx.c_ransac_device(_cu(keys0, np.float64), _cu(keys1, np.float64), _cu(dr, np.int64), max_iter,
                                                        draw_seed(), self.inliner_dist, match=_cu(pps, np.int64), want_triples=True)
                recall, cnt = (int(v) for v in res.cpu().numpy())
                if recall == NO_ESTIMATE:
                    np.savez(out, trans=np.eye(4), center=0, axis=0, recalltime=NO_ESTIMATE)
                    return
                if cnt > 0:
                    trans, tri = T.cpu().numpy(), tris[recall - 1].cpu().numpy()
            else:
                np.savez(out, trans=np.eye(4), center=0, axis=0, recalltime=NO_ESTIMATE)
                return
            km0, km1 = keys0[pps[:, 0]], keys1[pps[:, 1]]
        else:
            km0, km1 = keys0[pps[:, 0]], keys1[pps[:, 1]]
            est = self.estimate_host_sampled(km0, km1, dr, max_iter)
            if est is None:
                np.savez(out, trans=np.eye(4), center=0, axis=0, recalltime=NO_ESTIMATE)
                return
            trans, recall, tri = est
        center = np.concatenate([km0[tri], km1[tri]], axis=0) if tri is not None else np.ones([6, 3])
        np.savez(out, trans=trans, center=center, recalltime=recall)

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
        for id0, id1 in dataset.pair_ids:
            pps = np.load(f'{match_dir}/{id0}-{id1}.npy')
            km0 = dataset.get_kps(id0)[pps[:, 0]]
            km1 = dataset.get_kps(id1)[pps[:, 1]]
            hyps = np.load(f'{Trans_dir}/{id0}-{id1}.npy')
            order = np.arange(hyps.shape[0])
            np.random.shuffle(order)                       # tests/estimator.py:321-323: Trans[index[0:max_iter]]
            H = min(max_iter, hyps.shape[0])
            trans, recall = np.eye(4), 0
            if H > 0:
                res, _ = self.ctx.o_score(_cu(km0, np.float64), _cu(km1, np.float64), _cu(hyps, np.float64), _cu(order, np.int64), H,
                                          self.inliner_dist)
                bh, cnt = (int(v) for v in res.cpu().numpy())
                if cnt > 0:
                    trans, recall = hyps[order[bh]], bh
            np.savez(f'{Save_dir}/{id0}-{id1}.npz', trans=trans, recalltime=recall)
        R_pre_log(dataset, Save_dir)


name2estimator = {
    'yohoc': yohoc,
    'yohoc_mul': yohoc_mul,
    'yohoo': yohoo
}
