"""Drop-in for the reference's tests/evaluator.py: per-scene orchestration of the stage classes, Feature-Match-Recall and
Registration Recall.  Same class names (Evaluator_PartI = YOHO-C, Evaluator_PartII = YOHO-O), cfg attributes (extractor,
matcher, estimator, descriptor, fmr_ratio, ok_match_dist_threshold, RR_dist_threshold, testset_name) and outputs
(data/results.log, Eval_results/*/result.txt through RR_cal).

A scene is a list of stages, each `stage(dataset)` reading / writing the reference's stage cache (tests/evaluator.py:41-47,
:112-117); `eval` walks the scenes, scores the matches against the ground truth and hands the pre.log files to RR_cal.
For many GPUs use yoho_amd.run_dataset.eval_sharded (same result files, pairs sharded over the ranks).
"""
import os
import numpy as np

from . import RR_cal
from .dataset import get_dataset
from .utils import transform_points, dataset_feature_name
from .extractor import name2extractor, extractor_dr_index
from .matcher import name2matcher
from .estimator import name2estimator


def match_inlier_ratio(keys0, keys1, matches, gt, dist_threshold):
    """fraction of the matches (M,2) whose keypoints agree under the ground-truth transform gt (3,4) within dist_threshold"""
    gap = keys0[matches[:, 0]] - transform_points(keys1[matches[:, 1]], gt)
    return np.mean(np.sqrt(np.sum(np.square(gap), axis=-1)) < dist_threshold)


class _Evaluator:
    yoho_sign = None

    def __init__(self, cfg, max_iter):
        self.cfg, self.max_iter = cfg, max_iter
        self.extractor = name2extractor[cfg.extractor](cfg)
        self.matcher = name2matcher[cfg.matcher](cfg)
        self.drindex_extractor = extractor_dr_index(cfg)
        self.estimator = name2estimator[self._estimator_name()](cfg)

    def _estimator_name(self):
        return self.cfg.estimator

    def stages(self, dataset):
        raise NotImplementedError

    def run_onescene(self, dataset):
        for stage in self.stages(dataset):
            stage(dataset)

    def Feature_match_Recall(self, dataset, ratio=0.05):
        """tests/evaluator.py:49-71 / :120-142 -> (share of pairs whose inlier ratio exceeds `ratio`, the ratios)"""
        keys_dir = f'{self.cfg.origin_data_dir}/{dataset_feature_name(dataset.name)}/Keypoints_PC'
        match_dir = f'{self.cfg.output_cache_fn}/Testset/{dataset.name}/Match'
        keys = {}

        def kp(i):
            if i not in keys:
                keys[i] = np.load(f'{keys_dir}/cloud_bin_{i}Keypoints.npy')
            return keys[i]
        ratios = np.array([match_inlier_ratio(kp(a), kp(b), np.load(f'{match_dir}/{a}-{b}.npy'), dataset.get_transform(a, b),
                                              self.cfg.ok_match_dist_threshold) for a, b in dataset.pair_ids])
        return np.mean(ratios > ratio), ratios

    def eval(self, datasets=None, results_log='data/results.log'):
        """tests/evaluator.py:75-101 / :146-173.  `datasets` may be passed in (dict as get_dataset returns)."""
        if datasets is None:
            datasets = get_dataset(self.cfg, False)
        scene_fmr, ratios = [], []
        for scene, dataset in datasets.items():
            if scene == 'wholesetname':
                continue
            self.run_onescene(dataset)
            print(f'eval the FMR result on {dataset.name}')
            fmr, r = self.Feature_match_Recall(dataset, ratio=self.cfg.fmr_ratio)
            scene_fmr.append(fmr)
            ratios.append(r)
        scene_fmr, ratios = np.array(scene_fmr), np.concatenate(ratios, axis=0)
        rr, c_flags, c_errors = RR_cal.benchmark(self.cfg, datasets, self.max_iter, yoho_sign=self.yoho_sign)
        c = self.cfg
        msg = (f"{datasets['wholesetname']}-{c.descriptor}-{c.extractor}-{c.matcher}-{c.estimator}-{self.max_iter}iterations\n"
               f"correct ratio avg {np.mean(ratios):.5f}\n"
               f"correct ratio>0.05 avg {np.mean(scene_fmr):.5f}  std {np.std(scene_fmr):.5f}\n"
               f"Mean_Registration_Recall {rr}\n")
        if os.path.dirname(results_log):
            os.makedirs(os.path.dirname(results_log), exist_ok=True)
        with open(results_log, 'a') as f:
            f.write(msg + '\n')
        print(msg)
        return rr, scene_fmr, ratios


class Evaluator_PartI(_Evaluator):
    """tests/evaluator.py:29-101: descriptor -> matches -> coarse rotations -> YOHO-C"""
    yoho_sign = 'YOHO_C'

    def _estimator_name(self):
        return 'yohoc_mul' if self.max_iter > 500 else self.cfg.estimator       # :37-39

    def stages(self, dataset):
        seq = [] if dataset.name[0:4] == '3dLo' else [self.extractor.Extract]   # 3DLoMatch shares 3DMatch's descriptors (:43-44)
        return seq + [self.matcher.match, self.drindex_extractor.PartI_Rindex, lambda d: self.estimator.ransac(d, self.max_iter)]


class Evaluator_PartII(_Evaluator):
    """tests/evaluator.py:103-173: matches -> coarse rotations -> PartII hypotheses -> YOHO-O"""
    yoho_sign = 'YOHO_O'

    def stages(self, dataset):
        return [self.matcher.match, self.drindex_extractor.PartI_Rindex, self.extractor.PartII_R_pre,
                lambda d: self.estimator.ransac(d, self.max_iter)]


name2evaluator = {
    'PartI': Evaluator_PartI,
    'PartII': Evaluator_PartII
}
