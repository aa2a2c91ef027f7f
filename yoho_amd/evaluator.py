"""Drop-in for the reference's tests/evaluator.py: per-scene orchestration
extractor -> matcher -> dr_index -> (PartII) -> estimator, Feature-Match-Recall and Registration-Recall.
Same class names / cfg attributes (extractor, matcher, estimator, descriptor, fmr_ratio, ok_match_dist_threshold,
RR_dist_threshold, testset_name) and the same data/results.log + result.txt outputs."""
import os
import numpy as np

from . import RR_cal
from .dataset import get_dataset
from .utils import transform_points, dataset_feature_name
from .extractor import name2extractor, extractor_dr_index
from .matcher import name2matcher
from .estimator import name2estimator


class _EvaluatorBase:
    yoho_sign = None

    def Feature_match_Recall(self, dataset, ratio=0.05):
        """tests/evaluator.py:49-71 / :120-142"""
        Keys_dir = f'{self.cfg.origin_data_dir}/{dataset_feature_name(dataset.name)}/Keypoints_PC'
        pair_fmrs = []
        for pair in dataset.pair_ids:
            id0, id1 = pair
            matches = np.load(f'{self.cfg.output_cache_fn}/Testset/{dataset.name}/Match/{id0}-{id1}.npy')
            keys0 = np.load(f'{Keys_dir}/cloud_bin_{id0}Keypoints.npy')[matches[:, 0], :]
            keys1 = np.load(f'{Keys_dir}/cloud_bin_{id1}Keypoints.npy')[matches[:, 1], :]
            gt = dataset.get_transform(id0, id1)
            keys1 = transform_points(keys1, gt)
            dist = np.sqrt(np.sum(np.square(keys0 - keys1), axis=-1))
            pair_fmrs.append(np.mean(dist < self.cfg.ok_match_dist_threshold))      # ok ratio in one pair
        pair_fmrs = np.array(pair_fmrs)                                             # ok ratios in one scene
        FMR = np.mean(pair_fmrs > ratio)                                            # FMR in one scene
        return FMR, pair_fmrs

    def eval(self, datasets=None, results_log='data/results.log'):
        """tests/evaluator.py:75-101 / :146-173.  `datasets` may be passed in (dict as get_dataset returns)."""
        if datasets is None:
            datasets = get_dataset(self.cfg, False)
        FMRS, all_pair_fmrs = [], []
        for scene, dataset in datasets.items():
            if scene == 'wholesetname':
                continue
            self.run_onescene(dataset)
            print(f'eval the FMR result on {dataset.name}')
            FMR, pair_fmrs = self.Feature_match_Recall(dataset, ratio=self.cfg.fmr_ratio)
            FMRS.append(FMR)
            all_pair_fmrs.append(pair_fmrs)
        FMRS = np.array(FMRS)
        all_pair_fmrs = np.concatenate(all_pair_fmrs, axis=0)
        datasetname = datasets['wholesetname']
        Mean_Registration_Recall, c_flags, c_errors = RR_cal.benchmark(self.cfg, datasets, self.max_iter, yoho_sign=self.yoho_sign)
        msg = f'{datasetname}-{self.cfg.descriptor}-{self.cfg.extractor}-{self.cfg.matcher}-{self.cfg.estimator}-{self.max_iter}iterations\n'
        msg += f'correct ratio avg {np.mean(all_pair_fmrs):.5f}\n' \
               f'correct ratio>0.05 avg {np.mean(FMRS):.5f}  std {np.std(FMRS):.5f}\n' \
               f'Mean_Registration_Recall {Mean_Registration_Recall}\n'
        if os.path.dirname(results_log):
            os.makedirs(os.path.dirname(results_log), exist_ok=True)
        with open(results_log, 'a') as f:
            f.write(msg + '\n')
        print(msg)
        return Mean_Registration_Recall, FMRS, all_pair_fmrs


class Evaluator_PartI(_EvaluatorBase):
    """tests/evaluator.py:29-101 (YOHO-C)"""
    yoho_sign = 'YOHO_C'

    def __init__(self, cfg, max_iter):
        self.max_iter = max_iter
        self.cfg = cfg
        self.extractor = name2extractor[self.cfg.extractor](self.cfg)
        self.matcher = name2matcher[self.cfg.matcher](self.cfg)
        self.drindex_extractor = extractor_dr_index(self.cfg)
        est = self.cfg.estimator
        if self.max_iter > 500:
            est = 'yohoc_mul'
        self.estimator = name2estimator[est](self.cfg)

    def run_onescene(self, dataset):
        if not dataset.name[0:4] == '3dLo':
            self.extractor.Extract(dataset)
        self.matcher.match(dataset)
        self.drindex_extractor.PartI_Rindex(dataset)
        self.estimator.ransac(dataset, self.max_iter)


class Evaluator_PartII(_EvaluatorBase):
    """tests/evaluator.py:103-173 (YOHO-O)"""
    yoho_sign = 'YOHO_O'

    def __init__(self, cfg, max_iter):
        self.max_iter = max_iter
        self.cfg = cfg
        self.extractor = name2extractor[self.cfg.extractor](self.cfg)
        self.matcher = name2matcher[self.cfg.matcher](self.cfg)
        self.estimator = name2estimator[self.cfg.estimator](self.cfg)
        self.drindex_extractor = extractor_dr_index(self.cfg)

    def run_onescene(self, dataset):
        self.matcher.match(dataset)
        self.drindex_extractor.PartI_Rindex(dataset)
        self.extractor.PartII_R_pre(dataset)
        self.estimator.ransac(dataset, self.max_iter)


name2evaluator = {
    'PartI': Evaluator_PartI,
    'PartII': Evaluator_PartII
}
