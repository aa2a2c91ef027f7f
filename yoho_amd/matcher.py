"""Drop-in for tests/matcher.py: invariant pooling (numpy-order fp32 mean over the group axis) +
mutual nearest neighbour, on the HIP library.  Writes Match/{id0}-{id1}.npy (M,2) int64."""
import os
import numpy as np
import torch

from . import hip, store
from .knn_search import knn_module
from .utils import make_non_exists_dir, dataset_feature_name


class matcher_dual():
    def __init__(self, cfg):
        self.cfg = cfg
        self.KNN = knn_module.KNN(1)
        self.ctx = hip.get_context(so3_dir=getattr(cfg, "SO3_related_files", None))

    def _inv(self, fn):
        t = store.get(os.path.abspath(fn) + "#inv_np")
        if t is None:
            t = self.ctx.group_mean_np(store.load_npy(fn).contiguous())     # == np.mean(feats, axis=-1), bit-exact
            store.put(os.path.abspath(fn) + "#inv_np", t)
        return t

    def match(self, dataset):
        print(f'match the keypoints on {dataset.name}')
        Save_dir = f'{self.cfg.output_cache_fn}/Testset/{dataset.name}/Match'
        make_non_exists_dir(Save_dir)
        datasetname = dataset_feature_name(dataset.name)
        Feature_dir = f'{self.cfg.output_cache_fn}/Testset/{datasetname}/YOHO_Output_Group_feature'
        for pair in dataset.pair_ids:
            id0, id1 = pair
            if os.path.exists(f'{Save_dir}/{id0}-{id1}.npy'):
                continue
            feats0 = self._inv(f'{Feature_dir}/{id0}.npy')
            feats1 = self._inv(f'{Feature_dir}/{id1}.npy')
            # KNN(feats1, feats0) / KNN(feats0, feats1) + the mutual check of tests/matcher.py:37-48
            match_pps = self.ctx.mutual_nn(feats0, feats1)
            store.save_npy(f'{Save_dir}/{id0}-{id1}.npy', match_pps.contiguous())


name2matcher = {
    'Match': matcher_dual
}
