"""Drop-in for YOHO_testset.py: "PC*60 rotations -> FCGF backbone -> FCGF group feature for PC keypoints".

    testset_create(cfg).batch_feature_extraction()

writes ``{output_dir}/Testset/{dataset}/{scene}/FCGF_Input_Group_feature/{pc_id}.npy`` ((K,32,60) f32, group axis in the
order of Rotation.npy) for every fragment of every scene, exactly the files tests/extractor.py:46-49 reads.

Reference flow (YOHO_testset.py:20-166), kept step for step: rotate the fragment by R_g (f64), voxelise (first point of
every voxel, :39-49), run the backbone (:143-147), rotate the keypoints by R_g and take, for each, the feature row of its
nearest down-sampled point (:153-159, KNN(1), type-promoted f64 'L2' distance).  Here the fragment is uploaded once and
all 60 group elements run on the device (yoho_fcgf_voxelize / yoho_fcgf_forward / yoho_group_gather); there is no
DataLoader batching (the reference's batch of 4 clouds only amortises MinkowskiEngine launches).

``cfg`` needs ``model`` (FCGF checkpoint path or dict), ``voxel_size``, ``dataset``; optional ``output_dir`` /
``origin_dir`` (defaults './data/YOHO_FCGF', './data/origin_data' as the reference, :61-62) and ``datasets`` (a prebuilt
{scene: dataset} dict, otherwise ``get_dataset_name(dataset, origin_dir)``).
"""
import os

import numpy as np
import torch

from . import hip
from .dataset import get_dataset_name
from .fcgf_feat import fcgf_extractor
from .utils import make_non_exists_dir


class testset_create():
    def __init__(self, config, ctx=None):
        self.config = config
        self.dataset_name = self.config.dataset
        self.output_dir = getattr(config, 'output_dir', './data/YOHO_FCGF')
        self.origin_dir = getattr(config, 'origin_dir', './data/origin_data')
        self.datasets = getattr(config, 'datasets', None) or get_dataset_name(self.dataset_name, self.origin_dir)
        self.ctx = ctx if ctx is not None else hip.get_context()
        self.Rgroup = self.ctx.tables.R64
        self.fcgf = fcgf_extractor(self.config.model, ctx=self.ctx)

    def fragment_group_features(self, pc, keys):
        """pc (N,3), keys (K,3) f64 -> (K,32,60) f32 cuda tensor (one fragment, all 60 group elements)."""
        pc_d = torch.from_numpy(np.ascontiguousarray(np.asarray(pc, dtype=np.float64))).cuda()
        k_d = torch.from_numpy(np.ascontiguousarray(np.asarray(keys, dtype=np.float64))).cuda()
        out = torch.empty((k_d.shape[0], 32, 60), dtype=torch.float32, device="cuda")
        nb = 15                                                      # rotated copies per backbone pass (split further by voxel count)
        self.ctx.set_nn_grid(self.config.voxel_size)                 # the targets are one point per voxel: grid search, same winners
        try:
            for g0 in range(0, 60, nb):
                # rotated copies (pc @ R_g^T, :143) are never materialised: rotation, voxelisation and 'dspcd0' (the
                # down-sampled points, .float(), :92) come out of one pass over the cloud
                res = self.fcgf.extract_rotated_batch(pc_d, [self.Rgroup[g] for g in range(g0, g0 + nb)], self.config.voxel_size)
                for j, (sel, feat, pts) in enumerate(res):
                    self.ctx.group_gather(k_d, pts, feat, g0 + j, out)   # keys @ R_g^T, f64 NN, feature row -> out[:, :, g]
        finally:
            self.ctx.set_nn_grid(0)
        return out

    def Feature_extracting(self):
        for scene, dataset in self.datasets.items():
            if scene == 'wholesetname':
                continue
            save_dir = f'{self.output_dir}/Testset/{self.dataset_name}/{scene}/FCGF_Input_Group_feature'
            make_non_exists_dir(save_dir)
            for pc_id in dataset.pc_ids:
                out = self.fragment_group_features(dataset.get_pc(pc_id), dataset.get_kps(pc_id))
                np.save(f'{save_dir}/{pc_id}.npy', out.cpu().numpy())

    def batch_feature_extraction(self):
        self.Feature_extracting()
