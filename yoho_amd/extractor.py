"""Drop-in for the reference's tests/extractor.py (same class / method names, cfg attributes and
.npy stage-cache layout), running on the HIP library.

  extractor_PartI(cfg).Extract(dataset)              tests/extractor.py:18-60
  extractor_dr_index(cfg).{Des2R_torch, Batch_Des2R_torch, PartI_Rindex}   :64-100
  extractor_PartII(cfg).{batch_create, PartII_R_pre}                        :104-201
"""
import os
import numpy as np
import torch

from . import hip, store
from .network import name2network
from .utils import make_non_exists_dir, dataset_feature_name
from . import weights as W


def _load_best(network, best_model_fn, strict):
    if os.path.exists(best_model_fn):
        sd, best_para = W.load_checkpoint(best_model_fn)
        network.load_state_dict(sd, strict=strict)
        print(f'Resuming best para {best_para}')
    else:
        raise ValueError("No model exists")


class extractor_PartI():
    def __init__(self, cfg):
        self.cfg = cfg
        self.network = name2network[f'{self.cfg.test_network_type}'](self.cfg).cuda()
        self.model_fn = f'{self.cfg.model_fn}/{self.cfg.train_network_type}/model.pth'
        self.best_model_fn = f'{self.cfg.model_fn}/{self.cfg.train_network_type}/model_best.pth'

    def _load_model(self):
        _load_best(self.network, self.best_model_fn, strict=True)

    def Extract(self, dataset):
        # input 5000*32*60 (FCGF group feature) -> output 5000*32*60 (equivariant descriptor)
        self._load_model()
        self.network.eval()
        FCGF_input_dir = f'{self.cfg.output_cache_fn}/Testset/{dataset.name}/FCGF_Input_Group_feature'
        YOHO_output_dir = f'{self.cfg.output_cache_fn}/Testset/{dataset.name}/YOHO_Output_Group_feature'
        make_non_exists_dir(YOHO_output_dir)
        print(f'Extracting the PartI descriptors on {dataset.name}')
        ctx = self.network.ctx
        for pc_id in dataset.pc_ids:
            out_fn = f'{YOHO_output_dir}/{pc_id}.npy'
            if os.path.exists(out_fn):
                continue
            x = store.load_npy(f'{FCGF_input_dir}/{pc_id}.npy')        # H2D once, stays resident for PartII
            # the reference chunks by cfg.test_batch_size (900); the kernel takes the whole fragment
            out = ctx.partI_forward(x.contiguous(), want_inv=False, want_inv_np=True)
            store.save_npy(out_fn, out["eqv"])
            store.put(os.path.abspath(out_fn) + "#inv_np", out["inv_np"])


class extractor_dr_index():
    def __init__(self, cfg):
        self.cfg = cfg
        self.ctx = hip.get_context(so3_dir=getattr(cfg, "SO3_related_files", None))
        self.Nei_in_SO3 = torch.from_numpy(self.ctx.tables.P.reshape([-1]))   # 60_60.npy flattened (reference attribute name)

    @staticmethod
    def _dev(t):
        if not isinstance(t, torch.Tensor):
            t = torch.from_numpy(np.asarray(t, dtype=np.float32))
        return t.to(device="cuda", dtype=torch.float32).contiguous()

    def Des2R_torch(self, des1_eqv, des2_eqv):       # beforerot afterrot, (32,60) each
        return self.ctx.des2r(self._dev(des1_eqv)[None], self._dev(des2_eqv)[None])[0]

    def Batch_Des2R_torch(self, des1_eqv, des2_eqv):  # (B,32,60) each -> (B,) int64 (device tensor)
        return self.ctx.des2r(self._dev(des1_eqv), self._dev(des2_eqv))

    def PartI_Rindex(self, dataset):
        match_dir = f'{self.cfg.output_cache_fn}/Testset/{dataset.name}/Match'
        Save_dir = f'{match_dir}/DR_index'
        make_non_exists_dir(Save_dir)
        datasetname = dataset_feature_name(dataset.name)
        Feature_dir = f'{self.cfg.output_cache_fn}/Testset/{datasetname}/YOHO_Output_Group_feature'
        print(f'extract the drindex of the matches on {dataset.name}')
        for pair in dataset.pair_ids:
            id0, id1 = pair
            if os.path.exists(f'{Save_dir}/{id0}-{id1}.npy'):
                continue
            match_pps = store.load_npy(f'{match_dir}/{id0}-{id1}.npy', dtype=torch.int64)
            feats0 = store.load_npy(f'{Feature_dir}/{id0}.npy')
            feats1 = store.load_npy(f'{Feature_dir}/{id1}.npy')
            f0 = feats0[match_pps[:, 0]]
            f1 = feats1[match_pps[:, 1]]
            pre_idxs = self.Batch_Des2R_torch(f1, f0)
            store.save_npy(f'{Save_dir}/{id0}-{id1}.npy', pre_idxs)


class extractor_PartII():
    def __init__(self, cfg):
        self.cfg = cfg
        self.network = name2network[f'{self.cfg.test_network_type}'](self.cfg).cuda()
        self.model_fn = f'{self.cfg.model_fn}/{self.cfg.train_network_type}/model.pth'
        self.best_model_fn = f'{self.cfg.model_fn}/{self.cfg.train_network_type}/model_best.pth'
        self.Rgroup = self.network.ctx.tables.R32

    def _load_model(self):
        if os.path.exists(self.best_model_fn):
            print(self.best_model_fn)
        _load_best(self.network, self.best_model_fn, strict=False)

    def batch_create(self, feats0_fcgf, feats1_fcgf, feats0_yoho, feats1_yoho, index_pre, start, end):
        # attention (tests/extractor.py:125-138): feats0 -> feats1_in_batch for it is afterrot
        t = lambda a: torch.from_numpy(np.asarray(a[start:end]).astype(np.float32)) if not isinstance(a, torch.Tensor) else a[start:end]
        i = index_pre[start:end]
        i = torch.from_numpy(np.asarray(i).astype(np.int64)) if not isinstance(i, torch.Tensor) else i
        return {
            'before_eqv0': t(feats1_fcgf),   # exchanged
            'before_eqv1': t(feats0_fcgf),
            'after_eqv0': t(feats1_yoho),
            'after_eqv1': t(feats0_yoho),
            'pre_idx': i,
        }

    def PartII_R_pre(self, dataset):
        self._load_model()
        self.network.eval()
        ctx = self.network.ctx
        match_dir = f'{self.cfg.output_cache_fn}/Testset/{dataset.name}/Match'
        DRindex_dir = f'{match_dir}/DR_index'
        Save_dir = f'{match_dir}/Trans_pre'
        make_non_exists_dir(Save_dir)
        datasetname = dataset_feature_name(dataset.name)
        FCGF_dir = f'{self.cfg.output_cache_fn}/Testset/{datasetname}/FCGF_Input_Group_feature'
        YOHO_dir = f'{self.cfg.output_cache_fn}/Testset/{datasetname}/YOHO_Output_Group_feature'
        print(f'extracting the PartII feature on {dataset.name}')
        for pair in dataset.pair_ids:
            id0, id1 = pair
            if os.path.exists(f'{Save_dir}/{id0}-{id1}.npy'):
                continue
            pps = store.load_npy(f'{match_dir}/{id0}-{id1}.npy', dtype=torch.int64)
            m0, m1 = pps[:, 0], pps[:, 1]
            feats0_fcgf = store.load_npy(f'{FCGF_dir}/{id0}.npy')[m0]
            feats1_fcgf = store.load_npy(f'{FCGF_dir}/{id1}.npy')[m1]
            feats0_yoho = store.load_npy(f'{YOHO_dir}/{id0}.npy')[m0]
            feats1_yoho = store.load_npy(f'{YOHO_dir}/{id1}.npy')[m1]
            Index_pre = store.load_npy(f'{DRindex_dir}/{id0}-{id1}.npy', dtype=torch.int64)
            Keys0 = torch.from_numpy(np.ascontiguousarray(dataset.get_kps(id0), dtype=np.float64)).cuda()[m0].contiguous()
            Keys1 = torch.from_numpy(np.ascontiguousarray(dataset.get_kps(id1), dtype=np.float64)).cuda()[m1].contiguous()
            # one launch for all matches (the reference loops over batches of cfg.test_batch_size = 1000)
            batch = self.batch_create(feats0_fcgf, feats1_fcgf, feats0_yoho, feats1_yoho, Index_pre, 0, pps.shape[0])
            out = self.network(batch)
            # R = quat2mat(q) @ Rgroup[idx];  t = key0 - key1 @ R.T   (tests/extractor.py:187-199)
            Trans = ctx.hyp_from_quat(out['quaternion_pre'], Index_pre.contiguous(), Keys0, Keys1)
            store.save_npy(f'{Save_dir}/{id0}-{id1}.npy', Trans)


name2extractor = {
    'PartI': extractor_PartI,
    'PartII': extractor_PartII
}
