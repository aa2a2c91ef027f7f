"""Drop-in for simple_yoho/fcgf_feat.py: the FCGF sparse-conv backbone on the HIP library (csrc/sparse.hip).

    ext = fcgf_extractor('model/Backbone/best_val_checkpoint.pth')
    ds_points, feats = ext.run(pc, voxel_size=0.025)          # (n,3) ndarray, (n,32) cpu Tensor, unit rows

Same class / method names and return values as the reference (`fcgf_extractor.extract_features` returns
``(sel, F.cpu())`` with ``sel`` a LongTensor so that ``run`` can do ``pc[inds.numpy()]``, simple_yoho/fcgf_feat.py:33-54).
The checkpoint is the FCGF format the reference reads (:18-29): ``{'config': <namespace with model, model_n_out,
normalize_feature, conv1_kernel_size>, 'state_dict': ...}``; a dict with the same keys (or ``state_dict`` + explicit
arguments) is accepted as well, so that no MinkowskiEngine / easydict is needed to unpickle anything.
"""
import os

import numpy as np
import torch

from . import hip

# CHANNELS / TR_CHANNELS of the model classes (fcgf_model/resunet.py:14-16, 193-246); IN variants share the BN layout only
# for NORM_TYPE - instance-norm blocks are not supported.
MODEL_CHANNELS = {
    "ResUNet2": ((0, 32, 64, 128, 256), (0, 32, 64, 64, 128)),
    "ResUNetBN2": ((0, 32, 64, 128, 256), (0, 32, 64, 64, 128)),
    "ResUNetBN2B": ((0, 32, 64, 128, 256), (0, 64, 64, 64, 64)),
    "ResUNetBN2C": ((0, 32, 64, 128, 256), (0, 64, 64, 64, 128)),
    "ResUNetBN2D": ((0, 32, 64, 128, 256), (0, 64, 64, 128, 128)),
    "ResUNetBN2E": ((0, 128, 128, 128, 256), (0, 64, 128, 128, 128)),
}


def _cfg_get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class fcgf_extractor():
    def __init__(self, pth='model/Backbone/best_val_checkpoint.pth', ctx=None):
        self.pth_fn = pth
        self.device = "cuda"
        self.ctx = ctx if ctx is not None else hip.get_context()
        self._load_model()

    def _load_model(self):
        checkpoint = self.pth_fn if isinstance(self.pth_fn, dict) else torch.load(self.pth_fn, map_location="cpu", weights_only=False)
        config = checkpoint['config']
        name = _cfg_get(config, 'model', 'ResUNetBN2C')
        if name not in MODEL_CHANNELS:
            raise ValueError(f"FCGF model {name} is not supported (BN ResUNet2 family only)")
        ch, tr = MODEL_CHANNELS[name]
        if name == "ResUNet2":
            raise ValueError("ResUNet2 has no normalisation layers (NORM_TYPE None); use a BN variant")
        # config.normalize_feature goes to the library as it is: when it is set the model normalises (fcgf_model/resunet.py:187-190)
        # and simple_yoho/fcgf_feat.py:48-49 normalises once more (the library's row kernel then runs both passes); when it is not,
        # only the second normalisation happens - one pass, the same roundings as the reference.  Unit rows either way.
        self._load_args = dict(sd=checkpoint['state_dict'], channels=ch, tr_channels=tr,
                               out_channels=int(_cfg_get(config, 'model_n_out', 32)),
                               conv1_kernel_size=int(_cfg_get(config, 'conv1_kernel_size', 7)), in_channels=1,
                               normalize_feature=bool(_cfg_get(config, 'normalize_feature', True)))
        self.ctx.load_fcgf(owner=self, **self._load_args)

    def _resident(self, ctx=None):
        ctx = self.ctx if ctx is None else ctx
        if ctx.fcgf_owner is not self:                # another backbone object loaded its weights into the shared context
            ctx.load_fcgf(owner=self, **self._load_args)

    def lane_context(self):
        """A second library context with this backbone's weights: its own workspace, so that a backbone pass queued on another
        stream can build its coordinate / kernel maps while the convolutions of the previous pass still run out of the first
        context's workspace (yoho_extractor's two-lane pipeline).  The context is process-wide (hip.get_context(lane=1)): its workspace is sized once."""
        if getattr(self, "_lane_ctx", None) is None:
            self._lane_ctx = hip.get_context(self.ctx.device, self.ctx.tables.dir, lane=1)
        self._resident(self._lane_ctx)
        return self._lane_ctx

    def extract_features_dev(self, pts, voxel_size):
        """HBM-resident variant: pts (n,3) f64 cuda -> (sel int64 cuda, F (m,32) f32 cuda); no host copies."""
        self._resident()
        sel, coords = self.ctx.fcgf_voxelize(pts, voxel_size)
        return sel, self.ctx.fcgf_forward(coords)

    MAX_VOXELS_PER_PASS = int(os.environ.get("YOHO_FCGF_MAX_VOXELS", "1600000"))      # level-0 matrices are 96 columns wide: the 2 GiB gather window holds 5.5 M voxels

    def extract_features_dev_batch(self, pts_list, voxel_size):
        """several clouds (f64 cuda) in one backbone pass -> list of (sel, F)."""
        self._resident()
        vox = [self.ctx.fcgf_voxelize(p, voxel_size) for p in pts_list]
        # one pass addresses at most 2 GiB per feature matrix (<= 256 channels): split long lists by a voxel budget
        feats, group, rows = [], [], 0
        for _, c in vox:
            if group and (rows + c.shape[0] > self.MAX_VOXELS_PER_PASS or len(group) == 64):
                feats += self.ctx.fcgf_forward_batch(group)
                group, rows = [], 0
            group.append(c)
            rows += c.shape[0]
        if group:
            feats += self.ctx.fcgf_forward_batch(group)
        return [(sel, f) for (sel, _), f in zip(vox, feats)]

    def extract_rotated_batch(self, pts, rotations, voxel_size, ctx=None):
        """the backbone on rotated copies of one cloud: pts (n,3) f64 cuda, rotations = list of (3,3) R (p' = R p) ->
        list of (sel, F, rotated selected points (m,3) f32).  The copies are never materialised: rotation, voxelisation and
        the down-sampled points come from one pass over pts (yoho_fcgf_voxelize_rotated).  ctx: the library context whose
        workspace the pass uses (default: the extractor's; lane_context() for a pass queued on a second stream)."""
        ctx = self.ctx if ctx is None else ctx
        self._resident(ctx)
        vox = []
        for b0 in range(0, len(rotations), 64):        # one library call (one count read-back) per 64 copies
            vox += ctx.fcgf_voxelize_rotated_batch(pts, rotations[b0:b0 + 64], voxel_size)
        feats, group, rows = [], [], 0
        for _, c, _ in vox:
            if group and (rows + c.shape[0] > self.MAX_VOXELS_PER_PASS or len(group) == 64):
                feats += ctx.fcgf_forward_batch(group)
                group, rows = [], 0
            group.append(c)
            rows += c.shape[0]
        if group:
            feats += ctx.fcgf_forward_batch(group)
        return [(sel, f, ps) for (sel, _, ps), f in zip(vox, feats)]

    def extract_features(self, pc, voxel_size):
        pts = torch.from_numpy(np.ascontiguousarray(np.asarray(pc, dtype=np.float64))).cuda()
        sel, F = self.extract_features_dev(pts, voxel_size)
        return sel.cpu(), F.cpu()

    def run(self, pc, voxel_size=0.025):
        # get features. inds is the indexes in the input pc (indexes of down-sampled keypoints)
        inds, feat = self.extract_features(pc, voxel_size)
        # downsampled-kpts, feat w l2 normalization
        return pc[inds.numpy()], feat
