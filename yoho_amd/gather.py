"""The 60-fold FCGF keypoint-feature gather (reference: YOHO_testset.py:153-166).

For every group element g the backbone (yoho_amd.fcgf_feat, or any other source of per-cloud features)
yields the down-sampled rotated cloud ``pts_g (n_g,3) f32`` and unit-norm features ``feat_g (n_g,32)``.
The gather rotates the keypoints by R_g in f64, finds each rotated keypoint's nearest cloud point
(f64, sqrt(D2+1e-7) form, first minimum; brute force or the hash grid of csrc/gridnn.hip) and copies its feature row into
``out[:, :, g]`` - one yoho_group_gather launch per group element, output assembled in HBM.
"""
import numpy as np
import torch

from . import hip


def gather_group_features(keys, pts_list, feat_list, ctx=None, out=None, voxel_size=None):
    """keys (K,3) f64 (ndarray or cuda tensor); pts_list/feat_list: 60 arrays -> (K,32,60) f32 cuda tensor.
    voxel_size: the voxel size the clouds were down-sampled with, if known - the searches then run through a hash grid
    (yoho_set_nn_grid; identical result, ~9x faster at 10^5 points)."""
    ctx = ctx or hip.get_context()
    dev = lambda a, dt: (a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))).to(device="cuda", dtype=dt).contiguous()
    k = dev(keys, torch.float64)
    if len(pts_list) != 60 or len(feat_list) != 60:
        raise ValueError("need one (points, features) pair per group element (60)")
    if out is None:
        out = torch.empty((k.shape[0], 32, 60), dtype=torch.float32, device="cuda")
    if voxel_size:
        ctx.set_nn_grid(voxel_size)
    try:
        for g in range(60):
            ctx.group_gather(k, dev(pts_list[g], torch.float32), dev(feat_list[g], torch.float32), g, out)
    finally:
        if voxel_size:
            ctx.set_nn_grid(0)
    return out
