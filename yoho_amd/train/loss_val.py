"""Losses and validation metrics of the training path - same names and numbers as the reference's train/loss_val.py
(`name2loss`, `name2val`; Batch_hard_Rindex_loss :21-53, L2_loss :66-71, Validation_PartI :75-141, Validation_PartII
:143-198), organised around three small tensor helpers instead of per-sample Python loops.  Pure torch; everything runs on
whatever device the network outputs live on (the reference hard-codes .cuda())."""
import numpy as np
import torch
import torch.nn.functional as F

from .. import hip


def to_cuda(batch):
    """move every tensor of a (possibly nested) batch to the current HIP device (utils/utils.py to_cuda)"""
    if isinstance(batch, torch.Tensor):
        return batch.cuda()
    if isinstance(batch, dict):
        return {key: to_cuda(val) for key, val in batch.items()}
    if isinstance(batch, (list, tuple)):
        return type(batch)(to_cuda(val) for val in batch)
    return batch


def _pairwise_sq(a, b):
    """(n,d), (m,d) -> (n,m) squared Euclidean distances by explicit differences (as the reference: no GEMM expansion)"""
    return (a[:, None, :] - b[None, :, :]).pow(2).sum(-1)


def _nn_hits(inv0, inv1):
    """boolean (n,): is row i of inv0 the nearest neighbour of row i of inv1?  (recall / recall_index, :80-95)"""
    nearest = torch.argmin(torch.sqrt(_pairwise_sq(inv1, inv0)), dim=1)
    return nearest == torch.arange(inv0.shape[0], device=inv0.device)


def quaternion_angle_deg(q_pre, q_gt, eps=1e-7):
    """rotation angle between predicted (normalised here, norm clamped at 1e-4) and ground-truth unit quaternions, in
    degrees: acos(1 - 2 max(1 - <p,g>^2, eps))  (Validation_PartII.diff_cal :147-156, all rows at once)"""
    p = q_pre / torch.clamp_min(torch.norm(q_pre, dim=1, keepdim=True), 1e-4)
    miss = torch.clamp_min(1.0 - (p * q_gt).sum(1) ** 2, eps)
    return torch.acos(1 - 2 * miss) * (180.0 / np.pi)


class Batch_hard_Rindex_loss:
    """5 x batch-hard margin loss on the invariant descriptors (log-softmax over the squared distances of a batch row,
    margin 0.3, hardest negative after a +1 penalty on the diagonal) + cross-entropy of the 60 rotation-alignment scores of
    the equivariant descriptors against the true coarse rotation (:21-53)."""
    keys = ['triplet_ranking_Rindex_loss']

    def __init__(self, cfg):
        tables = hip.get_context(so3_dir=getattr(cfg, 'SO3_related_files', None)).tables
        self.perm = torch.from_numpy(tables.P.astype(np.int64))                  # P[a, g]
        self.margin, self.weight = 0.3, 5.0

    def rotation_scores(self, eqv0, eqv1):
        """score[b, a] = sum_{f,g} eqv0[b, f, P[a, g]] * eqv1[b, f, g]   (eqvloss :27-31)"""
        P = self.perm.to(eqv0.device)
        return torch.einsum('bfag,bfg->ba', eqv0[:, :, P], eqv1)

    def __call__(self, out):
        target = out['DR_true_index'].long()
        logp = F.log_softmax(_pairwise_sq(out['feats0_inv'], out['feats1_inv']), dim=1)
        positive = logp.diagonal()
        hardest = (logp + torch.eye(logp.shape[0], device=logp.device)).min(dim=1).values
        ranking = torch.clamp_min(positive - hardest + self.margin, 0).mean()
        alignment = F.cross_entropy(self.rotation_scores(out['feats0_eqv_af_conv'], out['feats1_eqv_af_conv']), target)
        return self.weight * ranking + alignment


class _SumLoss:
    def __init__(self, cfg, fn):
        self.fn = fn

    def __call__(self, pred, gt):
        return self.fn(pred, gt, reduction='sum')


class L1_loss(_SumLoss):
    keys = ['L1_Loss']

    def __init__(self, cfg):
        super().__init__(cfg, F.smooth_l1_loss)


class L2_loss(_SumLoss):
    keys = ['L2_Loss']

    def __init__(self, cfg):
        super().__init__(cfg, F.mse_loss)


class Validation_PartI:
    """per-batch loss and recall, recall over the whole validation set, and the share of correctly matched keypoint pairs
    whose coarse rotation index is right as well (:97-141)"""

    def __init__(self, cfg):
        self.cfg = cfg
        self.loss = name2loss[cfg.loss_type](cfg)

    def recall(self, data):
        return _nn_hits(data['feats0_inv'], data['feats1_inv']).float().mean()

    def recall_index(self, data):
        return torch.nonzero(_nn_hits(data['feats0_inv'], data['feats1_inv'])).flatten()

    @torch.no_grad()
    def __call__(self, model, eval_dataset):
        model.eval()
        inv0, inv1, losses, recalls, dr_right = [], [], [], [], []
        for batch in eval_dataset:
            out = model(to_cuda(batch))
            inv0.append(out['feats0_inv'].cpu())
            inv1.append(out['feats1_inv'].cpu())
            dr_right.append((out['DR_true_index'] == out['DR_pre_index']).cpu())
            losses.append(float(self.loss(out)))
            recalls.append(float(self.recall(out)))
        whole = {'feats0_inv': torch.cat(inv0), 'feats1_inv': torch.cat(inv1)}
        matched = self.recall_index(whole)
        return {'val_loss': torch.tensor(losses).mean(), 'whole_recall': self.recall(whole), 'batch_recall': torch.tensor(recalls).mean(),
                'PartI_DR_ability': torch.cat(dr_right)[matched].float().mean().item()}


class Validation_PartII:
    """mean residual-rotation error in degrees, its histogram over [0,6) degrees in 1-degree bins, loss, and PartI's
    coarse-rotation accuracy on the validation set (:158-198)"""

    def __init__(self, cfg):
        self.cfg = cfg
        self.loss = name2loss[cfg.loss_type](cfg)

    def diff_cal(self, R_pre, R_gt):
        return list(quaternion_angle_deg(R_pre, R_gt))

    def static(self, errors):
        whole = errors.long()
        return torch.bincount(whole[whole < 6], minlength=6).float() / errors.shape[0]

    @torch.no_grad()
    def __call__(self, model, eval_dataset):
        model.eval()
        ability, losses, errs = [], [], []
        for batch in eval_dataset:
            gt = torch.squeeze(batch['deltaR']).cpu()
            out = model(to_cuda(batch))
            quat = out['quaternion_pre'].cpu()
            ability.append(float(out['part1_ability']))
            losses.append(float(self.loss(quat, gt)))
            errs.append(quaternion_angle_deg(quat, gt))
        errs = torch.cat(errs)
        return {'val_loss': torch.tensor(losses).mean(), 'R_error': errs.mean(), 'part1_ability': torch.tensor(ability).mean(),
                'R_error_statics': self.static(errs)}


name2loss = {'Batch_hard_Rindex_loss': Batch_hard_Rindex_loss, 'L2_loss_partII': L2_loss}
name2val = {'Val_partI': Validation_PartI, 'Val_partII': Validation_PartII}
