"""Losses and validation: drop-in for train/loss_val.py (Batch_hard_Rindex_loss :21-53, L1/L2 :56-71, Validation_PartI
:75-141, Validation_PartII :143-198).  Pure torch; tensors live wherever the network outputs do (the reference hard-codes
.cuda())."""
import abc

import numpy as np
import torch

from .. import hip


def to_cuda(data):
    """utils/utils.py to_cuda: tensors (and lists / tuples of tensors) of a batch dict -> device."""
    if isinstance(data, (list, tuple)):
        return type(data)(to_cuda(v) for v in data)
    if isinstance(data, dict):
        return {k: to_cuda(v) for k, v in data.items()}
    if isinstance(data, torch.Tensor):
        return data.cuda()
    return data


class Loss(abc.ABC):
    def __init__(self, keys):
        self.keys = list(keys)

    @abc.abstractmethod
    def __call__(self, data_pr, data_gt, **kwargs):
        pass


class Batch_hard_Rindex_loss(Loss):
    def __init__(self, cfg):
        super().__init__(['triplet_ranking_Rindex_loss'])
        tables = hip.get_context(so3_dir=getattr(cfg, 'SO3_related_files', None)).tables
        self.R_perm = torch.from_numpy(tables.P.astype(np.int64).reshape([-1]))
        self.class_loss = torch.nn.CrossEntropyLoss()

    def eqvloss(self, eqvfeat0, eqvfeat1):
        B, F, G = eqvfeat0.shape
        eqvfeat0 = eqvfeat0[:, :, self.R_perm.to(eqvfeat0.device)].reshape([B, F, G, G])
        return torch.einsum('bfgk,bfk->bg', eqvfeat0, eqvfeat1)

    def __call__(self, data_pr):
        Index = data_pr['DR_true_index'].type(torch.int64)
        feats0 = data_pr['feats0_inv']                                  # bn,f
        feats1 = data_pr['feats1_inv']
        B, L = feats1.shape
        q_vec = feats0.contiguous().view(B, 1, L)
        ans_vecs = feats1.contiguous().view(1, B, L)
        dist = ((q_vec - ans_vecs) ** 2).sum(-1)
        dist = torch.nn.functional.log_softmax(dist, 1)
        loss_true = torch.diag(dist)
        loss_false = torch.min(dist + torch.eye(B, device=dist.device), dim=1)[0]
        loss = torch.mean(torch.clamp_min(loss_true - loss_false + 0.3, 0))
        score = self.eqvloss(data_pr['feats0_eqv_af_conv'], data_pr['feats1_eqv_af_conv'])
        eqv_loss = self.class_loss(score, Index)
        return 5 * loss + eqv_loss


class L1_loss(Loss):
    def __init__(self, cfg):
        super().__init__(['L1_Loss'])
        self.loss = torch.nn.SmoothL1Loss(reduction='sum')

    def __call__(self, patch_op, patch_gt):
        return self.loss(patch_op, patch_gt)


class L2_loss(Loss):
    def __init__(self, cfg):
        super().__init__(['L2_Loss'])
        self.loss = torch.nn.MSELoss(reduction='sum')

    def __call__(self, patch_op, patch_gt):
        return self.loss(patch_op, patch_gt)


class Validation_PartI:
    def __init__(self, cfg):
        self.cfg = cfg
        self.loss = name2loss[self.cfg.loss_type](cfg)

    def recall(self, data):
        feats0, feats1 = data["feats0_inv"], data['feats1_inv']
        bn = feats0.shape[0]
        scores = torch.norm(feats0[None, :, :] - feats1[:, None, :], dim=-1)
        idxs_pr = torch.argmin(scores, 1)
        idxs_gt = torch.arange(bn).to(feats0.device).long()
        return torch.mean((idxs_pr == idxs_gt).float())

    def recall_index(self, data):
        feats0, feats1 = data["feats0_inv"], data['feats1_inv']
        bn = feats0.shape[0]
        scores = torch.norm(feats0[None, :, :] - feats1[:, None, :], dim=-1)
        idxs_pr = torch.argmin(scores, 1)
        idxs_gt = torch.arange(bn).to(feats0.device).long()
        return torch.where(idxs_gt == idxs_pr)[0]

    def __call__(self, model, eval_dataset):
        model.eval()
        alloutput0, alloutput1, allloss, all_batch_recall, all_DR_ok = [], [], [], [], []
        for data in eval_dataset:
            data = to_cuda(data)
            with torch.no_grad():
                outputs = model(data)
                alloutput0.append(outputs['feats0_inv'].cpu())
                alloutput1.append(outputs['feats1_inv'].cpu())
                all_DR_ok.append((outputs['DR_true_index'] == outputs['DR_pre_index']).cpu().numpy())
                allloss.append(self.loss(outputs))
                all_batch_recall.append(self.recall(outputs))
        val_loss = torch.mean(torch.tensor(allloss))
        batch_recall = torch.mean(torch.tensor(all_batch_recall))
        alloutputs = {'feats0_inv': torch.cat(alloutput0, dim=0), 'feats1_inv': torch.cat(alloutput1, dim=0)}
        whole_recall = self.recall(alloutputs)
        ok_index = self.recall_index(alloutputs).cpu().numpy().astype(int)
        all_DR_ok = np.concatenate(all_DR_ok)
        double_ok_rate = np.mean(all_DR_ok[ok_index])
        return {"val_loss": val_loss, "whole_recall": whole_recall, 'batch_recall': batch_recall, 'PartI_DR_ability': double_ok_rate}


class Validation_PartII:
    def __init__(self, cfg):
        self.cfg = cfg
        self.loss = name2loss[self.cfg.loss_type](self.cfg)

    def diff_cal(self, R_pre, R_gt):
        eps = 1e-7
        result = []
        R_pre = R_pre / torch.clamp_min(torch.norm(R_pre, dim=1, keepdim=True), min=1e-4)
        for i in range(R_pre.shape[0]):
            loss_q = torch.clamp_min((1.0 - torch.sum(R_pre[i] * R_gt[i]) ** 2), min=eps)
            err_q = torch.acos(1 - 2 * loss_q)
            result.append(err_q / np.pi * 180)
        return result

    def static(self, errors):
        result = torch.zeros(6)
        for e in errors:
            e_index = int(e)
            if e_index < 6:
                result[e_index] += 1
        result /= errors.shape[0]
        return result

    def __call__(self, model, eval_dataset):
        model.eval()
        part1_ability, all_loss, all_R_error = [], [], []
        for data in eval_dataset:
            quaternion_gt = torch.squeeze(data['deltaR'])
            data = to_cuda(data)
            with torch.no_grad():
                outputs = model(data)
                part1_ability.append(outputs['part1_ability'])
                quaternion = outputs['quaternion_pre'].cpu()
                all_loss.append(self.loss(quaternion, quaternion_gt))
                all_R_error.extend(self.diff_cal(quaternion, quaternion_gt))
        all_loss = torch.Tensor(all_loss)
        all_R_error = torch.Tensor(all_R_error)
        part1_ability = torch.Tensor(part1_ability)
        return {'val_loss': torch.mean(all_loss), 'R_error': torch.mean(all_R_error), 'part1_ability': torch.mean(part1_ability),
                'R_error_statics': self.static(all_R_error)}


name2loss = {'Batch_hard_Rindex_loss': Batch_hard_Rindex_loss, 'L2_loss_partII': L2_loss}
name2val = {"Val_partI": Validation_PartI, 'Val_partII': Validation_PartII}
