"""Trainer drop-in for train/trainer.py (Trainer_partI :22-141, Trainer_partII :143-266) and the two training datasets of
utils/dataset.py:241-317 (pre-generated batches `{output_cache_fn}/Train_val_list/{trainset,valset}/{i}.pth`).

Same cfg attributes (model_fn, train_network_type, trainset_type, batch_size, worker_num, lr_init, lr_decay_rate,
lr_decay_step, loss_type, val_type, epochs, train_log_step, val_interval, save_interval, PartI_pretrained_model_fn,
train_pcpair_list_fn, val_pppair_list_fn, output_cache_fn, SO3_related_files), same checkpoint files
(`model.pth`, `model_best.pth` = {'step','best_para','network_state_dict','optimizer_state_dict'}), same loop: Adam,
exponential step decay of the learning rate, validation / best-model / periodic saves.  The tensorboard logger and the tqdm
bar are replaced by a plain text log (`{model_dir}/train.log`)."""
import os
import pickle

import numpy as np
import torch
from torch.optim import Adam
from torch.utils.data import DataLoader, Dataset

from .. import hip
from ..utils import quaternion_from_matrix
from . import loss_val
from . import network
from .loss_val import to_cuda


def read_pickle(fn):
    with open(fn, 'rb') as f:
        return pickle.load(f)


class Enhanced_train_dataset_PartI(Dataset):
    def __init__(self, cfg, is_training=True):
        self.cfg = cfg
        self.output_dir = self.cfg.output_cache_fn
        self.is_training = is_training
        self.Rgroup = hip.get_context(so3_dir=getattr(cfg, 'SO3_related_files', None)).tables.R32
        if self.is_training:
            self.name_pair_ids = read_pickle(cfg.train_pcpair_list_fn)          # list: name id0 id1 pt1 pt2
        else:
            self.name_pair_ids = read_pickle(cfg.val_pppair_list_fn)[0:3000]

    def __getitem__(self, index):
        sub = 'trainset' if self.is_training else 'valset'
        return torch.load(f'{self.output_dir}/Train_val_list/{sub}/{index}.pth', weights_only=False)

    def __len__(self):
        return len(self.name_pair_ids)


class Enhanced_train_dataset_PartII(Enhanced_train_dataset_PartI):
    def DeltaR(self, R, index):
        R_anchor = self.Rgroup[index]                                            # R = Rres @ Ranc -> Rres = R @ Ranc.T
        return quaternion_from_matrix(R @ R_anchor.T)

    def __getitem__(self, index):
        item = super().__getitem__(index)
        if not self.is_training:
            deltaR = self.DeltaR(item['R'].numpy(), int(item['true_idx']))
            item['deltaR'] = torch.from_numpy(deltaR.astype(np.float32))
        return item


name2traindataset = {"Enhanced_train_dataset_PartI": Enhanced_train_dataset_PartI,
                     "Enhanced_train_dataset_PartII": Enhanced_train_dataset_PartII}


class ExpDecayLR():
    def __init__(self, cfg, decay_step):
        self.lr_init = cfg.lr_init
        self.decay_step = decay_step
        self.decay_rate = cfg.lr_decay_rate

    def __call__(self, step, *args, **kwargs):
        return self.lr_init * (self.decay_rate ** (step // self.decay_step))


def reset_learning_rate(optimizer, lr):
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr
    return lr


class _TrainerBase:
    part = None

    def __init__(self, cfg):
        self.cfg = cfg
        self.model_dir = f'{self.cfg.model_fn}/{self.cfg.train_network_type}'
        os.makedirs(self.model_dir, exist_ok=True)
        self.pth_fn = os.path.join(self.model_dir, 'model.pth')
        self.best_pth_fn = os.path.join(self.model_dir, 'model_best.pth')
        self._init_dataset()
        self._init_network()

    def _init_dataset(self):
        sets = getattr(self.cfg, 'train_val_sets', None)                         # optional: (train Dataset, val Dataset)
        if sets is None:
            cls = name2traindataset[self.cfg.trainset_type]
            sets = (cls(self.cfg, is_training=True), cls(self.cfg, is_training=False))
        nw = getattr(self.cfg, 'worker_num', 0)
        self.train_set = DataLoader(sets[0], 1, shuffle=True, num_workers=nw)
        self.val_set = DataLoader(sets[1], self.cfg.batch_size, shuffle=False, num_workers=nw, drop_last=True)

    def _make_network(self):
        return network.name2network[self.cfg.train_network_type](self.cfg).cuda()

    def _init_network(self):
        self.network = self._make_network()
        self.optimizer = Adam(filter(lambda p: p.requires_grad, self.network.parameters()), lr=self.cfg.lr_init)
        self.loss = loss_val.name2loss[self.cfg.loss_type](self.cfg)
        self.val_evaluator = loss_val.name2val[self.cfg.val_type](self.cfg)
        self.lr_setter = ExpDecayLR(self.cfg, len(self.train_set) * self.cfg.lr_decay_step)

    def _load_model(self, best_init):
        best_para, start_step = best_init, 0
        if os.path.exists(self.pth_fn):
            checkpoint = torch.load(self.pth_fn, weights_only=False)
            best_para, start_step = checkpoint['best_para'], checkpoint['step']
            self.network.load_state_dict(checkpoint['network_state_dict'])
            self.optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
            print(f'==> resuming from step {start_step} best para {best_para}')
        return best_para, start_step

    def _save_model(self, step, best_para, save_fn=None):
        torch.save({'step': step, 'best_para': best_para, 'network_state_dict': self.network.state_dict(),
                    'optimizer_state_dict': self.optimizer.state_dict()}, self.pth_fn if save_fn is None else save_fn)

    def _log(self, prefix, step, results):
        with open(os.path.join(self.model_dir, 'train.log'), 'a') as f:
            vals = ' '.join(f'{k} {float(np.mean(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v)):.6f}'
                            for k, v in results.items() if np.ndim(v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) == 0)
            f.write(f'{prefix} step {step} {vals}\n')

    # -- the two hooks that differ between PartI and PartII
    def _step_loss(self, train_data, outputs):
        raise NotImplementedError

    def _is_better(self, val_results, best_para):
        raise NotImplementedError

    def train_step(self, train_data, step):
        """one optimiser step (train/trainer.py:100-111); returns the detached loss"""
        self.network.train()
        reset_learning_rate(self.optimizer, self.lr_setter(step))
        self.optimizer.zero_grad()
        self.network.zero_grad()
        outputs = self.network(train_data)
        loss = self._step_loss(train_data, outputs)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def run(self):
        best_para, start_step = self._load_model(self.best_init)
        step, wholeloss = start_step, 0
        start_epoch = start_step // len(self.train_set)
        whole_step = len(self.train_set) * self.cfg.epochs
        for epoch in range(start_epoch, self.cfg.epochs):
            for train_data in self.train_set:
                step += 1
                train_data = to_cuda(train_data)
                wholeloss += self.train_step(train_data, step)
                if (step + 1) % self.cfg.train_log_step == 0:
                    self._log('train', step + 1, {'loss': wholeloss / self.cfg.train_log_step})
                    wholeloss = 0
                if (step + 1) % self.cfg.val_interval == 0:
                    val_results = self.val_evaluator(self.network, self.val_set)
                    better, val_para = self._is_better(val_results, best_para)
                    if better:
                        best_para = val_para
                        self._save_model(step + 1, best_para, self.best_pth_fn)
                    self._log('val', step + 1, val_results)
                if (step + 1) % self.cfg.save_interval == 0:
                    self._save_model(step + 1, best_para)
                if step >= whole_step:
                    return


class Trainer_partI(_TrainerBase):
    best_init = 0

    def _step_loss(self, train_data, outputs):
        return self.loss(outputs)

    def _is_better(self, val_results, best_para):
        v = val_results['whole_recall']
        return v >= best_para, v


class Trainer_partII(_TrainerBase):
    best_init = 100

    def _make_network(self):
        net = super()._make_network()
        pre = torch.load(self.cfg.PartI_pretrained_model_fn, weights_only=False)['network_state_dict']
        net.load_state_dict({f'PartI_net.{k}': v for k, v in pre.items()}, strict=False)
        return net

    def _step_loss(self, train_data, outputs):
        return self.loss(outputs['quaternion_pre'], torch.squeeze(train_data['deltaR']))

    def _is_better(self, val_results, best_para):
        v = val_results['R_error']
        return v <= best_para, v


name2trainer = {'PartI': Trainer_partI, 'PartII': Trainer_partII}
