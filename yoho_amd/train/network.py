"""Trainable YOHO networks: drop-in for utils/network.py:11-211 (Comb_Conv, Residual_Comb_Conv, PartI_network, PartI_train,
PartII_train) with identical module / parameter names, so reference checkpoints load with load_state_dict and checkpoints
written here load in the reference.

The reference materialises the 13-neighbour gather (`data[:, :, Nei].reshape(B, C, 60, 13)`, :46-52) and runs
BatchNorm2d -> ReLU -> Conv2d(Cin, Cout, (1,13)) on it.  Here the gather + convolution is ONE autograd function on the HIP
library (yoho_gconv_layer: forward and data gradient on the fp32 MFMA kernel; yoho_gconv_wgrad: weight and bias gradient read
through the neighbour table), and BatchNorm + ReLU are ONE autograd function as well (yoho_bn_stats / _bn_relu_apply /
_bn_relu_backward) on the un-gathered (B, C, 60) tensor - they are element-wise per channel, so they commute with the gather;
the batch statistics are identical (every element appears exactly 13 times in the gathered tensor), only the
unbiased-variance correction of running_var counts the gathered size B*60*13 as the reference does.  torch autograd only
chains these functions; no einsum / batch_norm / conv kernel of torch runs in a training step of the group-conv stack.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import hip


class _GroupConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, hctx, nei):
        x = x.contiguous()
        ctx.hctx, ctx.nei = hctx, nei
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return hctx.gconv_layer(x, weight.contiguous(), bias.contiguous() if bias is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = ctx.hctx.gconv_layer(dy, weight.contiguous(), None, transpose=True)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dW, db = ctx.hctx.gconv_wgrad(x, dy, want_bias=ctx.has_bias)   # (Cout, Cin, 1, 13), (Cout,)
        return dx, dW, db, None, None


class GroupConv2d(nn.Module):
    """nn.Conv2d(in_dim, out_dim, (1,13)) applied to the 13-neighbour gather of a (B, in_dim, 60) tensor -> (B, out_dim, 60).
    Parameters `weight` (out,in,1,13) / `bias` (out) and their default initialisation are nn.Conv2d's."""

    def __init__(self, in_dim, out_dim, hctx, nei):
        super().__init__()
        ref = nn.Conv2d(in_dim, out_dim, (1, 13), 1)                    # same parameter shapes, init and RNG consumption
        self.weight, self.bias = ref.weight, ref.bias
        self._hctx, self._nei = hctx, nei

    def forward(self, x):
        return _GroupConvFn.apply(x, self.weight, self.bias, self._hctx, self._nei)


class _BNReLUFn(torch.autograd.Function):
    """relu(batch_norm(x)) on the HIP library; forward returns y, backward the gradients of x, gamma, beta"""

    @staticmethod
    def forward(ctx, x, gamma, beta, mean, var, eps, batch_stats, hctx):
        x = x.contiguous()
        rstd = torch.rsqrt(var + eps)
        scale = (gamma * rstd).contiguous()
        y = hctx.bn_relu_apply(x, scale, (beta - mean * scale).contiguous())
        ctx.hctx, ctx.batch_stats = hctx, batch_stats
        ctx.save_for_backward(x, y, gamma.detach().contiguous(), mean.contiguous(), rstd.contiguous())
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, rstd = ctx.saved_tensors
        dx, dg, db = ctx.hctx.bn_relu_backward(x, y, dy.contiguous(), gamma, mean, rstd, ctx.batch_stats)
        return dx, dg, db, None, None, None, None, None


class GroupBatchNorm(nn.Module):
    """nn.BatchNorm2d(C) + nn.ReLU of the reference's gathered (B,C,60,13) tensor, evaluated on the un-gathered (B,C,60) tensor
    (fused: the nn.ReLU that follows it in the reference's Sequential is kept as a parameter-free placeholder so that the
    state_dict keys stay BatchNorm2d's at index 0 - weight, bias, running_mean, running_var, num_batches_tracked - and the conv's
    at index 2; ReLU of a ReLU output is the identity)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, hctx=None):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self._hctx = hctx
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def forward(self, x):                                               # (B, C, 60)
        if self._hctx is not None:
            x = x.contiguous()
            if self.training:
                with torch.no_grad():
                    mean, var = self._hctx.bn_stats(x)
                    n = x.shape[0] * x.shape[2] * 13                     # elements per channel of the gathered tensor
                    self.running_mean.mul_(1 - self.momentum).add_(self.momentum * mean)
                    self.running_var.mul_(1 - self.momentum).add_(self.momentum * var * (n / (n - 1)))
                    self.num_batches_tracked += 1
            else:
                mean, var = self.running_mean, self.running_var
            return _BNReLUFn.apply(x, self.weight, self.bias, mean, var, self.eps, self.training, self._hctx)
        if self.training:
            mean = x.mean((0, 2))
            var = x.var((0, 2), unbiased=False)
            with torch.no_grad():
                n = x.shape[0] * x.shape[2] * 13                         # elements per channel of the gathered tensor
                self.running_mean.mul_(1 - self.momentum).add_(self.momentum * mean)
                self.running_var.mul_(1 - self.momentum).add_(self.momentum * var * (n / (n - 1)))
                self.num_batches_tracked += 1
        else:
            mean, var = self.running_mean, self.running_var
        scale = self.weight * torch.rsqrt(var + self.eps)
        return x * scale[None, :, None] + (self.bias - mean * scale)[None, :, None]


class Comb_Conv(nn.Module):
    def __init__(self, in_dim, out_dim, hctx, nei):
        super().__init__()
        self.comb_layer = nn.Sequential(GroupBatchNorm(in_dim, hctx=hctx), nn.ReLU(), GroupConv2d(in_dim, out_dim, hctx, nei))

    def forward(self, input):                                           # (B, in_dim, 60) -> (B, out_dim, 60)
        return self.comb_layer(input)


class Residual_Comb_Conv(nn.Module):
    def __init__(self, in_dim, middle_dim, out_dim, hctx, nei):
        super().__init__()
        self.comb_layer_in = nn.Sequential(GroupBatchNorm(in_dim, hctx=hctx), nn.ReLU(), GroupConv2d(in_dim, middle_dim, hctx, nei))
        self.comb_layer_out = nn.Sequential(GroupBatchNorm(middle_dim, hctx=hctx), nn.ReLU(), GroupConv2d(middle_dim, out_dim, hctx, nei))
        self.short_cut = False
        if not in_dim == out_dim:
            self.short_cut = True
            self.short_cut_layer = nn.Sequential(GroupBatchNorm(in_dim, hctx=hctx), nn.ReLU(), GroupConv2d(in_dim, out_dim, hctx, nei))

    def forward(self, feat_input):                                      # bn*f*60
        feat = self.comb_layer_out(self.comb_layer_in(feat_input))
        feat_sc = self.short_cut_layer(feat_input) if self.short_cut else feat_input
        return feat + feat_sc


def _tables(cfg):
    hctx = hip.get_context(so3_dir=getattr(cfg, 'SO3_related_files', None))
    dev = torch.device('cuda', hctx.device) if hasattr(hctx, 'device') else torch.device('cuda')
    nei = torch.from_numpy(hctx.tables.N.astype(np.int64).reshape(-1)).to(dev)
    perm = torch.from_numpy(hctx.tables.P.astype(np.int64)).to(dev)
    return hctx, nei, perm


class PartI_network(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self._hctx, self.Nei_in_SO3, _ = _tables(cfg)
        self.Conv_in = nn.Sequential(GroupConv2d(32, 256, self._hctx, self.Nei_in_SO3))
        self.SO3_Conv_layers = nn.ModuleList([Residual_Comb_Conv(256, 512, 256, self._hctx, self.Nei_in_SO3)])
        self.Conv_out = Comb_Conv(256, 32, self._hctx, self.Nei_in_SO3)

    def SO3_Conv(self, data):                                           # data: bn,f,gn
        data = self.Conv_in(data)
        for layer in self.SO3_Conv_layers:
            data = layer(data)
        return self.Conv_out(data)

    def forward(self, feats):
        if feats.dim() == 2:
            feats = feats[None]
        feats_eqv = self.SO3_Conv(feats) + feats
        feats_inv = torch.mean(feats_eqv, dim=-1)
        feats_eqv = feats_eqv / torch.clamp_min(torch.norm(feats_eqv, dim=1, keepdim=True), min=1e-4)
        feats_inv = feats_inv / torch.clamp_min(torch.norm(feats_inv, dim=1, keepdim=True), min=1e-4)
        return {'inv': feats_inv, 'eqv': feats_eqv}


class PartI_train(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.PartI_net = PartI_network(self.cfg)
        self._hctx, _, self.R_index_permu = _tables(cfg)

    def Des2DR(self, Des1, Des2):                                       # before_rot after_rot
        return self._hctx.des2r(Des1.detach().contiguous(), Des2.detach().contiguous())

    def forward(self, data):
        feats0 = torch.squeeze(data['feats0'])                          # bn,32,60
        feats1 = torch.squeeze(data['feats1'])
        true_idxs = torch.squeeze(data['true_idx'])
        yoho_0 = self.PartI_net(feats0)
        yoho_1 = self.PartI_net(feats1)
        pre_idxs = self.Des2DR(yoho_0['eqv'], yoho_1['eqv'])
        part1_ability = torch.mean((pre_idxs == true_idxs).type(torch.float32))
        return {'feats0_eqv_bf_conv': feats0, 'feats1_eqv_bf_conv': feats1,
                'feats0_eqv_af_conv': yoho_0['eqv'], 'feats1_eqv_af_conv': yoho_1['eqv'],
                'feats0_inv': yoho_0['inv'], 'feats1_inv': yoho_1['inv'],
                'DR_pre_ability': part1_ability, 'DR_true_index': true_idxs, 'DR_pre_index': pre_idxs}


class PartII_train(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self._hctx, self.Nei_in_SO3, self.R_index_permu = _tables(cfg)
        self.PartI_net = PartI_train(self.cfg)
        self.Conv_init = Comb_Conv(32 * 4, 256, self._hctx, self.Nei_in_SO3)
        self.PartII_SO3_Conv_layers = nn.ModuleList([Residual_Comb_Conv(256, 512, 256, self._hctx, self.Nei_in_SO3)])
        self.PartII_To_R_dims = [256, 512, 128, 4]
        d = self.PartII_To_R_dims
        self.PartII_To_R_FC = nn.Sequential(
            nn.Conv2d(d[0], d[1], 1, 1), nn.BatchNorm2d(d[1]), nn.ReLU(),
            nn.Conv2d(d[1], d[2], 1, 1), nn.BatchNorm2d(d[2]), nn.ReLU(),
            nn.Conv2d(d[2], d[3], 1, 1))

    def PartII_SO3_Conv(self, data):                                    # data: bn,f,gn
        data = self.Conv_init(data)
        for layer in self.PartII_SO3_Conv_layers:
            data = layer(data)
        return data

    def forward(self, data):
        true_idxs = torch.squeeze(data['true_idx'])
        self.PartI_net.eval()
        with torch.no_grad():
            PartI_output = self.PartI_net(data)
        perm = self.R_index_permu[true_idxs]                            # (bn, 60)
        gather = lambda t: torch.gather(t, 2, perm[:, None, :].expand(-1, t.shape[1], -1))
        feats0 = gather(PartI_output['feats0_eqv_bf_conv'].detach())    # the reference permutes in place, row by row
        feats1 = PartI_output['feats1_eqv_bf_conv'].detach()
        feats0_eqv = gather(PartI_output['feats0_eqv_af_conv'].detach())
        feats1_eqv = PartI_output['feats1_eqv_af_conv'].detach()
        part1_ability = PartI_output['DR_pre_ability'].detach()
        pre_idxs = PartI_output['DR_pre_index'].detach()
        feats_eqv = torch.cat([feats0, feats1, feats0_eqv, feats1_eqv], dim=1)
        feats_eqv = self.PartII_SO3_Conv(feats_eqv)                     # bn f gn
        feats_inv = self.PartII_To_R_FC(feats_eqv.unsqueeze(-1))        # bn 4 gn 1
        quaternion_pre = feats_inv[:, :, 0, 0]
        return {'quaternion_pre': quaternion_pre, 'part1_ability': part1_ability, 'pre_idxs': pre_idxs, 'true_idxs': true_idxs}


def _test_networks():
    from ..network import PartI_test, PartII_test
    return PartI_test, PartII_test


name2network = {'PartI_train': PartI_train, 'PartII_train': PartII_train}
