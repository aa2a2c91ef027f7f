"""Training path on the GPU against golden vectors produced by the reference's own networks, losses and
torch autograd (oracle/gen_golden_train.py): forward in train mode, loss, every parameter gradient, BN running stats."""
import os
import types

import numpy as np
import pytest
import torch

from yoho_amd import synth, weights as W

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train.npz")


def rel(a, b):
    a = a.numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def cfg():
    return types.SimpleNamespace(SO3_related_files=None)


def _batch(gold, tables):
    b = synth.train_batch(int(gold["bn"]), tables.P, seed=int(gold["seed"]))
    return {k: torch.from_numpy(v).cuda() for k, v in b.items()}


def test_gconv_layer_forward_and_gradients_vs_torch(hip, tables):
    """yoho_gconv_layer (both directions) and the autograd function against torch's conv2d on the gathered tensor"""
    from yoho_amd.train.network import GroupConv2d
    c = hip.get_context()
    nei = torch.from_numpy(tables.N.astype(np.int64).reshape(-1)).cuda()
    torch.manual_seed(0)
    for cin, cout, B in ((32, 256, 7), (128, 64, 33), (256, 32, 5)):
        layer = GroupConv2d(cin, cout, c, nei).cuda()
        x = torch.randn(B, cin, 60, device="cuda", requires_grad=True)
        y = layer(x)
        xr = x.detach().clone().requires_grad_(True)
        wr, br = layer.weight.detach().clone().requires_grad_(True), layer.bias.detach().clone().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr[:, :, nei].reshape(B, cin, 60, 13), wr, br)[:, :, :, 0]
        assert rel(y.detach().cpu(), yr.detach().cpu()) < 1e-5
        gy = torch.randn_like(y)
        y.backward(gy)
        yr.backward(gy)
        assert rel(x.grad.cpu(), xr.grad.cpu()) < 1e-5, (cin, cout)
        assert rel(layer.weight.grad.cpu(), wr.grad.cpu()) < 1e-4 and rel(layer.bias.grad.cpu(), br.grad.cpu()) < 1e-5


def test_fused_bn_relu_forward_and_gradients_vs_torch(hip):
    """yoho_bn_stats / _bn_relu_apply / _bn_relu_backward (one autograd function) against torch's batch_norm + relu under
    autograd, with batch statistics (training) and with running statistics (eval)"""
    from yoho_amd.train.network import GroupBatchNorm
    c = hip.get_context()
    torch.manual_seed(1)
    for C, B in ((32, 5), (256, 17), (512, 3)):
        for training in (True, False):
            m = GroupBatchNorm(C, hctx=c).cuda()
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 1.5)
            m.train(training)
            rm0, rv0 = m.running_mean.clone(), m.running_var.clone()
            x = (torch.randn(B, C, 60, device="cuda") * 1.7 + 0.3).requires_grad_(True)
            y = m(x)
            xr = x.detach().clone().requires_grad_(True)
            gr, br = m.weight.detach().clone().requires_grad_(True), m.bias.detach().clone().requires_grad_(True)
            rm, rv = rm0.clone(), rv0.clone()
            # the reference normalises the gathered (B,C,60,13) tensor: same mean / biased variance, 13x the element count
            xg = xr[:, :, :, None].expand(B, C, 60, 13)
            yr = torch.relu(torch.nn.functional.batch_norm(xg, rm, rv, gr, br, training, 0.1, 1e-5))[:, :, :, 0]
            assert rel(y.detach().cpu(), yr.detach().cpu()) < 2e-6, (C, training)
            gy = torch.randn_like(y)
            y.backward(gy)
            (yr * gy).sum().backward()
            assert rel(x.grad.cpu(), xr.grad.cpu()) < 2e-5, (C, training)
            assert rel(m.weight.grad.cpu(), gr.grad.cpu()) < 2e-5 and rel(m.bias.grad.cpu(), br.grad.cpu()) < 2e-5
            if training:
                assert rel(m.running_mean.cpu(), rm.cpu()) < 1e-5 and rel(m.running_var.cpu(), rv.cpu()) < 1e-5


def test_partI_train_step_matches_reference(gold, cfg, tables):
    from yoho_amd.train import network, loss_val
    sd1 = W.synth_state_dict(W.PARTI_SPEC, 7)
    net = network.PartI_train(cfg).cuda()
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd1.items()})      # the reference's key names
    net.train()
    o = net(_batch(gold, tables))
    loss = loss_val.Batch_hard_Rindex_loss(cfg)(o)
    loss.backward()
    assert rel(o["feats0_inv"].detach().cpu().numpy(), gold["p1_inv0"]) < 1e-4
    assert rel(o["feats1_eqv_af_conv"].detach().cpu().numpy(), gold["p1_eqv1"]) < 1e-4
    assert np.array_equal(o["DR_pre_index"].cpu().numpy(), gold["p1_pre_idx"])
    assert abs(loss.item() - float(gold["p1_loss"])) < 1e-4 * abs(float(gold["p1_loss"]))
    worst = 0.0
    for name, p in net.named_parameters():
        d, g = synth.tensor_digest(p.grad.cpu().numpy()), gold["p1_grad_" + name]
        # relative to the gradient's norm; conv biases in front of a BatchNorm have an analytically zero gradient
        # (1e-7 rounding noise in the reference as well), hence the floor
        err = np.abs(d - g).max() / max(g[0], 0.05)
        worst = max(worst, err)
        assert err < 2e-4, (name, err)
    for name, b in net.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            d, g = synth.tensor_digest(b.cpu().numpy()), gold["p1_buf_" + name]
            assert np.abs(d - g).max() / max(g[0], 1e-12) < 1e-4, name
    print("PartI_train: loss %.6f, worst gradient digest error %.2e" % (loss.item(), worst))
    sd = net.state_dict()
    assert set(sd.keys()) == {n for n, _ in W.PARTI_SPEC}                              # checkpoint-compatible with the reference


def test_partII_train_step_matches_reference(gold, cfg, tables):
    from yoho_amd.train import network, loss_val
    sd1, sd2 = W.synth_state_dict(W.PARTI_SPEC, 7), W.synth_state_dict(W.PARTII_SPEC, 8)
    net = network.PartII_train(cfg).cuda()
    state = {k: torch.from_numpy(np.array(v)) for k, v in sd2.items()}
    state.update({"PartI_net." + k: torch.from_numpy(np.array(v)) for k, v in sd1.items()})
    net.load_state_dict(state)
    net.train()
    data = _batch(gold, tables)
    keep = data["feats0"].clone()
    o = net(data)
    assert torch.equal(keep, data["feats0"])                                             # inputs are not permuted in place
    loss = loss_val.L2_loss(cfg)(o["quaternion_pre"], torch.squeeze(data["deltaR"]))
    loss.backward()
    assert rel(o["quaternion_pre"].detach().cpu().numpy(), gold["p2_quat"]) < 1e-4
    assert abs(loss.item() - float(gold["p2_loss"])) < 1e-4 * abs(float(gold["p2_loss"]))
    n = 0
    for name, p in net.named_parameters():
        key = "p2_grad_" + name
        if p.grad is None:
            assert key not in gold.files, name
            continue
        d, g = synth.tensor_digest(p.grad.cpu().numpy()), gold[key]
        assert np.abs(d - g).max() / max(g[0], 0.05) < 2e-4, name
        n += 1
    assert n == sum(1 for k in gold.files if k.startswith("p2_grad_"))
    for name, b in net.named_buffers():
        if (name.endswith("running_mean") or name.endswith("running_var")) and not name.startswith("PartI_net"):
            d, g = synth.tensor_digest(b.cpu().numpy()), gold["p2_buf_" + name]
            assert np.abs(d - g).max() / max(g[0], 1e-12) < 1e-4, name


def test_trainer_runs_and_checkpoints(tmp_path, tables):
    """Trainer_partI drop-in on an in-memory dataset: the loss goes down, checkpoints have the reference's format"""
    from yoho_amd.train import trainer

    class DS(torch.utils.data.Dataset):
        def __init__(self, n, off):
            self.items = [{k: torch.from_numpy(v[0]) for k, v in synth.train_batch(8, tables.P, seed=off + i).items()} for i in range(n)]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            return self.items[i]

    cfg = types.SimpleNamespace(SO3_related_files=None, model_fn=str(tmp_path), train_network_type="PartI_train", batch_size=1, worker_num=0,
                                lr_init=1e-3, lr_decay_rate=0.5, lr_decay_step=100, loss_type="Batch_hard_Rindex_loss", val_type="Val_partI",
                                epochs=3, train_log_step=2, val_interval=6, save_interval=6, train_val_sets=(DS(6, 0), DS(2, 100)))
    torch.manual_seed(1)
    tr = trainer.name2trainer["PartI"](cfg)
    first = [tr.train_step(trainer.to_cuda(d), 1).item() for d in tr.train_set]
    tr.run()
    last = [tr.train_step(trainer.to_cuda(d), 1).item() for d in tr.train_set]
    assert np.mean(last) < np.mean(first)
    ck = torch.load(os.path.join(str(tmp_path), "PartI_train", "model.pth"), weights_only=False)
    assert set(ck.keys()) == {"step", "best_para", "network_state_dict", "optimizer_state_dict"}
    assert set(ck["network_state_dict"].keys()) == {n for n, _ in W.PARTI_SPEC}
    assert os.path.exists(os.path.join(str(tmp_path), "PartI_train", "model_best.pth"))
