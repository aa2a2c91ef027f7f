"""Checkpoint handling for the PartI / PartII group-conv networks.

* ``PARTI_SPEC`` / ``PARTII_SPEC``: the reference's ``state_dict`` key order and shapes
  (reference: utils/network.py:67-79 PartI_network, :140-147 PartI_test, :218-241 PartII_test;
  nn.Sequential indices 0=BatchNorm2d, 2=Conv2d inside Comb_Conv / Residual_Comb_Conv :12-44).
* ``synth_state_dict``: a build-owned deterministic generator (counter-based integer hash,
  pure numpy) so that the GPU box can regenerate bit-identical weights from a seed alone -
  the pretrained ``model_best.pth`` files are absent from the reference tree
  (.MISSING_LARGE_BLOBS) and ``torch.manual_seed`` default init is not stable across
  torch versions.
* ``load_checkpoint``: reads a real ``model_best.pth`` (``['network_state_dict']``) exactly
  as tests/extractor.py:26-34 / :114-122 does.
"""
import zlib
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _bn(prefix, c):
    return [(f"{prefix}.weight", (c,)), (f"{prefix}.bias", (c,)),
            (f"{prefix}.running_mean", (c,)), (f"{prefix}.running_var", (c,)),
            (f"{prefix}.num_batches_tracked", ())]


def _conv(prefix, co, ci, k):
    return [(f"{prefix}.weight", (co, ci, 1, k)), (f"{prefix}.bias", (co,))]


PARTI_SPEC = (
    _conv("PartI_net.Conv_in.0", 256, 32, 13)
    + _bn("PartI_net.SO3_Conv_layers.0.comb_layer_in.0", 256)
    + _conv("PartI_net.SO3_Conv_layers.0.comb_layer_in.2", 512, 256, 13)
    + _bn("PartI_net.SO3_Conv_layers.0.comb_layer_out.0", 512)
    + _conv("PartI_net.SO3_Conv_layers.0.comb_layer_out.2", 256, 512, 13)
    + _bn("PartI_net.Conv_out.comb_layer.0", 256)
    + _conv("PartI_net.Conv_out.comb_layer.2", 32, 256, 13)
)

PARTII_SPEC = (
    _bn("Conv_init.comb_layer.0", 128)
    + _conv("Conv_init.comb_layer.2", 256, 128, 13)
    + _bn("PartII_SO3_Conv_layers.0.comb_layer_in.0", 256)
    + _conv("PartII_SO3_Conv_layers.0.comb_layer_in.2", 512, 256, 13)
    + _bn("PartII_SO3_Conv_layers.0.comb_layer_out.0", 512)
    + _conv("PartII_SO3_Conv_layers.0.comb_layer_out.2", 256, 512, 13)
    + _conv("PartII_To_R_FC.0", 512, 256, 1)
    + _bn("PartII_To_R_FC.1", 512)
    + _conv("PartII_To_R_FC.3", 128, 512, 1)
    + _bn("PartII_To_R_FC.4", 128)
    + _conv("PartII_To_R_FC.6", 4, 128, 1)
)


def fcgf_spec(channels=(None, 32, 64, 128, 256), tr_channels=(None, 64, 64, 64, 128), out_channels=32, conv1_kernel_size=7,
              in_channels=1):
    """state_dict keys / shapes of the FCGF backbone ResUNet2 family (reference fcgf_model/resunet.py:21-139; defaults =
    ResUNetBN2C :200-203 with the 3DMatch settings).  MinkowskiConvolution kernels are (K^3, Cin, Cout), (Cin, Cout) for
    kernel size 1; MinkowskiBatchNorm wraps BatchNorm1d as `.bn`."""
    C, T = channels, tr_channels

    def bnk(prefix, c):
        return _bn(prefix + ".bn", c)

    def blk(prefix, c):
        return ([(f"{prefix}.conv1.kernel", (27, c, c))] + bnk(f"{prefix}.norm1", c)
                + [(f"{prefix}.conv2.kernel", (27, c, c))] + bnk(f"{prefix}.norm2", c))

    spec = [("conv1.kernel", (conv1_kernel_size ** 3, in_channels, C[1]))] + bnk("norm1", C[1]) + blk("block1", C[1])
    spec += [("conv2.kernel", (27, C[1], C[2]))] + bnk("norm2", C[2]) + blk("block2", C[2])
    spec += [("conv3.kernel", (27, C[2], C[3]))] + bnk("norm3", C[3]) + blk("block3", C[3])
    spec += [("conv4.kernel", (27, C[3], C[4]))] + bnk("norm4", C[4]) + blk("block4", C[4])
    spec += [("conv4_tr.kernel", (27, C[4], T[4]))] + bnk("norm4_tr", T[4]) + blk("block4_tr", T[4])
    spec += [("conv3_tr.kernel", (27, C[3] + T[4], T[3]))] + bnk("norm3_tr", T[3]) + blk("block3_tr", T[3])
    spec += [("conv2_tr.kernel", (27, C[2] + T[3], T[2]))] + bnk("norm2_tr", T[2]) + blk("block2_tr", T[2])
    spec += [("conv1_tr.kernel", (C[1] + T[2], T[1])), ("final.kernel", (T[1], out_channels)), ("final.bias", (1, out_channels))]
    return spec


FCGF_SPEC = fcgf_spec()


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def hash_uniform(seed, name, n):
    """n floats in [0,1) keyed by (seed, name, flat_index); float32-exact (24-bit)."""
    with np.errstate(over="ignore"):
        key = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF) * np.uint64(0x100000001B3)
        key = (key ^ (np.uint64(seed) * np.uint64(0xD6E8FEB86659FD93))) & _M64
        ctr = (np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + key) & _M64
        z = _splitmix64(_splitmix64(ctr))
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def synth_state_dict(spec, seed=0):
    """Deterministic, non-trivial weights: conv ~ U(+-sqrt(3/fan_in)), bias +-0.1,
    BN gamma in [0.5,1.5], beta +-0.1, mean +-0.1, var in [0.5,1.5] (so BN != identity)."""
    sd = {}
    for name, shape in spec:
        n = int(np.prod(shape)) if len(shape) else 1
        if name.endswith("num_batches_tracked"):
            sd[name] = np.zeros((), dtype=np.int64)
            continue
        u = hash_uniform(seed, name, n)
        if name.endswith("running_var"):
            v = 0.5 + u
        elif name.endswith("running_mean"):
            v = (u - 0.5) * 0.2
        elif len(shape) == 4:                       # conv weight
            fan_in = shape[1] * shape[3]
            v = (u * 2.0 - 1.0) * np.float32(np.sqrt(3.0 / fan_in))
        elif name.endswith(".kernel"):              # sparse conv kernel (K, Cin, Cout) / (Cin, Cout); ~1/3 of a region is occupied
            fan_in = shape[0] * shape[1] / 3.0 if len(shape) == 3 else shape[0]
            v = (u * 2.0 - 1.0) * np.float32(np.sqrt(3.0 / max(fan_in, 1.0)))
        elif name.endswith(".weight"):              # BN gamma
            v = 0.5 + u
        else:                                        # conv bias / BN beta
            v = (u - 0.5) * 0.2
        sd[name] = np.ascontiguousarray(v.astype(np.float32).reshape(shape))
    return sd


def identity_head(sd2, gain=0.02):
    """A PartII state dict whose quaternion head answers close to the identity rotation: the last 1x1 layer
    (PartII_To_R_FC.6) scaled by `gain`, its bias set to (1, 0, 0, 0).  With random-init weights the head's output is
    an arbitrary rotation, so no YOHO-O hypothesis can ever be right; with this head a hypothesis is R_residual ~ I
    times the coarse group rotation, i.e. right exactly for the pairs whose planted residual rotation is small -
    which is what lets a synthetic scene have a Registration Recall strictly between 0 and 1 (tests/golden/scene6.npz)."""
    sd = {k: np.array(v, copy=True) for k, v in sd2.items()}
    sd["PartII_To_R_FC.6.weight"] = (sd["PartII_To_R_FC.6.weight"] * np.float32(gain)).astype(np.float32)
    sd["PartII_To_R_FC.6.bias"] = np.array([1.0, 0.0, 0.0, 0.0], dtype=np.float32)
    return sd


def to_numpy_state_dict(sd):
    """Accept a torch state_dict or a dict of ndarrays; return contiguous ndarrays."""
    out = {}
    for k, v in sd.items():
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if v.dtype.kind == "f":
            v = v.astype(np.float32)
        out[k] = np.ascontiguousarray(v) if v.ndim else v.copy()
    return out


def load_checkpoint(path):
    """Reference checkpoint format: torch.save({'step','best_para','network_state_dict',...})
    (train/trainer.py:64-71; read at tests/extractor.py:26-31)."""
    import torch
    ck = torch.load(path, map_location="cpu")
    return to_numpy_state_dict(ck["network_state_dict"]), ck.get("best_para", 0)


def save_checkpoint(path, sd, best_para=0.0):
    import torch
    torch.save({"step": 0, "best_para": best_para,
                "network_state_dict": {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}},
               path)


def check_state_dict(sd, spec, strict=True):
    for name, shape in spec:
        if name not in sd:
            if strict or not name.endswith("num_batches_tracked"):
                raise KeyError(f"missing key in state_dict: {name}")
            continue
        if name.endswith("num_batches_tracked"):
            continue
        if tuple(sd[name].shape) != tuple(shape):
            raise ValueError(f"{name}: shape {tuple(sd[name].shape)} != {tuple(shape)}")
