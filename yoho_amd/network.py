"""Host-side mirror of utils/network.py's test-time networks (PartI_test / PartII_test).

Same constructor (``Cls(cfg)`` reading ``cfg.SO3_related_files``), ``load_state_dict`` /
``eval`` / ``cuda`` and ``forward`` contract, but the arithmetic is the HIP library's
(yoho_partI_forward / yoho_partII_forward).  Differences from the reference, all supersets:
  * any batch size B >= 1 works (the reference's torch.squeeze breaks B == 1, utils/network.py:81);
  * PartII_test.forward does not permute the caller's tensors in place (utils/network.py:266-268).
Network objects share the device context (one yoho_ctx per device) but stay independent like the reference's
nn.Modules: each remembers its own state dict and uploads it again before a forward pass if another object's
weights have been loaded into the context since (``ctx.partI_owner`` / ``ctx.partII_owner``).
The training twins (PartI_train / PartII_train) live in yoho_amd.train.network.
"""
import numpy as np
import torch

from . import hip
from . import weights as W


class _Net:
    SPEC = None

    def __init__(self, cfg):
        self.cfg = cfg
        self.ctx = hip.get_context(so3_dir=getattr(cfg, "SO3_related_files", None))
        self.Rgroup_npy = self.ctx.tables.R32
        self.training = False
        self._sd = None

    # torch.nn.Module surface the reference's callers use
    def cuda(self, *a, **k):
        return self

    def eval(self):
        self.training = False
        return self

    def to(self, *a, **k):
        return self

    def state_dict(self):
        return self._sd

    def load_state_dict(self, sd, strict=True):
        sd = W.to_numpy_state_dict(sd)
        if strict:
            W.check_state_dict(sd, self.SPEC, strict=False)
            extra = set(sd) - {k for k, _ in self.SPEC}
            if extra:
                raise RuntimeError(f"Unexpected key(s) in state_dict: {sorted(extra)[:4]}")
        self._sd = sd
        self._load(sd)

    def _resident(self):
        """make this object's weights the resident ones (another network object may have loaded its own since)"""
        if self._sd is None:
            return                                     # never loaded: the library reports YOHO_ENOWEIGHTS
        if getattr(self.ctx, self.OWNER) is not self:
            self._load(self._sd)

    def __call__(self, x):
        return self.forward(x)


class PartI_test(_Net):
    """utils/network.py:140-147; forward(group_feat (B,32,60)) -> {'inv': (B,32), 'eqv': (B,32,60)}"""
    SPEC = W.PARTI_SPEC
    OWNER = "partI_owner"

    def _load(self, sd):
        self.ctx.load_partI(sd, owner=self)

    def forward(self, group_feat):
        x = group_feat
        if not isinstance(x, torch.Tensor):
            x = torch.from_numpy(np.asarray(x, dtype=np.float32))
        x = x.to(device="cuda", dtype=torch.float32)
        if x.dim() == 2:
            x = x[None]
        self._resident()
        out = self.ctx.partI_forward(x.contiguous(), want_inv=True)
        return {"inv": out["inv"], "eqv": out["eqv"]}


class PartII_test(_Net):
    """utils/network.py:218-278; forward(dict before_eqv0/1, after_eqv0/1, pre_idx) ->
    {'quaternion_pre': (B,4), 'pre_idxs': (B,)}"""
    SPEC = W.PARTII_SPEC
    OWNER = "partII_owner"

    def _load(self, sd):
        self.ctx.load_partII(sd, owner=self)

    def load_state_dict(self, sd, strict=True):
        # tests/extractor.py:119 loads with strict=False (the checkpoint also carries PartI_net.* keys)
        sd = W.to_numpy_state_dict(sd)
        sd = {k: v for k, v in sd.items() if k in {n for n, _ in self.SPEC}} if not strict else sd
        super().load_state_dict(sd, strict=strict)

    def forward(self, data):
        g = lambda k: data[k].to(device="cuda", dtype=torch.float32).contiguous()
        idx = data["pre_idx"].to(device="cuda", dtype=torch.int64).contiguous()
        self._resident()
        q = self.ctx.partII_forward(g("before_eqv0"), g("before_eqv1"), g("after_eqv0"), g("after_eqv1"), idx)
        return {"quaternion_pre": q, "pre_idxs": data["pre_idx"]}


name2network = {"PartI_test": PartI_test, "PartII_test": PartII_test}
