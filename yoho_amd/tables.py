"""Icosahedral group tables (the L1 constant tables of the reference).

The reference loads three ``.npy`` files from ``cfg.SO3_related_files``
(reference: utils/network.py:72-74,112,154-157,223-226; tests/extractor.py:67,110;
tests/estimator.py:283-284):

* ``Rotation.npy``                      (60,3,3) f64   R_g
* ``Nei_Index_in_SO3_ordered_13.npy``   (60,13)  f64-encoded ints  N[g,k] = idx(R_{N[0,k]} R_g)
* ``60_60.npy``                         (60,60)  f64-encoded ints  P[i,g] = idx(R_g R_i)

The column order of ``Nei`` defines the tap order of every (1,13) conv weight, so the
files ship byte-identical as *data* in ``yoho_amd/group_related``; a caller may still
point ``SO3_related_files`` at its own copy exactly as with the reference.
"""
import os
import numpy as np

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "group_related")

G = 60          # group order
NTAP = 13       # self + 12 nearest group neighbours
F = 32          # FCGF feature width


class GroupTables:
    """Host copy of the three tables with the dtypes the HIP library wants."""

    def __init__(self, so3_dir=None):
        d = so3_dir if so3_dir is not None else _PKG_DIR
        if not os.path.exists(os.path.join(d, "Rotation.npy")):
            d = _PKG_DIR
        self.dir = d
        self.R64 = np.load(os.path.join(d, "Rotation.npy")).astype(np.float64)
        self.R32 = self.R64.astype(np.float32)
        nei = np.load(os.path.join(d, "Nei_Index_in_SO3_ordered_13.npy"))
        perm = np.load(os.path.join(d, "60_60.npy"))
        self.N = np.ascontiguousarray(nei.astype(np.int64))      # (60,13)
        self.P = np.ascontiguousarray(perm.astype(np.int64))     # (60,60)
        assert self.R64.shape == (G, 3, 3) and self.N.shape == (G, NTAP) and self.P.shape == (G, G)
        self.N_u8 = np.ascontiguousarray(self.N.astype(np.uint8))
        self.P_u8 = np.ascontiguousarray(self.P.astype(np.uint8))

    # receptive cone of group element 0 (used by the pruned PartII path)
    def cone(self):
        one = [int(v) for v in self.N[0]]                       # 13 elements, tap order
        two = sorted({int(v) for g in one for v in self.N[g]})  # 45 elements
        return one, two


_default = None


def default_tables():
    global _default
    if _default is None:
        _default = GroupTables()
    return _default
