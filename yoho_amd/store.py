"""HBM-resident stage store.

The reference passes every stage through .npy files (~300 MB of reads per
pair).  The drop-in classes keep writing those files byte-compatibly, but they also park the
device tensors here so that the next stage of the same process reads HBM instead of disk.
With 288 GB per MI355X a whole 3DMatch scene (<= 66 fragments x 2 x 38.4 MB) stays resident.
"""
import os
import numpy as np
import torch

_MAX_BYTES = int(os.environ.get("YOHO_STORE_BYTES", str(64 << 30)))
_store = {}
_bytes = 0


def put(key, tensor):
    global _bytes
    if key in _store:
        _bytes -= _store[key].numel() * _store[key].element_size()
    nb = tensor.numel() * tensor.element_size()
    while _store and _bytes + nb > _MAX_BYTES:
        k, v = next(iter(_store.items()))
        _bytes -= v.numel() * v.element_size()
        del _store[k]
    _store[key] = tensor
    _bytes += nb


def get(key):
    return _store.get(key)


def clear():
    global _bytes
    _store.clear()
    _bytes = 0


def load_npy(path, dtype=torch.float32):
    """Device tensor for a cached .npy stage file: resident copy if we produced it, else disk."""
    key = os.path.abspath(path)
    t = get(key)
    if t is None:
        t = torch.from_numpy(np.ascontiguousarray(np.load(path))).to(device="cuda", dtype=dtype)
        put(key, t)
    return t


def save_npy(path, tensor):
    np.save(path, tensor.cpu().numpy())
    put(os.path.abspath(path), tensor)
